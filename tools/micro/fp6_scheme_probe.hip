// Probe: the "fp16 + MX-fp6 corrections" contraction on one 16x16 tile over 64 channels, exactly as the conv kernel would issue
// it: two v_mfma_f32_16x16x32_f16 on fp16(x) * fp16(w) plus ONE v_mfma_scale_f32_16x16x128_f8f6f4 with e2m3 operands whose
// K = 128 carries both first-order corrections, operands produced by v_cvt_scalef32_2xpk16_fp6_f32 with a per-lane (= per 32
// K-elements) E8M0 block scale.  Prints the error against a float64 dot product next to fp16-only and to the fp8 (uniform
// scale) variant.  Also checks the pieces it relies on: the conversion divides by the scale and saturates, and a lane's scale
// byte acts on that lane's own 32 K-elements.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i8v;
typedef __attribute__((ext_vector_type(6))) unsigned u6v;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// hipcc (ROCm 7.2) lets the destination of v_cvt_scalef32_2xpk16_fp6_f32 overlap its scale / source registers (seen:
// "v[8:13], v[34:49], v[50:65], v8"), and the instruction writes results before it has read everything: the outputs past the
// overlap are garbage.  The early-clobber constraint keeps the destination disjoint from every input.
__device__ inline u6v cvt_2xpk16_fp6(const f16v& a, const f16v& b, float scale) {
    u6v out;
    asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(out) : "v"(a), "v"(b), "v"(scale));
    return out;
}

__device__ inline unsigned e8m0_of_max(float mx) {
    // smallest power of two 2^e with mx / 2^e <= 7.5 (e2m3 max), as a biased E8M0 byte, >= 13 so that byte - 12 stays valid
    const float q = mx * (1.0f / 7.5f);
    unsigned b = (__float_as_uint(q) + 0x7FFFFFu) >> 23;
    return b < 13u ? 13u : (b > 254u ? 254u : b);
}

__global__ void k(const float* A, const float* B, float* D, int mode, unsigned* dump = nullptr) {
    // A [16][64] (row-major: weights, row = output channel), B [64][16] (activations, col = position)
    const int l = threadIdx.x, rc = l & 15, kq = l >> 4;
    f4 c = {0, 0, 0, 0};
    h8 a0, a1, b0, b1;
    for (int j = 0; j < 8; ++j) {
        a0[j] = (_Float16)A[rc * 64 + 8 * kq + j];
        a1[j] = (_Float16)A[rc * 64 + 32 + 8 * kq + j];
        b0[j] = (_Float16)B[(8 * kq + j) * 16 + rc];
        b1[j] = (_Float16)B[(32 + 8 * kq + j) * 16 + rc];
    }
    const bool corr_only = mode & 4;
    mode &= 3;
    if (!corr_only) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c, 0, 0, 0);
    }
    if (mode == 1) {  // fp6 corrections with per-lane block scales
        f16v w0, w1, x0, x1;
        float mw = 0.f, mx = 0.f;
        for (int i = 0; i < 16; ++i) {
            const float w = A[rc * 64 + 16 * kq + i], x = B[(16 * kq + i) * 16 + rc];
            w0[i] = w; w1[i] = (w - (float)(_Float16)w) * 4096.f;
            x0[i] = (x - (float)(_Float16)x) * 4096.f; x1[i] = x;
            mw = fmaxf(mw, fmaxf(fabsf(w0[i]), fabsf(w1[i])));
            mx = fmaxf(mx, fmaxf(fabsf(x0[i]), fabsf(x1[i])));
        }
        // A lane's elements 0..15 / 16..31 lie in scale blocks (kq >> 1) / 2 + (kq >> 1) (32 consecutive k each) and the scale of
        // block blk is read from lane rc + 16 * blk: lanes kq = 2h, 2h + 1 must convert with ONE common scale (the maximum over
        // the 32 channels 32h .. 32h + 31, both kinds), and lane kq supplies the scale of the pair kq & 1.
        const unsigned ba = e8m0_of_max(mw), bb = e8m0_of_max(mx);
        const unsigned ba_s = ba, bb_s = bb;
        const u6v pa = cvt_2xpk16_fp6(w0, w1, __uint_as_float(ba << 23));
        const u6v pb = cvt_2xpk16_fp6(x0, x1, __uint_as_float(bb << 23));
        i8v a = {(int)pa[0], (int)pa[1], (int)pa[2], (int)pa[3], (int)pa[4], (int)pa[5], 0, 0};
        i8v b = {(int)pb[0], (int)pb[1], (int)pb[2], (int)pb[3], (int)pb[4], (int)pb[5], 0, 0};
        if (dump) {
            for (int i = 0; i < 6; ++i) { dump[l * 16 + i] = pa[i]; dump[l * 16 + 6 + i] = pb[i]; }
            dump[l * 16 + 12] = ba; dump[l * 16 + 13] = bb;
        }
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, (int)ba_s, 0, (int)(bb_s - 12u));
    } else if (mode == 2) {  // the built fp8 scheme (uniform scales)
        int a[8], b[8];
        for (int q = 0; q < 4; ++q) {
            float w[4], wl[4], x[4], xl[4];
            for (int j = 0; j < 4; ++j) {
                const float wv = A[rc * 64 + 16 * kq + 4 * q + j], xv = B[(16 * kq + 4 * q + j) * 16 + rc];
                w[j] = __builtin_amdgcn_fmed3f(wv, -448.f, 448.f);
                wl[j] = __builtin_amdgcn_fmed3f((wv - (float)(_Float16)wv) * 4096.f, -448.f, 448.f);
                x[j] = __builtin_amdgcn_fmed3f(xv, -448.f, 448.f);
                xl[j] = __builtin_amdgcn_fmed3f((xv - (float)(_Float16)xv) * 4096.f, -448.f, 448.f);
            }
            a[q] = __builtin_amdgcn_cvt_pk_fp8_f32(w[0], w[1], 0, false); a[q] = __builtin_amdgcn_cvt_pk_fp8_f32(w[2], w[3], a[q], true);
            a[4 + q] = __builtin_amdgcn_cvt_pk_fp8_f32(wl[0], wl[1], 0, false); a[4 + q] = __builtin_amdgcn_cvt_pk_fp8_f32(wl[2], wl[3], a[4 + q], true);
            b[q] = __builtin_amdgcn_cvt_pk_fp8_f32(xl[0], xl[1], 0, false); b[q] = __builtin_amdgcn_cvt_pk_fp8_f32(xl[2], xl[3], b[q], true);
            b[4 + q] = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false); b[4 + q] = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], b[4 + q], true);
        }
        i8v av = {a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]}, bv = {b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7]};
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, 127, 0, 115);
    }
    for (int r = 0; r < 4; ++r) D[(4 * kq + r) * 16 + rc] = c[r];
}

// conversion facts: out = fp6(in / scale); overflow saturates to 7.5
__global__ void kcvt(float* out, const float* in, float scale) {
    f16v a, b;
    for (int i = 0; i < 16; ++i) { a[i] = in[i]; b[i] = in[16 + i]; }
    const u6v p = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    if (threadIdx.x == 0) for (int i = 0; i < 6; ++i) reinterpret_cast<unsigned*>(out)[i] = p[i];
}

static float silu(float u) { return u / (1.f + expf(-u)); }

int main() {
    unsigned hp[6];
    float* dp; hipMalloc(&dp, 64);
    float hin[32], *din;
    for (int i = 0; i < 16; ++i) { hin[i] = (float)i; hin[16 + i] = 100.f + i; }
    hipMalloc(&din, 128);
    hipMemcpy(din, hin, 128, hipMemcpyHostToDevice);
    kcvt<<<1, 64>>>(dp, din, 2.0f);
    hipMemcpy(hp, dp, 24, hipMemcpyDeviceToHost);
    printf("raw dwords: %08x %08x %08x %08x %08x %08x\n", hp[0], hp[1], hp[2], hp[3], hp[4], hp[5]);
    printf("cvt_scalef32_2xpk16_fp6_f32(a = 0..15, b = 100..115, scale 2): 6-bit codes:");
    for (int j = 0; j < 32; ++j) {
        unsigned code = 0;
        for (int t = 0; t < 6; ++t) { const int bit = 6 * j + t; code |= ((hp[bit >> 5] >> (bit & 31)) & 1u) << t; }
        printf(" %02x", code);
    }
    printf("\n  (e2m3: 0x08 = 1.0, 0x10 = 2, 0x18 = 4, 0x1f = 7.5; a[i] / 2 -> i/2; b -> saturated 0x1f if the conversion saturates)\n");

    for (int i = 0; i < 16; ++i) { hin[i] = 0.125f * i; hin[16 + i] = -(0.9f + 0.05f * i); }
    hipMemcpy(din, hin, 128, hipMemcpyHostToDevice);
    kcvt<<<1, 64>>>(dp, din, 1.0f);
    hipMemcpy(hp, dp, 24, hipMemcpyDeviceToHost);
    printf("a = 0.125 i, b = -(0.9 + 0.05 i), scale 1:");
    for (int j = 0; j < 32; ++j) {
        unsigned code = 0;
        for (int t = 0; t < 6; ++t) { const int bit = 6 * j + t; code |= ((hp[bit >> 5] >> (bit & 31)) & 1u) << t; }
        printf(" %02x", code);
    }
    printf("\n");
    srand(1);
    auto rnd = [] { return (rand() + 0.5f) / (RAND_MAX + 1.0f); };
    auto gauss = [&] { return sqrtf(-2.f * logf(rnd())) * cosf(6.2831853f * rnd()); };
    double worst[3] = {0, 0, 0}, rms[3] = {0, 0, 0};
    const int trials = 200;
    float *dA, *dB, *dD;
    hipMalloc(&dA, 16 * 64 * 4); hipMalloc(&dB, 64 * 16 * 4); hipMalloc(&dD, 256 * 4);
    for (int t = 0; t < trials; ++t) {
        std::vector<float> A(16 * 64), B(64 * 16), D(256);
        const float wscale = (t % 3 == 0) ? 0.02f : ((t % 3 == 1) ? 0.3f : 5.f);
        for (auto& v : A) v = wscale * gauss() * ((rand() % 50 == 0) ? 8.f : 1.f);          // heavy-tailed weights
        for (int i = 0; i < 64 * 16; ++i) {
            float u = gauss() * ((rand() % 40 == 0) ? 10.f : 1.f);                            // SiLU(GN) activations with outliers
            if (t % 4 == 3) u *= 300.f * (1 + (i / 16) % 7);                                   // residual-stream magnitudes (inside fp16 range), per-channel spread
            B[i] = (t % 2) ? silu(u) : u;
        }
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 3; ++mode) {
            k<<<1, 64>>>(dA, dB, dD, mode);
            hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
            double mx = 0, err = 0, se = 0;
            for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
                double ref = 0;
                for (int kk = 0; kk < 64; ++kk) ref += (double)A[r * 64 + kk] * B[kk * 16 + c];
                mx = fmax(mx, fabs(ref)); err = fmax(err, fabs(ref - D[r * 16 + c])); se += (ref - D[r * 16 + c]) * (ref - D[r * 16 + c]);
            }
            worst[mode] = fmax(worst[mode], err / mx);
            rms[mode] += sqrt(se / 256) / mx / trials;
        }
    }
    {   // the correction term alone against its exact value sum(w * xl + wl * x)
        std::vector<float> A(16 * 64), B(64 * 16), D(256);
        for (auto& v : A) v = 0.02f * gauss();
        for (auto& v : B) v = gauss();
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        unsigned* ddump; hipMalloc(&ddump, 64 * 16 * 4);
        unsigned hdump[64 * 16];
        for (int mode = 1; mode <= 2; ++mode) {
            k<<<1, 64>>>(dA, dB, dD, mode | 4, ddump);
            if (mode == 1) {
                hipMemcpy(hdump, ddump, sizeof(hdump), hipMemcpyDeviceToHost);
                auto code = [&](int l, int ab, int j) { unsigned c = 0; const unsigned* w = hdump + l * 16 + 6 * ab;
                    for (int t = 0; t < 6; ++t) { const int bit = 6 * j + t; c |= ((w[bit >> 5] >> (bit & 31)) & 1u) << t; } return c; };
                auto dec = [](unsigned cd) { const int sg = (cd >> 5) & 1, e = (cd >> 3) & 3, m = cd & 7;
                    const double v = e == 0 ? m / 8.0 : (1.0 + m / 8.0) * (double)(1 << (e - 1)); return sg ? -v : v; };
                // lane 0 (row 0 / col 0, channels 0..15): what was converted against what should have been
                printf("   lane 0: ba %u bb %u\n", hdump[12], hdump[13]);
                for (int i = 0; i < 4; ++i) {
                    const float w = A[i], x = B[i * 16];
                    printf("   ch %d: w %.5e -> %.5e | wl*4096 %.5e -> %.5e | xl*4096 %.5e -> %.5e | x %.5e -> %.5e\n", i,
                           w, dec(code(0, 0, 2 * i)) * ldexp(1.0, (int)hdump[12] - 127), (w - (float)(_Float16)w) * 4096.f, dec(code(0, 0, 2 * i + 1)) * ldexp(1.0, (int)hdump[12] - 127),
                           (x - (float)(_Float16)x) * 4096.f, dec(code(0, 1, 2 * i)) * ldexp(1.0, (int)hdump[13] - 127), x, dec(code(0, 1, 2 * i + 1)) * ldexp(1.0, (int)hdump[13] - 127));
                }
            }
            hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
            double num = 0, den = 0;
            for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
                double ref = 0;
                for (int kk = 0; kk < 64; ++kk) {
                    const double w = A[r * 64 + kk], x = B[kk * 16 + c];
                    const double wh = (double)(float)(_Float16)A[r * 64 + kk], xh = (double)(float)(_Float16)B[kk * 16 + c];
                    ref += wh * (x - xh) + (w - wh) * xh;
                }
                num += (ref - D[r * 16 + c]) * (ref - D[r * 16 + c]); den += ref * ref;
                if (r == 0 && c < 4) printf("   corr term mode %d [0][%d]: got %.6e want %.6e\n", mode, c, D[c], ref);
            }
            printf("correction term alone, mode %d: relative rms error %.3e\n", mode, sqrt(num / den));
        }
    }
    const char* names[3] = {"fp16 only", "fp16 + fp6 corrections (per-lane block scales)", "fp16 + fp8 corrections (uniform scales, built)"};
    for (int m = 0; m < 3; ++m) printf("%-52s max err / max|D| %.3e   mean rms err / max|D| %.3e\n", names[m], worst[m], rms[m]);
    return 0;
}
