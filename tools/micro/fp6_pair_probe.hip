// Probe: v_cvt_scalef32_2xpk16_fp6_f32 -> v_mfma_scale_f32_16x16x128_f8f6f4 (fp6 e2m3) end to end on random data.  The kernel
// dumps the converted fragments and scale bytes; the host decodes them and evaluates D under the assumed semantics
// (element j of lane (r, g) of A pairs with element j of lane (c, g) of B; block scale = the lane's own byte).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) int i8v;
typedef __attribute__((ext_vector_type(6))) unsigned u6v;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;

__global__ void k(const float* A, const float* B, const int* sa, const int* sb, float* D, unsigned* dump) {
    const int l = threadIdx.x;
    f16v a0, a1, b0, b1;
    for (int i = 0; i < 16; ++i) {
        a0[i] = A[l * 32 + i]; a1[i] = A[l * 32 + 16 + i];
        b0[i] = B[l * 32 + i]; b1[i] = B[l * 32 + 16 + i];
    }
    const u6v pa = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a0, a1, __uint_as_float((unsigned)sa[l] << 23));
    const u6v pb = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(b0, b1, __uint_as_float((unsigned)sb[l] << 23));
    for (int i = 0; i < 6; ++i) { dump[l * 12 + i] = pa[i]; dump[l * 12 + 6 + i] = pb[i]; }
    i8v a = {(int)pa[0], (int)pa[1], (int)pa[2], (int)pa[3], (int)pa[4], (int)pa[5], 0, 0};
    i8v b = {(int)pb[0], (int)pb[1], (int)pb[2], (int)pb[3], (int)pb[4], (int)pb[5], 0, 0};
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

static double dec(unsigned code) {
    const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
    const double v = e == 0 ? m / 8.0 : (1.0 + m / 8.0) * (double)(1 << (e - 1));
    return s ? -v : v;
}

int main() {
    srand(5);
    float hA[64 * 32], hB[64 * 32];
    int hsa[64], hsb[64];
    for (int l = 0; l < 64; ++l) {
        hsa[l] = 124 + rand() % 6; hsb[l] = 125 + rand() % 5;
        for (int i = 0; i < 32; ++i) {
            hA[l * 32 + i] = ldexpf((rand() % 1000) / 70.f - 7.f, hsa[l] - 127);
            hB[l * 32 + i] = ldexpf((rand() % 1000) / 70.f - 7.f, hsb[l] - 127);
        }
    }
    float *dA, *dB, *dD; int *dsa, *dsb; unsigned* dd;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, 1024); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 64 * 12 * 4);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dsa, dsb, dD, dd);
    float hD[256]; unsigned hd[64 * 12];
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost); hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    auto code = [&](int l, int ab, int j) { unsigned c = 0; const unsigned* w = hd + l * 12 + 6 * ab;
        for (int t = 0; t < 6; ++t) { const int bit = 6 * j + t; c |= ((w[bit >> 5] >> (bit & 31)) & 1u) << t; } return c; };
    // hypotheses on the pairing of A element ja with B element jb inside lane group g
    const char* names[4] = {"same j", "A j <-> B j^1 (swap pair partners)", "A j <-> B (j>>1)+16*(j&1) (de-interleaved)", "A de-interleaved, B interleaved"};
    for (int hyp = 0; hyp < 4; ++hyp) {
        double worst = 0, mx = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int col = l & 15, row = (l >> 4) * 4 + r;
            double ref = 0;
            for (int g = 0; g < 4; ++g) {
                const int la = row + 16 * g, lb = col + 16 * g;
                double s = 0;
                for (int j = 0; j < 32; ++j) {
                    int jb = j;
                    if (hyp == 1) jb = j ^ 1;
                    if (hyp == 2) jb = (j >> 1) + 16 * (j & 1);
                    s += dec(code(la, 0, j)) * dec(code(lb, 1, jb));
                }
                ref += s * ldexp(1.0, hsa[la] - 127) * ldexp(1.0, hsb[lb] - 127);
            }
            worst = fmax(worst, fabs(ref - hD[l * 4 + r])); mx = fmax(mx, fabs(ref));
        }
        printf("hypothesis '%s': max |D - expected| / max|D| = %.3e\n", names[hyp], worst / mx);
    }
    // and against the un-quantised inputs (pairing as the data was generated: a0[i]*b0[i] + a1[i]*b1[i])
    double worst = 0, mx = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int col = l & 15, row = (l >> 4) * 4 + r;
        double ref = 0;
        for (int g = 0; g < 4; ++g) for (int i = 0; i < 32; ++i) ref += (double)hA[(row + 16 * g) * 32 + i] * hB[(col + 16 * g) * 32 + i];
        worst = fmax(worst, fabs(ref - hD[l * 4 + r])); mx = fmax(mx, fabs(ref));
    }
    printf("against the fp32 inputs (same-index pairing): %.3e\n", worst / mx);
    return 0;
}
