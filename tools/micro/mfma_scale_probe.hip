// Probe (exact small integers): operand / scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 (e4m3) operands, and the
// semantics of the fp8 conversion instructions.  Developer tool; results are recorded in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int i8v;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) short s2v;

__global__ void k_mfma(const int* A, const int* B, const int* SA, const int* SB, float* D) {
    const int l = threadIdx.x;
    i8v a, b;
    for (int i = 0; i < 8; ++i) { a[i] = A[l * 8 + i]; b[i] = B[l * 8 + i]; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, SA[l], 0, SB[l]);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

typedef __attribute__((ext_vector_type(16))) float f16v;
__global__ void k_mfma32(const int* A, const int* B, float* D) {
    const int l = threadIdx.x;
    i8v a, b;
    for (int i = 0; i < 8; ++i) { a[i] = A[l * 8 + i]; b[i] = B[l * 8 + i]; }
    f16v c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 126);  // B scale 2^-1
    for (int r = 0; r < 16; ++r) D[l * 16 + r] = c[r];
}

__global__ void k_cvt(float* out, unsigned* bits) {
    // non-scaled conversion: which e4m3 flavour?
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(1.0f, 2.0f, 0, false);
    bits[0] = (unsigned)w;
    int w2 = __builtin_amdgcn_cvt_pk_fp8_f32(0.3f, -448.0f, 0, false);
    bits[1] = (unsigned)w2;
    int w3 = __builtin_amdgcn_cvt_pk_fp8_f32(1000.0f, 0.001f, 0, false);  // saturation / underflow
    bits[2] = (unsigned)w3;
    // scaled conversion: multiply or divide by the scale?
    s2v z = {0, 0};
    s2v r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, 8.0f, 1.0f, 4.0f, false);
    bits[3] = (unsigned)(unsigned short)r[0] | ((unsigned)(unsigned short)r[1] << 16);
    s2v r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, 1.0e6f, -1.0e6f, 1.0f, false);  // saturates or NaN?
    bits[4] = (unsigned)(unsigned short)r2[0];
    s2v r3 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, 3.0f, 5.0f, 1.0f, true);         // which half does `true` write?
    bits[5] = (unsigned)(unsigned short)r3[0] | ((unsigned)(unsigned short)r3[1] << 16);
    out[0] = 0;
}

static unsigned char f8(int v) {  // exact e4m3 (OCP, bias 7) encoding of small non-negative integers 0..15
    static const unsigned char t[16] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50, 0x51, 0x52, 0x53, 0x54, 0x55, 0x56, 0x57};
    return t[v];
}

int main() {
    // ---- conversions
    float* dout; unsigned* dbits;
    hipMalloc(&dout, 64); hipMalloc(&dbits, 64);
    k_cvt<<<1, 1>>>(dout, dbits);
    unsigned hb[6];
    hipMemcpy(hb, dbits, 24, hipMemcpyDeviceToHost);
    printf("cvt_scalef32_pk_fp8(1e6, -1e6, scale 1) = 0x%04x (0x7e/0xfe = saturates, 0x7f = NaN)\n", hb[4]);
    printf("cvt_scalef32_pk_fp8(3, 5, hi=true) on zero = 0x%08x\n", hb[5]);
    printf("cvt_pk_fp8(1.0, 2.0) = 0x%08x   (OCP e4m3: 0x..4038, fnuz: 0x..4840)\n", hb[0]);
    printf("cvt_pk_fp8(0.3, -448) = 0x%08x\n", hb[1]);
    printf("cvt_pk_fp8(1000, 0.001) = 0x%08x  (saturate 0x7e = 448?)\n", hb[2]);
    printf("cvt_scalef32_pk_fp8(8.0, 1.0, scale 4.0) = 0x%08x  (2.0 = 0x40 -> divides; 32 = 0x60 -> multiplies)\n", hb[3]);

    // ---- MFMA layout hypothesis H1: lane l holds A[row l&15][k = 32*(l>>4) + j], B[k = 32*(l>>4)+j][col l&15], byte j of 32
    int hA[64 * 8], hB[64 * 8], hSA[64], hSB[64];
    float Aref[16][128], Bref[128][16];
    srand(1);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) { Aref[i][k] = rand() % 4; Bref[k][i] = rand() % 4; }
    unsigned char* pa = (unsigned char*)hA; unsigned char* pb = (unsigned char*)hB;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            const int k = 32 * (l >> 4) + j;
            pa[l * 32 + j] = f8((int)Aref[l & 15][k]);
            pb[l * 32 + j] = f8((int)Bref[k][l & 15]);
        }
    int *dA, *dB, *dSA, *dSB; float* dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dSA, 256); hipMalloc(&dSB, 256); hipMalloc(&dD, 1024);
    float hD[256];
    for (int test = 0; test < 3; ++test) {
        // test 0: all scales 127 (1.0).  test 1: per-lane A scales 127 + (l>>4) (block-dependent).  test 2: per-lane A scale by row, B by block
        for (int l = 0; l < 64; ++l) {
            hSA[l] = 127; hSB[l] = 127;
            if (test == 1) hSA[l] = 127 + (l >> 4);
            if (test == 2) { hSA[l] = 127 + ((l & 15) % 3); hSB[l] = 127 - (l >> 4); }
        }
        hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        hipMemcpy(dSA, hSA, 256, hipMemcpyHostToDevice); hipMemcpy(dSB, hSB, 256, hipMemcpyHostToDevice);
        k_mfma<<<1, 64>>>(dA, dB, dSA, dSB, dD);
        hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int col = l & 15, row = (l >> 4) * 4 + r;
                double ref = 0;
                for (int blk = 0; blk < 4; ++blk) {
                    double s = 0;
                    for (int j = 0; j < 32; ++j) s += Aref[row][32 * blk + j] * Bref[32 * blk + j][col];
                    const int sa = hSA[row + 16 * blk], sb = hSB[col + 16 * blk];
                    ref += s * ldexp(1.0, sa - 127) * ldexp(1.0, sb - 127);
                }
                if (fabs(ref - hD[l * 4 + r]) > 1e-3) { if (bad < 4) printf("  mismatch test %d row %d col %d: got %g want %g\n", test, row, col, hD[l * 4 + r], ref); ++bad; }
            }
        printf("H1 layout, scale test %d: %s (%d mismatches)\n", test, bad ? "FAIL" : "ok", bad);
    }
    // ---- 32x32x64: hypothesis lane l holds A[row l&31][k(h = l>>5, j)], B[k(h, j)][col l&31] with the same k(h, j) on both sides;
    // C/D: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    {
        float A2[32][64], B2[64][32];
        for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) { A2[i][k] = rand() % 4; B2[k][i] = rand() % 4; }
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 32; ++j) {
                const int k = 32 * (l >> 5) + j;
                pa[l * 32 + j] = f8((int)A2[l & 31][k]);
                pb[l * 32 + j] = f8((int)B2[k][l & 31]);
            }
        float* dD2; hipMalloc(&dD2, 64 * 16 * 4);
        float hD2[64 * 16];
        hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        k_mfma32<<<1, 64>>>(dA, dB, dD2);
        hipMemcpy(hD2, dD2, sizeof(hD2), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                double ref = 0;
                for (int k = 0; k < 64; ++k) ref += A2[row][k] * B2[k][col];
                ref *= 0.5;
                if (fabs(ref - hD2[l * 16 + r]) > 1e-3) { if (bad < 4) printf("  32x32x64 mismatch row %d col %d: got %g want %g\n", row, col, hD2[l * 16 + r], ref); ++bad; }
            }
        printf("32x32x64 fp8 layout (uniform scales, B scale 2^-1): %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    }
    return 0;
    // ---- which lane group's scale acts on the K-block held by lane group b?
    for (int which = 0; which < 2; ++which)
        for (int b = 0; b < 4; ++b) {
            for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
                pa[l * 32 + j] = (which == 0 && (l >> 4) != b) ? 0x00 : 0x38;
                pb[l * 32 + j] = (which == 1 && (l >> 4) != b) ? 0x00 : 0x38;
            }
            hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
            for (int q = 0; q < 4; ++q) {
                for (int l = 0; l < 64; ++l) { hSA[l] = 127; hSB[l] = 127; }
                for (int l = 16 * q; l < 16 * q + 16; ++l) (which ? hSB : hSA)[l] = 128;
                hipMemcpy(dSA, hSA, 256, hipMemcpyHostToDevice); hipMemcpy(dSB, hSB, 256, hipMemcpyHostToDevice);
                k_mfma<<<1, 64>>>(dA, dB, dSA, dSB, dD);
                hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
                printf("%s data only in lane group %d, scale x2 in lane group %d: D[0][0] = %g\n", which ? "B" : "A", b, q, hD[0]);
            }
        }
    return 0;
}
