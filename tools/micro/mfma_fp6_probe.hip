// Probe: operand packing of v_mfma_scale_f32_16x16x128_f8f6f4 with fp6 (e2m3) operands (cbsz = blgp = 2): hypothesis
// "element j of a lane = bits [6j, 6j+6) of its 24-byte little-endian fragment, K index as for fp8 (k = 64*(j>>4) + 16*g + (j&15))".
// Exact small values: e2m3 encodes v in {0, 0.125, ..., 0.875 (subnormal), 1, 1.125, ... 7.5}: code = round(v * 8) for v < 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int i8v;
typedef __attribute__((ext_vector_type(4))) float f4;

__global__ void k(const int* A, const int* B, float* D) {
    const int l = threadIdx.x;
    i8v a, b;
    for (int i = 0; i < 8; ++i) { a[i] = A[l * 8 + i]; b[i] = B[l * 8 + i]; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, 127, 0, 127);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

static unsigned enc(float v) {  // e2m3, bias 1: 0 ee mmm
    if (v == 0) return 0;
    int e; float m = frexpf(v, &e);  // v = m * 2^e, m in [0.5, 1)
    float x = v;
    if (x < 1.0f) return (unsigned)lrintf(x * 8.0f);            // subnormal: 0.mmm
    int ee = (int)floorf(log2f(x));                             // 0..2
    return (unsigned)(((ee + 1) << 3) | (int)lrintf((x / exp2f((float)ee) - 1.0f) * 8.0f));
}

int main() {
    const float vals[4] = {0.0f, 0.5f, 1.0f, 1.5f};
    float Ar[16][128], Br[128][16];
    srand(3);
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 128; ++kk) { Ar[i][kk] = vals[rand() % 4]; Br[kk][i] = vals[rand() % 4]; }
    for (int hyp = 0; hyp < 2; ++hyp) {
        unsigned char pa[64 * 32], pb[64 * 32];
        memset(pa, 0, sizeof(pa)); memset(pb, 0, sizeof(pb));
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 32; ++j) {
                const int g = l >> 4;
                const int kk = hyp == 0 ? (64 * (j >> 4) + 16 * g + (j & 15)) : (32 * g + j);
                const unsigned ca = enc(Ar[l & 15][kk]), cb = enc(Br[kk][l & 15]);
                const int bit = 6 * j;
                for (int t = 0; t < 6; ++t) {
                    if ((ca >> t) & 1) pa[l * 32 + ((bit + t) >> 3)] |= (unsigned char)(1u << ((bit + t) & 7));
                    if ((cb >> t) & 1) pb[l * 32 + ((bit + t) >> 3)] |= (unsigned char)(1u << ((bit + t) & 7));
                }
            }
        int *dA, *dB; float* dD;
        hipMalloc(&dA, sizeof(pa)); hipMalloc(&dB, sizeof(pb)); hipMalloc(&dD, 1024);
        hipMemcpy(dA, pa, sizeof(pa), hipMemcpyHostToDevice); hipMemcpy(dB, pb, sizeof(pb), hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dD);
        float hD[256];
        hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int col = l & 15, row = (l >> 4) * 4 + r;
                double ref = 0;
                for (int kk = 0; kk < 128; ++kk) ref += Ar[row][kk] * Br[kk][col];
                if (fabs(ref - hD[l * 4 + r]) > 1e-3) { if (bad < 3) printf("  hyp %d mismatch (%d,%d): got %g want %g\n", hyp, row, col, hD[l * 4 + r], ref); ++bad; }
            }
        printf("fp6 e2m3, 6-bit little-endian packing, K map %s: %s (%d mismatches)\n", hyp == 0 ? "interleaved (as fp8)" : "k = 32 g + j", bad ? "FAIL" : "ok", bad);
    }
    return 0;
}
