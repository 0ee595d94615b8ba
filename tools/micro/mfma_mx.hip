// Micro-benchmark: per 64-deep K step of one 16x16 tile, (a) six bf16 MFMAs (bf16x3: hi.hi + hi.lo + lo.hi) against
// (b) two f16 MFMAs + one block-scaled fp8 MFMA (K = 128: both correction products) -- random operands, sustained.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(8))) int i8v;
typedef __attribute__((ext_vector_type(4))) float f4;

__device__ inline unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline float rnd(unsigned x) { return (float)(int)hashu(x) * (1.0f / 2147483648.0f); }

template <int MT, int NT, int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    f4 acc[MT][NT];
    for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) acc[m][n] = f4{0, 0, 0, 0};
    if (MODE == 0) {
        bf8 ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
        for (int s = 0; s < 2; ++s) {
            for (int m = 0; m < MT; ++m) for (int i = 0; i < 8; ++i) { ah[s][m][i] = (__bf16)rnd(tid * 977 + s * 64 + m * 8 + i); al[s][m][i] = (__bf16)(rnd(tid * 31 + s * 64 + m * 8 + i + 5) * 0.004f); }
            for (int n = 0; n < NT; ++n) for (int i = 0; i < 8; ++i) { bh[s][n][i] = (__bf16)rnd(tid * 13 + s * 64 + n * 8 + i + 99); bl[s][n][i] = (__bf16)(rnd(tid * 7 + s * 64 + n * 8 + i + 3) * 0.004f); }
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[s][m], bh[s][n], acc[m][n], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[s][m], bl[s][n], acc[m][n], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[s][m], bh[s][n], acc[m][n], 0, 0, 0);
                }
            asm volatile("" : "+v"(ah[0][0]));
        }
    } else {
        h8 ah[2][MT], bh[2][NT];
        i8v ac[MT], bc[NT];
        for (int s = 0; s < 2; ++s) {
            for (int m = 0; m < MT; ++m) for (int i = 0; i < 8; ++i) ah[s][m][i] = (_Float16)rnd(tid * 977 + s * 64 + m * 8 + i);
            for (int n = 0; n < NT; ++n) for (int i = 0; i < 8; ++i) bh[s][n][i] = (_Float16)rnd(tid * 13 + s * 64 + n * 8 + i + 99);
        }
        // fp8 bytes: random exponent/mantissa, avoid NaN (0x7f / 0xff)
        for (int m = 0; m < MT; ++m) for (int i = 0; i < 8; ++i) ac[m][i] = (int)(hashu(tid * 41 + m * 8 + i) & 0xbfbfbfbfu);
        for (int n = 0; n < NT; ++n) for (int i = 0; i < 8; ++i) bc[n][i] = (int)(hashu(tid * 43 + n * 8 + i + 7) & 0xbfbfbfbfu);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s][m], bh[s][n], acc[m][n], 0, 0, 0);
                if (MODE == 1) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ac[m], bc[n], acc[m][n], 0, 0, 0, 127, 0, 115);
                }
            }
            asm volatile("" : "+v"(ah[0][0]));
        }
    }
    f4 s = f4{0, 0, 0, 0};
    for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) s += acc[m][n];
    out[tid] = s[0] + s[1] + s[2] + s[3];
}

typedef __attribute__((ext_vector_type(16))) float f16v;
// 32x32 tiles: per 32-deep K step two f16 32x32x16 + one scaled fp8 32x32x64 (K = 64: both correction products of 32 channels)
template <int NT32, int MODE>
__global__ __launch_bounds__(512, 2) void k32(float* out, int iters) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    f16v acc[NT32];
    for (int n = 0; n < NT32; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    h8 ah[2], bh[2][NT32];
    i8v ac, bc[NT32];
    for (int s = 0; s < 2; ++s) {
        for (int i = 0; i < 8; ++i) ah[s][i] = (_Float16)rnd(tid * 977 + s * 64 + i);
        for (int n = 0; n < NT32; ++n) for (int i = 0; i < 8; ++i) bh[s][n][i] = (_Float16)rnd(tid * 13 + s * 64 + n * 8 + i + 99);
    }
    for (int i = 0; i < 8; ++i) ac[i] = (int)(hashu(tid * 41 + i) & 0xbfbfbfbfu);
    for (int n = 0; n < NT32; ++n) for (int i = 0; i < 8; ++i) bc[n][i] = (int)(hashu(tid * 43 + n * 8 + i + 7) & 0xbfbfbfbfu);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NT32; ++n) {
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s][n], acc[n], 0, 0, 0);
            if (MODE == 1) acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ac, bc[n], acc[n], 0, 0, 0, 127, 0, 115);
        }
        asm volatile("" : "+v"(ah[0]));
    }
    float s = 0;
    for (int n = 0; n < NT32; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[tid] = s;
}

template <int MODE>
void run32(int iters) {
    float* out;
    const int nb = 256;
    hipMalloc(&out, sizeof(float) * nb * 512);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k32<4, MODE><<<nb, 512>>>(out, 10);
    hipEventRecord(e0);
    k32<4, MODE><<<nb, 512>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double useful = 2.0 * 32 * 32 * 32 * 4.0 * (double)iters * nb * 8;
    printf("%s iters %d: %.3f ms, %.1f TFLOP/s algorithmic\n", MODE == 1 ? "32x32: f16 x2 + scaled fp8 K=64" : "32x32: f16 x2 only            ", iters, ms, useful / ms * 1e-9);
    hipFree(out);
}

template <int MODE>
void run(int iters) {
    float* out;
    const int nb = 256;
    hipMalloc(&out, sizeof(float) * nb * 512);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<2, 8, MODE><<<nb, 512>>>(out, 10);
    hipEventRecord(e0);
    k<2, 8, MODE><<<nb, 512>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double useful = 2.0 * 16 * 16 * 64 * (2.0 * 8) * (double)iters * nb * 8;  // algorithmic FLOP (K = 64 per step)
    printf("%s iters %d: %.3f ms, %.1f TFLOP/s algorithmic\n", MODE == 0 ? "bf16x3 (6 MFMA / 64 K)      " : MODE == 1 ? "f16 x2 + scaled fp8 K=128   " : "f16 x2 only (no correction) ", iters, ms, useful / ms * 1e-9);
    hipFree(out);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) { run<0>(100000); run<1>(100000); run32<1>(200000); run32<2>(200000); }
    return 0;
}
