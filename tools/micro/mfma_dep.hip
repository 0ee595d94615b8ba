// Micro-benchmark: v_mfma_f32_16x16x32_bf16 issue rate against the distance between dependent (same accumulator) MFMAs.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_dep.hip -o gpurun_out/mfma_dep && gpurun_out/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;

template <int DIST, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void k(float* out, int iters) {
    f4 acc[DIST];
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    for (int d = 0; d < DIST; ++d) acc[d] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 48 / DIST; ++r)
#pragma unroll
            for (int d = 0; d < DIST; ++d) acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[d], 0, 0, 0);
    }
    f4 s = acc[0];
    for (int d = 1; d < DIST; ++d) s += acc[d];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// the conv kernel's MFMA phase without anything else: MT x NT accumulators, 3 products each, operands held in registers
__device__ inline float rnd(unsigned x) {  // uniform in (-1, 1), full-entropy mantissas (switching activity like real data)
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(int)x * (1.0f / 2147483648.0f);
}

template <int MT, int NT, int WAVES, bool RANDOM = false>
__global__ __launch_bounds__(64 * WAVES, 2) void kconv(float* out, int iters) {
    f4 acc[MT][NT];
    bf8 ah[MT], al[MT], bh[NT], bl[NT];
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int m = 0; m < MT; ++m)
        for (int i = 0; i < 8; ++i) {
            ah[m][i] = (__bf16)(RANDOM ? rnd(tid * 64 + m * 8 + i) : threadIdx.x * 0.001f + i + m);
            al[m][i] = (__bf16)(RANDOM ? rnd(tid * 64 + m * 8 + i + 7777) * 0.004f : i * 0.25f + m);
        }
    for (int n = 0; n < NT; ++n)
        for (int i = 0; i < 8; ++i) {
            bh[n][i] = (__bf16)(RANDOM ? rnd(tid * 128 + n * 8 + i + 99999) : threadIdx.x * 0.002f + i + n);
            bl[n][i] = (__bf16)(RANDOM ? rnd(tid * 128 + n * 8 + i + 5555) * 0.004f : i * 0.125f + n);
        }
    for (int m = 0; m < MT; ++m)
        for (int n = 0; n < NT; ++n) acc[m][n] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
        }
        // keep the operands live and changing so nothing is hoisted
        if (!RANDOM) ah[0][0] = (__bf16)((float)ah[0][0] + 1.0f);
        else asm volatile("" : "+v"(ah[0]));
    }
    f4 s = f4{0, 0, 0, 0};
    for (int m = 0; m < MT; ++m)
        for (int n = 0; n < NT; ++n) s += acc[m][n];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int MT, int NT, int WAVES, bool RANDOM = false>
void runconv(int blocks_per_cu, int iters) {
    float* out;
    const int nb = 256 * blocks_per_cu;
    hipMalloc(&out, sizeof(float) * nb * 64 * WAVES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kconv<MT, NT, WAVES, RANDOM><<<nb, 64 * WAVES>>>(out, 10);
    hipEventRecord(e0);
    kconv<MT, NT, WAVES, RANDOM><<<nb, 64 * WAVES>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 32 * (3.0 * MT * NT) * (double)iters * nb * WAVES;
    printf("%s conv-pattern MT %d NT %d waves/WG %d WG/CU %d iters %d: %.1f TFLOP/s (%.3f ms)\n", RANDOM ? "random" : "const ", MT, NT, WAVES, blocks_per_cu, iters,
           flops / ms * 1e-9, ms);
    hipFree(out);
}

template <int DIST, int WAVES>
void run(int blocks_per_cu, int iters = 2000) {
    float* out;
    const int nb = 256 * blocks_per_cu;
    hipMalloc(&out, sizeof(float) * nb * 64 * WAVES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<DIST, WAVES><<<nb, 64 * WAVES>>>(out, 10);
    hipEventRecord(e0);
    k<DIST, WAVES><<<nb, 64 * WAVES>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 32 * (48 / DIST * DIST) * (double)iters * nb * WAVES;
    printf("dist %2d waves/WG %d WG/CU %d : %.1f TFLOP/s (%.3f ms)\n", DIST, WAVES, blocks_per_cu, flops / ms * 1e-9, ms);
    hipFree(out);
}

int main() {
    run<1, 4>(1); run<2, 4>(1); run<3, 4>(1); run<4, 4>(1); run<6, 4>(1); run<8, 4>(1); run<12, 4>(1);
    run<1, 4>(2); run<2, 4>(2); run<4, 4>(2); run<8, 4>(2);
    run<2, 8>(1); run<4, 8>(1);
    // sustained (hundreds of ms): does the clock hold?
    run<4, 8>(1, 200000); run<4, 8>(2, 200000);
    runconv<1, 8, 8>(1, 4000); runconv<1, 8, 8>(2, 4000); runconv<2, 8, 4>(1, 4000); runconv<2, 8, 4>(2, 4000);
    runconv<1, 8, 8>(2, 400000);
    runconv<2, 8, 8, true>(1, 4000); runconv<2, 8, 8, true>(1, 400000); runconv<2, 8, 8, true>(1, 400000);
    runconv<2, 8, 8, false>(1, 400000);
    return 0;
}
