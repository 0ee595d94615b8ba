// Micro-benchmark: do VALU instructions of a wave issue under its own (or a co-resident wave's) MFMAs on gfx950?
// Per loop iteration: 48 independent-accumulator v_mfma_f32_16x16x32_bf16 interleaved with NV plain VALU ops (v_fma_f32) and NE
// transcendental ops (v_exp_f32) on registers the MFMAs do not touch.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu.hip -o gpurun_out/mfma_valu && gpurun_out/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;

template <int NM, int NV, int NE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(float* out, int iters) {
    f4 acc[8];
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    for (int d = 0; d < 8; ++d) acc[d] = f4{0, 0, 0, 0};
    float v[8], e[4];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    for (int i = 0; i < 4; ++i) e[i] = threadIdx.x * 0.001f + i * 0.1f;
    for (int it = 0; it < iters; ++it) {
        constexpr int STEPS = NM > 0 ? NM : 48;
#pragma unroll
        for (int r = 0; r < STEPS; ++r) {
            if (NM > 0) acc[r & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[r & 7], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV / STEPS; ++q) {
                float& x = v[(r * (NV / STEPS) + q) & 7];
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(1.0001f));
            }
            if (NE > 0 && (r % (STEPS / (NE > STEPS ? STEPS : NE))) == 0) {
#pragma unroll
                for (int q = 0; q < (NE + STEPS - 1) / STEPS; ++q) {
                    float& x = e[(r + q) & 3];
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                }
            }
        }
    }
    f4 s = acc[0];
    for (int d = 1; d < 8; ++d) s += acc[d];
    float t = s[0] + s[1] + s[2] + s[3];
    for (int i = 0; i < 8; ++i) t += v[i];
    for (int i = 0; i < 4; ++i) t += e[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int NM, int NV, int NE, int WAVES>
void run(const char* name, float* out) {
    const int iters = 2000, grid = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV, NE, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV, NE, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // cycles per iteration per SIMD at a nominal 2.4 GHz (the printed us/iter is the robust number)
    printf("%-44s waves/SIMD %d: %7.3f us per iteration (%d MFMA, %d fma, %d exp per wave)\n", name, WAVES / 4, ms * 1e3 / iters, NM, NV, NE);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    run<48, 0, 0, 4>("MFMA only", out);
    run<0, 96, 0, 4>("96 fma only", out);
    run<0, 192, 0, 4>("192 fma only", out);
    run<0, 0, 48, 4>("48 exp only", out);
    run<48, 96, 0, 4>("MFMA + 96 fma", out);
    run<48, 192, 0, 4>("MFMA + 192 fma", out);
    run<48, 0, 48, 4>("MFMA + 48 exp", out);
    run<48, 96, 48, 4>("MFMA + 96 fma + 48 exp", out);
    run<48, 0, 0, 8>("MFMA only", out);
    run<0, 192, 0, 8>("192 fma only", out);
    run<48, 96, 0, 8>("MFMA + 96 fma", out);
    run<48, 192, 0, 8>("MFMA + 192 fma", out);
    run<48, 0, 48, 8>("MFMA + 48 exp", out);
    run<48, 96, 48, 8>("MFMA + 96 fma + 48 exp", out);
    return 0;
}
