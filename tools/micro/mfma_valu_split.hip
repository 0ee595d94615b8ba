// Micro-benchmark: two waves per SIMD, SPECIALISED -- waves 0-3 of the workgroup issue only MFMAs (48 v_mfma_f32_16x16x32_bf16 per
// iteration, 8 independent accumulators), waves 4-7 only VALU (NV v_fma_f32 per iteration); waves w and w + 4 share a SIMD.
// Compare with mfma_valu.hip, where every wave interleaves both: does the SIMD run one wave's MFMAs under the other's VALU?
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_split.hip -o gpurun_out/mfma_valu_split && gpurun_out/mfma_valu_split
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;

template <int MODE, int NV>   // MODE 0: both kinds of wave; 1: MFMA waves only (the others exit); 2: VALU waves only
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    float t = 0.f;
    if (wave < 4) {
        if (MODE == 2) return;
        f4 acc[8];
        bf8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
        for (int d = 0; d < 8; ++d) acc[d] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 48; ++r) acc[r & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[r & 7], 0, 0, 0);
        }
        f4 s = acc[0];
        for (int d = 1; d < 8; ++d) s += acc[d];
        t = s[0] + s[1] + s[2] + s[3];
    } else {
        if (MODE == 1) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < NV; ++r) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r & 7]) : "v"(1.0001f));
        }
        for (int i = 0; i < 8; ++i) t += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int MODE, int NV>
void run(const char* name, float* out) {
    const int iters = 2000, grid = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(grid), dim3(512), 0, 0, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(grid), dim3(512), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %7.3f us per iteration\n", name, ms * 1e3 / iters);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    run<1, 192>("MFMA waves alone (48 MFMA / iteration)", out);
    run<2, 192>("VALU waves alone (192 fma / iteration)", out);
    run<0, 192>("MFMA waves + VALU waves (192 fma) on the same SIMDs", out);
    run<2, 96>("VALU waves alone (96 fma / iteration)", out);
    run<0, 96>("MFMA waves + VALU waves (96 fma) on the same SIMDs", out);
    run<2, 384>("VALU waves alone (384 fma / iteration)", out);
    run<0, 384>("MFMA waves + VALU waves (384 fma) on the same SIMDs", out);
    return 0;
}
