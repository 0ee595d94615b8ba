// How fast does ONE wave per SIMD issue the MFMA stream of conv1d_w4.hip (4 x fp16, 4 x fp16, 4 x block-scaled fp6 per step, operands
// in registers, nothing else in the loop)?  Prints shader cycles per MFMA for: A operands in AccVGPRs / in VGPRs, with / without the
// scaled MFMAs, 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 w4_mfma_rate.hip -o w4_mfma_rate && ./w4_mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));

template <int MODE>   // 0: A in AGPR, all 12; 1: A in VGPR, all 12; 2: A in AGPR, fp16 only (8 per step); 3: A in AGPR, fp6 only (4 per step)
__global__ __launch_bounds__(256, 1) void k(const float* in, float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    f16x8 h0[4], h1[4]; u32x6 c6[4]; int sc[4];
    const float s = in[threadIdx.x];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 8; ++j) { h0[i][j] = (_Float16)(s * (i + j + 1)); h1[i][j] = (_Float16)(s * (i * j + 2)); }
        for (int j = 0; j < 6; ++j) c6[i][j] = __float_as_uint(s * (i + 3 * j)) * 2654435761u;
        sc[i] = 127;
    }
    f16x8 b0 = h0[1], b1 = h1[2]; u32x6 bc = c6[3]; int sb = 115;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tb = 0; tb < 8; ++tb) {
            f32x4 &a0 = acc[0][tb], &a1 = acc[1][tb], &a2 = acc[2][tb], &a3 = acc[3][tb];
            if (MODE == 0 || MODE == 2) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %8, %0\n\tv_mfma_f32_16x16x32_f16 %1, %5, %8, %1\n\t"
                             "v_mfma_f32_16x16x32_f16 %2, %6, %8, %2\n\tv_mfma_f32_16x16x32_f16 %3, %7, %8, %3"
                             : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3) : "a"(h0[0]), "a"(h0[1]), "a"(h0[2]), "a"(h0[3]), "v"(b0));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %8, %0\n\tv_mfma_f32_16x16x32_f16 %1, %5, %8, %1\n\t"
                             "v_mfma_f32_16x16x32_f16 %2, %6, %8, %2\n\tv_mfma_f32_16x16x32_f16 %3, %7, %8, %3"
                             : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3) : "a"(h1[0]), "a"(h1[1]), "a"(h1[2]), "a"(h1[3]), "v"(b1));
            } else if (MODE == 1) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %8, %0\n\tv_mfma_f32_16x16x32_f16 %1, %5, %8, %1\n\t"
                             "v_mfma_f32_16x16x32_f16 %2, %6, %8, %2\n\tv_mfma_f32_16x16x32_f16 %3, %7, %8, %3"
                             : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3) : "v"(h0[0]), "v"(h0[1]), "v"(h0[2]), "v"(h0[3]), "v"(b0));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %8, %0\n\tv_mfma_f32_16x16x32_f16 %1, %5, %8, %1\n\t"
                             "v_mfma_f32_16x16x32_f16 %2, %6, %8, %2\n\tv_mfma_f32_16x16x32_f16 %3, %7, %8, %3"
                             : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3) : "v"(h1[0]), "v"(h1[1]), "v"(h1[2]), "v"(h1[3]), "v"(b1));
            }
            if (MODE == 0 || MODE == 3) {
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %4, %8, %0, %9, %13 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
                             "v_mfma_scale_f32_16x16x128_f8f6f4 %1, %5, %8, %1, %10, %13 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
                             "v_mfma_scale_f32_16x16x128_f8f6f4 %2, %6, %8, %2, %11, %13 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
                             "v_mfma_scale_f32_16x16x128_f8f6f4 %3, %7, %8, %3, %12, %13 op_sel_hi:[0,0,0] cbsz:2 blgp:2"
                             : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3)
                             : "a"(c6[0]), "a"(c6[1]), "a"(c6[2]), "a"(c6[3]), "v"(bc), "v"(sc[0]), "v"(sc[1]), "v"(sc[2]), "v"(sc[3]), "v"(sb));
            } else if (MODE == 1) {
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %4, %8, %0, %9, %13 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
                             "v_mfma_scale_f32_16x16x128_f8f6f4 %1, %5, %8, %1, %10, %13 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
                             "v_mfma_scale_f32_16x16x128_f8f6f4 %2, %6, %8, %2, %11, %13 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
                             "v_mfma_scale_f32_16x16x128_f8f6f4 %3, %7, %8, %3, %12, %13 op_sel_hi:[0,0,0] cbsz:2 blgp:2"
                             : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3)
                             : "v"(c6[0]), "v"(c6[1]), "v"(c6[2]), "v"(c6[3]), "v"(bc), "v"(sc[0]), "v"(sc[1]), "v"(sc[2]), "v"(sc[3]), "v"(sb));
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) r += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float *in, *out; unsigned long long* cyc;
    hipMalloc(&in, 4096); hipMalloc(&out, 4 * 512 * 512); hipMalloc(&cyc, 8 * 512);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (float)((i * 7919) % 997) - 0.4f;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    const int iters = 200;
    auto run = [&](auto kern, int nthreads, int mfma_per_step, const char* name) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(nthreads), 0, 0, in, out, cyc, iters); hipDeviceSynchronize(); }
        std::vector<unsigned long long> c(256);
        hipMemcpy(c.data(), cyc, 8 * 256, hipMemcpyDeviceToHost);
        std::sort(c.begin(), c.end());
        const double per = (double)c[128] / ((double)iters * 8 * mfma_per_step);
        printf("%-46s %d waves/SIMD: %.2f cycles per MFMA per wave (%.2f per SIMD)\n", name, nthreads / 256, per, per / (nthreads / 256));
    };
    run(k<0>, 256, 12, "A in AccVGPRs, 8 fp16 + 4 fp6-scaled per step");
    run(k<1>, 256, 12, "A in VGPRs,    8 fp16 + 4 fp6-scaled per step");
    run(k<2>, 256, 8, "A in AccVGPRs, fp16 only");
    run(k<3>, 256, 4, "A in AccVGPRs, fp6-scaled only");
    return 0;
}
