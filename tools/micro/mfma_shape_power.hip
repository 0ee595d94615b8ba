// At the power limit, does the MFMA SHAPE matter?  The same matrix work per step (98,304 FLOP per wave) and the same LDS operand traffic
// (4 ds_read_b128 per step) issued as six v_mfma_f32_16x16x32_f16 or as three v_mfma_f32_32x32x16_f16, eight waves per CU (two per SIMD,
// like the conv kernel), every CU busy, random fp16 operands.  Prints wall time per launch, TFLOP/s and the in-kernel clock
// (s_memtime / s_memrealtime) after a warm-up of back-to-back launches; with and without the LDS reads.
//   hipcc --offload-arch=gfx950 -O3 mfma_shape_power.hip -o mfma_shape_power && ./mfma_shape_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }
// two random fp16 in [-2, 2) (exponent field 12..15, random sign and mantissa): finite, many toggling bits
__device__ __forceinline__ unsigned rnd_h2(unsigned s) {
    const unsigned r = mix(s);
    const unsigned lo = (r & 0x83FFu) | ((12u + ((r >> 10) & 3u)) << 10);
    const unsigned r2 = mix(r);
    const unsigned hi = (r2 & 0x83FFu) | ((12u + ((r2 >> 10) & 3u)) << 10);
    return lo | (hi << 16);
}

template <int SHAPE, bool LDSR>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* stamps, int iters) {
    extern __shared__ uint4 lds[];   // 64 KB
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 512)
        lds[i] = make_uint4(rnd_h2(i * 4 + blockIdx.x), rnd_h2(i * 4 + 1), rnd_h2(i * 4 + 2), rnd_h2(i * 4 + 3));
    __syncthreads();
    uint4 a_u[4];
    for (int j = 0; j < 4; ++j) a_u[j] = make_uint4(rnd_h2(tid + 7 * j), rnd_h2(tid + 11 * j + 1), rnd_h2(tid + 13 * j + 2), rnd_h2(tid + 17 * j + 3));
    f32x4 acc16[2][4];
    f32x16 acc32[2];
    for (int i = 0; i < 2; ++i) { for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0, 0, 0, 0}; for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f; }
    uint4 f[4] = {lds[lane], lds[lane + 64], lds[lane + 128], lds[lane + 192]};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tb = 0; tb < 8; ++tb) {
            if (LDSR) {
                const int base = ((it & 1) * 2048 + tb * 256 + lane);   // 16 bytes per lane, lane-contiguous: conflict-free
                f[0] = lds[base]; f[1] = lds[base + 64]; f[2] = lds[base + 128]; f[3] = lds[base + 192];
            }
            const f16x8 b0 = __builtin_bit_cast(f16x8, f[0]), b1 = __builtin_bit_cast(f16x8, f[1]);
            const f16x8 b2 = __builtin_bit_cast(f16x8, f[2]), b3 = __builtin_bit_cast(f16x8, f[3]);
            const f16x8 a0 = __builtin_bit_cast(f16x8, a_u[0]), a1 = __builtin_bit_cast(f16x8, a_u[1]);
            const f16x8 a2 = __builtin_bit_cast(f16x8, a_u[2]), a3 = __builtin_bit_cast(f16x8, a_u[3]);
            if (SHAPE == 0) {
                acc16[0][tb & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc16[0][tb & 3], 0, 0, 0);
                acc16[1][tb & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc16[1][tb & 3], 0, 0, 0);
                acc16[0][tb & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, acc16[0][tb & 3], 0, 0, 0);
                acc16[1][tb & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3, b1, acc16[1][tb & 3], 0, 0, 0);
                acc16[0][tb & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b2, acc16[0][tb & 3], 0, 0, 0);
                acc16[1][tb & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b3, acc16[1][tb & 3], 0, 0, 0);
            } else {
                acc32[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc32[0], 0, 0, 0);
                acc32[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc32[1], 0, 0, 0);
                acc32[tb & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, (tb & 1) ? b3 : b2, acc32[tb & 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) { for (int j = 0; j < 4; ++j) s += acc16[i][j][0] + acc16[i][j][3]; for (int j = 0; j < 16; ++j) s += acc32[i][j]; }
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, bool LDSR>
void run(const char* name, int nwg, int iters) {
    float* out; unsigned long long* st;
    hipMalloc(&out, sizeof(float) * nwg * 512);
    hipMalloc(&st, sizeof(unsigned long long) * 2 * nwg);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE, LDSR>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL((k<SHAPE, LDSR>), dim3(nwg), dim3(512), 65536, 0, out, st, iters);   // ~0.6 s warm-up
    hipEventRecord(e0);
    const int n = 50;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k<SHAPE, LDSR>), dim3(nwg), dim3(512), 65536, 0, out, st, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * nwg);
    hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * nwg, hipMemcpyDeviceToHost);
    double cyc = 0, ticks = 0;
    for (int i = 0; i < nwg; ++i) { cyc += (double)h[2 * i]; ticks += (double)h[2 * i + 1]; }
    const double flop = (double)nwg * 8 * iters * 8 * 98304.0;
    printf("%-34s %4d workgroups: %8.1f us per launch, %7.1f TFLOP/s, clock %.3f GHz, %.1f cycles per step and wave\n", name, nwg,
           1e3 * ms / n, flop / (ms / n * 1e-3) / 1e12, cyc / (ticks * 10.0), cyc / nwg / ((double)iters * 8));
    hipFree(out); hipFree(st);
}

// Operand-traffic mixes per 48 MFMAs of the conv's MFMA stream (16x16x32 f16), one "tap" per outer step:
//   MIX 0 (the conv today: wave tile 32 channels x 128 positions): 32 ds_read_b128 + 8 global_load_dwordx4 (weights, L2-resident)
//   MIX 1 (wave tile 64 channels x 64 positions, activation fragments of 4 t-blocks held in registers): 16 + 16
//   MIX 2: 16 + 8 (what MIX 1 would be if the two position-waves' weight loads were free)
template <int MIX>
__global__ __launch_bounds__(512, 2) void kmix(const uint4* __restrict__ wts, float* out, int iters, int wmask) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 512)
        lds[i] = make_uint4(rnd_h2(i * 4 + blockIdx.x), rnd_h2(i * 4 + 1), rnd_h2(i * 4 + 2), rnd_h2(i * 4 + 3));
    __syncthreads();
    constexpr int NWT = (MIX == 1) ? 16 : 8;      // weight fragments per tap
    constexpr int NLD = (MIX == 0) ? 32 : 16;     // LDS fragments per tap
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = f32x4{0, 0, 0, 0};
    uint4 w[2][NWT];
    const uint4* wp = wts + (size_t)(wave * 64 + lane);
    for (int q = 0; q < NWT; ++q) { w[0][q] = wp[q * 512]; w[1][q] = wp[(q + NWT) * 512]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint4 (&wc)[NWT] = w[half];
            if constexpr (MIX == 0) {
#pragma unroll
                for (int tb = 0; tb < 8; ++tb) {
                    const int base = (((it + half) & 1) * 2048 + tb * 256 + lane);
                    const uint4 f0 = lds[base], f1 = lds[base + 64], f2 = lds[base + 128], f3 = lds[base + 192];
                    const f16x8 b0 = __builtin_bit_cast(f16x8, f0), b1 = __builtin_bit_cast(f16x8, f1);
                    const f16x8 b2 = __builtin_bit_cast(f16x8, f2), b3 = __builtin_bit_cast(f16x8, f3);
                    acc[2 * tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[0]), b0, acc[2 * tb], 0, 0, 0);
                    acc[2 * tb + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[4]), b0, acc[2 * tb + 1], 0, 0, 0);
                    acc[2 * tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[1]), b1, acc[2 * tb], 0, 0, 0);
                    acc[2 * tb + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[5]), b1, acc[2 * tb + 1], 0, 0, 0);
                    acc[2 * tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[2]), b2, acc[2 * tb], 0, 0, 0);
                    acc[2 * tb + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[6]), b3, acc[2 * tb + 1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                uint4 fr[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) fr[q] = lds[((it + half) & 1) * 2048 + q * 64 + lane];
#pragma unroll
                for (int cp = 0; cp < 2; ++cp)   // channel-block pair
#pragma unroll
                    for (int tb = 0; tb < 4; ++tb) {
                        const f16x8 b0 = __builtin_bit_cast(f16x8, fr[4 * tb]), b1 = __builtin_bit_cast(f16x8, fr[4 * tb + 1]);
                        const f16x8 b2 = __builtin_bit_cast(f16x8, fr[4 * tb + 2]), b3 = __builtin_bit_cast(f16x8, fr[4 * tb + 3]);
                        const int wo = (MIX == 1) ? 8 * cp : 0;
                        const int a0 = 4 * cp + 2 * (tb & 1);   // (8 of the 16 accumulators per pair: same count of MFMAs as MIX 0)
                        acc[a0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[wo + 0]), b0, acc[a0], 0, 0, 0);
                        acc[a0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[wo + 4]), b0, acc[a0 + 1], 0, 0, 0);
                        acc[a0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[wo + 1]), b1, acc[a0], 0, 0, 0);
                        acc[a0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[wo + 5]), b1, acc[a0 + 1], 0, 0, 0);
                        acc[a0 + 8] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[wo + 2]), b2, acc[a0 + 8], 0, 0, 0);
                        acc[a0 + 9] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[wo + 6]), b3, acc[a0 + 9], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
            // refill this buffer with the weights of two taps ahead (a 720 KB L2-resident array walked cyclically)
            const int tap = (2 * it + half + 2) & wmask;
#pragma unroll
            for (int q = 0; q < NWT; ++q) wc[q] = wp[(size_t)(tap * NWT + q) * 512];
        }
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][2];
    out[blockIdx.x * 512 + tid] = s;
}

template <int MIX>
void runmix(const char* name, int nwg, int iters) {
    float* out; uint4* wts;
    const int ntap = 64;   // 64 taps x 16 fragments x 512 lanes x 16 B = 8 MB (L2-resident per XCD only in part; MALL-resident)
    hipMalloc(&out, sizeof(float) * nwg * 512);
    hipMalloc(&wts, (size_t)(ntap * 16 + 32) * 512 * sizeof(uint4));
    hipMemset(wts, 0x3c, (size_t)(ntap * 16 + 32) * 512 * sizeof(uint4));
    hipFuncSetAttribute(reinterpret_cast<const void*>(kmix<MIX>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL((kmix<MIX>), dim3(nwg), dim3(512), 65536, 0, wts, out, iters, 7);
    hipEventRecord(e0);
    const int n = 50;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((kmix<MIX>), dim3(nwg), dim3(512), 65536, 0, wts, out, iters, 7);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)nwg * 8 * iters * 2 * 48 * 16384.0;
    printf("%-60s %4d workgroups: %8.1f us per launch, %7.1f TFLOP/s\n", name, nwg, 1e3 * ms / n, flop / (ms / n * 1e-3) / 1e12);
    hipFree(out); hipFree(wts);
}

int main() {
    const int iters = 600;
    for (int nwg : {64, 256}) {
        run<0, true>("16x16x32 f16, 4 ds_read_b128/step", nwg, iters);
        run<1, true>("32x32x16 f16, 4 ds_read_b128/step", nwg, iters);
        run<0, false>("16x16x32 f16, no LDS reads", nwg, iters);
        run<1, false>("32x32x16 f16, no LDS reads", nwg, iters);
    }
    for (int nwg : {64, 256}) {
        runmix<0>("per 48 MFMAs: 32 LDS + 8 weight loads (the conv today)", nwg, 300);
        runmix<1>("per 48 MFMAs: 16 LDS + 16 weight loads (64 ch x 64 pos waves)", nwg, 300);
        runmix<2>("per 48 MFMAs: 16 LDS + 8 weight loads", nwg, 300);
    }
    return 0;
}
