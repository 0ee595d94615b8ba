// Constructive micro-benchmark of the conv kernel's MFMA phase (bf16x3, 16x16x32): LDS fragment reads with the kernel's
// lookahead + MFMAs, 8 waves per workgroup, one workgroup per CU, random data.  MODE 0: MFMAs on static registers, no LDS reads;
// 1: reads issued but not consumed; 2: reads feed the MFMAs (as in the kernel); 3: as 2 plus 4 global (L2-resident) weight
// loads per tap feeding the A operands; 4: as 3 plus a workgroup barrier every 5 taps.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;
union Frag { bf8 v; uint4 u; };

__device__ inline unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const uint4* __restrict__ wts, float* out, int taps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // fill 2 planes x 144 rows x 64 B with random bf16 bit patterns of moderate exponent
    for (int i = tid; i < 2 * 144 * 16; i += 512) {
        unsigned a = hashu(i * 7 + blockIdx.x), b = hashu(i * 13 + 5);
        uint4 v = make_uint4((a & 0x807f807fu) | 0x3f003f00u, (b & 0x807f807fu) | 0x3e803e80u, (a * 3 & 0x807f807fu) | 0x3f003f00u,
                             (b * 5 & 0x807f807fu) | 0x3e003e00u);
        reinterpret_cast<uint4*>(lds)[i] = v;
    }
    __syncthreads();
    f4 acc[2][8];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f4{0, 0, 0, 0};
    Frag ah[2], al[2];
    for (int c = 0; c < 2; ++c) {
        ah[c].u = wts[(c * 2 + 0) * 64 + lane];
        al[c].u = wts[(c * 2 + 1) * 64 + lane];
    }
    const int kq = lane >> 4;
    const unsigned char* hi_plane = lds;
    const unsigned char* lo_plane = lds + 144 * 64;
    const uint4* wbase = wts + lane;
    for (int tap = 0; tap < taps; ++tap) {
        const int k = tap % 5;
        const int rowk = (lane & 15) + k;
        const int b0 = rowk * 64 + ((kq ^ (((rowk >> 2) & 1) << 1)) << 4);
        Frag bh[3], bl[3];
        if (MODE >= 1) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bh[t].u = *reinterpret_cast<const uint4*>(hi_plane + b0 + t * 1024);
                bl[t].u = *reinterpret_cast<const uint4*>(lo_plane + b0 + t * 1024);
            }
        }
#pragma unroll
        for (int tb = 0; tb < 8; ++tb) {
            if (MODE >= 1 && tb + 2 < 8) {
                bh[(tb + 2) % 3].u = *reinterpret_cast<const uint4*>(hi_plane + b0 + (tb + 2) * 1024);
                bl[(tb + 2) % 3].u = *reinterpret_cast<const uint4*>(lo_plane + b0 + (tb + 2) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf8 xh = (MODE >= 2) ? bh[tb % 3].v : ah[1 - c].v;
                const bf8 xl = (MODE >= 2) ? bl[tb % 3].v : al[1 - c].v;
                acc[c][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[c].v, xh, acc[c][tb], 0, 0, 0);
                acc[c][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[c].v, xl, acc[c][tb], 0, 0, 0);
                acc[c][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[c].v, xh, acc[c][tb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1) {
#pragma unroll
            for (int t = 0; t < 3; ++t) asm volatile("" ::"v"(bh[t].u.x), "v"(bh[t].u.w), "v"(bl[t].u.x), "v"(bl[t].u.w));
        }
        if (MODE >= 3) {
            const uint4* wp = wbase + (size_t)((tap + 2) & 15) * 256;
#pragma unroll
            for (int c = 0; c < 2; ++c) { ah[c].u = wp[(c * 2 + 0) * 64]; al[c].u = wp[(c * 2 + 1) * 64]; }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE >= 4 && k == 4) __syncthreads();
    }
    f4 s = f4{0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j];
    out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}

template <int MODE>
void run(const uint4* wts, float* out, int taps, int nb) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 144 * 64);
    k<MODE><<<nb, 512, 2 * 144 * 64>>>(wts, out, 20);
    hipEventRecord(e0);
    k<MODE><<<nb, 512, 2 * 144 * 64>>>(wts, out, taps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 16 * 16 * 32 * 48.0 * taps * nb * 8;
    printf("mode %d, %d WGs, %d taps: %.3f ms  %.0f TFLOP/s executed\n", MODE, nb, taps, ms, fl / ms * 1e-9);
}

int main() {
    uint4* wts; float* out;
    hipMalloc(&wts, 16 * 256 * 16 + 4096);
    hipMalloc(&out, 1024 * 512 * 4);
    unsigned* h = (unsigned*)malloc(16 * 256 * 16 + 4096);
    for (int i = 0; i < (16 * 256 * 16 + 4096) / 4; ++i) { unsigned a = i * 2654435761u; a ^= a >> 13; h[i] = (a & 0x807f807fu) | 0x3e803e80u; }
    hipMemcpy(wts, h, 16 * 256 * 16 + 4096, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(wts, out, 40000, 256); run<1>(wts, out, 40000, 256); run<2>(wts, out, 40000, 256); run<3>(wts, out, 40000, 256); run<4>(wts, out, 40000, 256);
    }
    run<4>(wts, out, 80, 512);   // the conv launch's own size: 2 rounds of 80 taps
    run<4>(wts, out, 80, 512);
    return 0;
}
