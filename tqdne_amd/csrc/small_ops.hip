// Memory-bound side kernels of the tqdne hot path on gfx950: GroupNorm statistics finalisation, the
// NCW<->channels-last boundary convolutions (3-channel stem and head), embedding MLPs, EDM scalar maps,
// noise injection / loss, and the fp64 Heun state updates.  All are plain wave64 VALU kernels with
// 16-byte coalesced accesses; none is GEMM-shaped enough to be worth MFMA.
#include "common.hpp"
#include "gn_fold.hpp"
#include "../../include/tqdne_hip.h"

using namespace tq;

extern "C" int tq_abi_version(void) { return TQ_ABI_VERSION; }
extern "C" int tq_build_flags(void) {
#ifdef TQ_BUILD_EXPERIMENTS
    return TQ_BUILD_EXPERIMENTS_BIT;
#else
    return 0;
#endif
}

// =================================================================================================
// GroupNorm32 finalisation: per-channel partial (sum, sumsq) of up to two concatenated sources ->
// folded per-(b,c) scale/shift.  One workgroup per sample.  Sums are combined in fp64 so that
// var = E[x^2] - mean^2 does not cancel (the partials themselves are fp32 sums over <= 128 positions).
// =================================================================================================
namespace {
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ st0, int C0,
                                                          const float* __restrict__ st1, int C1, int T, int nslots0, int nslots1,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ gscale, float* __restrict__ gshift,
                                                          float* __restrict__ mean_rstd) {
    extern __shared__ double sh[];  // [C][2] channel sums, then [32][2] group mean/rstd
    gn_fold_sample<false, false>(sh, blockIdx.x, st0, C0, st1, C1, T, nslots0, nslots1, gamma, beta, gscale, gshift, mean_rstd);
}
}  // namespace

extern "C" int tq_gn_finalize(const float* stats0, int C0, const float* stats1, int C1, int B, int T, const float* gamma,
                              const float* beta, float* gscale, float* gshift, float* mean_rstd, int slot0, int slot1,
                              hipStream_t stream) {
    if (!stats0 || !gamma || !beta || !gscale || !gshift || (C1 > 0 && !stats1)) return TQ_ERR_ARG;
    const int C = C0 + C1;
    if (B <= 0 || T <= 0 || C <= 0 || C % GN_GROUPS || C > 3840) return TQ_ERR_SHAPE;  // (LDS: 16 C + 512 bytes <= 64 KB)
    if (slot0 == 0) slot0 = STAT_SLOT;
    if (slot1 == 0) slot1 = STAT_SLOT;
    if ((slot0 != STAT_SLOT && slot0 != 32) || (slot1 != STAT_SLOT && slot1 != 32)) return TQ_ERR_ARG;
    const int nslots0 = (T + slot0 - 1) / slot0, nslots1 = (T + slot1 - 1) / slot1;
    const size_t shbytes = (size_t)(2 * C + 2 * GN_GROUPS) * sizeof(double);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), shbytes, stream, stats0, C0, stats1, C1, T, nslots0, nslots1, gamma,
                       beta, gscale, gshift, mean_rstd);
    TQ_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================
// Stem: (B, C_in<=16, T) NCW -> conv k "same" -> (B, T, C_out) channels-last, + bias, + partial stats.
// One workgroup per (b, 128-position slot).  Thread = (4 output channels, 128/(256/(C_out/4)) positions).
// =================================================================================================
namespace {
template <int KT>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ y, float* __restrict__ stats, int C_in, int T,
                                                        int C_out, int nslots) {
    extern __shared__ float shm[];
    constexpr int PAD = KT / 2;
    constexpr int TW = STAT_SLOT + KT - 1;
    float* xs = shm;                        // [C_in][TW]
    float* ws = xs + C_in * TW;             // [KT][C_in][C_out]
    float* red = ws + KT * C_in * C_out;    // [nrow][C_out][2]
    const int slot = blockIdx.x % nslots;
    const int b = blockIdx.x / nslots;
    const int t0 = slot * STAT_SLOT;
    const float sc = in_scale ? in_scale[b] : 1.0f;
    for (int i = threadIdx.x; i < C_in * TW; i += 256) {
        const int c = i / TW, j = i % TW;
        const int t = t0 - PAD + j;
        xs[i] = (t >= 0 && t < T) ? x[((size_t)b * C_in + c) * T + t] * sc : 0.f;
    }
    for (int i = threadIdx.x; i < KT * C_in * C_out; i += 256) {
        const int co = i % C_out, r = i / C_out;
        const int ci = r % C_in, k = r / C_in;
        ws[i] = w[((size_t)co * C_in + ci) * KT + k];
    }
    __syncthreads();
    const int ngrp = C_out >> 2;            // groups of 4 output channels
    const int nrow = 256 / ngrp;            // positions processed per pass
    const int cg = threadIdx.x % ngrp;
    const int tr = threadIdx.x / ngrp;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4*>(bias + 4 * cg);
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    const int nkc = KT * C_in;
    if (tr < nrow && nkc <= 16) {
        // the usual stem (3 input channels, k = 5): the thread's 4 output channels' weights stay in registers and the (tap, channel)
        // offsets into the input tile in SGPRs -- per output float4 15 LDS reads + 60 FMAs instead of 30 + 60 (round 3; 30.8 -> 30.3 us at
        // B = 64, T = 4096: the launch is not bound by its instruction count)
        float4 wr[16];
        int xoff[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ii = i < nkc ? i : 0;
            wr[i] = *reinterpret_cast<const float4*>(ws + ii * C_out + 4 * cg);
            if (i >= nkc) wr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int k = ii / C_in, ci = ii % C_in;   // ws is [k][ci][co]
            xoff[i] = ci * TW + k;
        }
        for (int tl = tr; tl < STAT_SLOT; tl += nrow) {
            const int t = t0 + tl;
            if (t >= T) break;
            float4 a = bv;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i < nkc) {   // (uniform)
                    const float xv = xs[xoff[i] + tl];
                    a.x = fmaf(wr[i].x, xv, a.x); a.y = fmaf(wr[i].y, xv, a.y);
                    a.z = fmaf(wr[i].z, xv, a.z); a.w = fmaf(wr[i].w, xv, a.w);
                }
            }
            *reinterpret_cast<float4*>(y + ((size_t)b * T + t) * C_out + 4 * cg) = a;
            s1[0] += a.x; s1[1] += a.y; s1[2] += a.z; s1[3] += a.w;
            s2[0] += a.x * a.x; s2[1] += a.y * a.y; s2[2] += a.z * a.z; s2[3] += a.w * a.w;
        }
    } else if (tr < nrow) {
        for (int tl = tr; tl < STAT_SLOT; tl += nrow) {
            const int t = t0 + tl;
            if (t >= T) break;
            float4 a = bv;
            for (int k = 0; k < KT; ++k)
                for (int ci = 0; ci < C_in; ++ci) {
                    const float xv = xs[ci * TW + tl + k];
                    const float4 wv = *reinterpret_cast<const float4*>(ws + (k * C_in + ci) * C_out + 4 * cg);
                    a.x = fmaf(wv.x, xv, a.x); a.y = fmaf(wv.y, xv, a.y);
                    a.z = fmaf(wv.z, xv, a.z); a.w = fmaf(wv.w, xv, a.w);
                }
            *reinterpret_cast<float4*>(y + ((size_t)b * T + t) * C_out + 4 * cg) = a;
            s1[0] += a.x; s1[1] += a.y; s1[2] += a.z; s1[3] += a.w;
            s2[0] += a.x * a.x; s2[1] += a.y * a.y; s2[2] += a.z * a.z; s2[3] += a.w * a.w;
        }
    }
    if (stats) {
        if (tr < nrow) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[(tr * C_out + 4 * cg + j) * 2] = s1[j];
                red[(tr * C_out + 4 * cg + j) * 2 + 1] = s2[j];
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C_out; c += 256) {
            float a1 = 0.f, a2 = 0.f;
            for (int r = 0; r < nrow; ++r) { a1 += red[(r * C_out + c) * 2]; a2 += red[(r * C_out + c) * 2 + 1]; }
            float* st = stats + (((size_t)b * nslots + slot) * C_out + c) * 2;
            st[0] = a1; st[1] = a2;
        }
    }
}
}  // namespace

extern "C" int tq_stem_conv_fwd(const float* x, const float* in_scale, const float* w, const float* bias, float* y,
                                float* stats, int B, int C_in, int T, int C_out, int ktaps, hipStream_t stream) {
    if (!x || !w || !y) return TQ_ERR_ARG;
    if (B <= 0 || T <= 0 || C_in <= 0 || C_in > 16 || C_out < 4 || C_out % 4 || C_out > 1024 || 256 % (C_out / 4)) return TQ_ERR_SHAPE;
    const int nslots = (T + STAT_SLOT - 1) / STAT_SLOT;
    const int ngrp = C_out / 4, nrow = 256 / ngrp;
    const size_t sh = ((size_t)C_in * (STAT_SLOT + ktaps - 1) + (size_t)ktaps * C_in * C_out + (size_t)nrow * C_out * 2) * sizeof(float);
    if (sh > 160 * 1024) return TQ_ERR_SHAPE;
#define TQ_STEM(K)                                                                                          \
    {                                                                                                       \
        auto kern = stem_conv_kernel<K>;                                                                    \
        if (sh > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
        hipLaunchKernelGGL(kern, dim3(B * nslots), dim3(256), sh, stream, x, in_scale, w, bias, y, stats, C_in, T, C_out, nslots); \
    }
    if (ktaps == 5) TQ_STEM(5)
    else if (ktaps == 3) TQ_STEM(3)
    else if (ktaps == 1) TQ_STEM(1)
    else return TQ_ERR_SHAPE;
#undef TQ_STEM
    TQ_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================
// Head: GroupNorm+SiLU (folded) -> conv k "same" to C_out <= 16 -> NCW output with the EDM / consistency skip connection folded
// in (unet.py:355-357,398; edm.py:111-113).  HBM-bound: the (B, T, C_in) input is read once.
// One workgroup per (b, 128 output positions), "row stationary": four threads share one input row (position), each activates its
// quarter of the channels (16-byte loads, a quad reads 64 contiguous bytes per instruction) and forms that row's KT x C_out
// partial dot products against the weights (LDS, broadcast reads); the quad sums them with two cross-lane steps and parks the
// row's KT x C_out sums in LDS; after one barrier output (t, co) adds the KT entries of rows t .. t + KT - 1.
// (The first version staged a transposed [ci][t] tile; its output was found corrupted when an attention kernel ran on the same CU
// from another stream -- every launch of the plan is now also checked under that kind of concurrency, tests/test_concurrency.py.)
// =================================================================================================
namespace {
template <int KT, int MAXCO>
__global__ __launch_bounds__(256) void head_conv_kernel(const float* __restrict__ x, const float* __restrict__ gscale,
                                                        const float* __restrict__ gshift, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ c_out,
                                                        const float* __restrict__ c_skip, const float* __restrict__ skip_src,
                                                        float* __restrict__ y, int T, int C_in, int C_out, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float shm[];
    constexpr int PAD = KT / 2;
    // 128 INPUT rows per workgroup = two full passes of the 64 quads, 128 - (KT - 1) output positions.  (Round 3: with 128 outputs
    // the 132 rows took three passes, the third with 4 of 64 quads at work: 42.1 -> 37.5 us at B = 64, T = 4096, same box.)
    constexpr int TW = 128;
    constexpr int NOUT = TW - (KT - 1);
    constexpr int NP = KT * MAXCO;     // partial sums per input row
    constexpr int MAXV = 8;            // float4 per thread and row: C_in <= 128
    float* wl = shm;                   // [C_in][KT][MAXCO] weights, co fastest (16-byte broadcast reads)
    float* part = wl + C_in * NP;      // [TW][NP]
    for (int i = threadIdx.x; i < C_in * NP; i += 256) {
        const int co = i % MAXCO, r = i / MAXCO;
        const int k = r % KT, ci = r / KT;
        wl[i] = (co < C_out) ? w[((size_t)co * C_in + ci) * KT + k] : 0.f;
    }
    const int tile = blockIdx.x % ntiles;
    const int b = blockIdx.x / ntiles;
    const int t0 = tile * NOUT;
    const int q = threadIdx.x & 3;
    const int nv = C_in >> 4;          // float4 per thread and row (the quad covers C_in channels: 4 threads x nv x 4)
    // folded GroupNorm coefficients of this thread's channels: float4 number q + 4 j
    float4 ga[MAXV], gs[MAXV];
#pragma unroll
    for (int jv = 0; jv < MAXV; ++jv) {
        ga[jv] = make_float4(1.f, 1.f, 1.f, 1.f); gs[jv] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gscale && jv < nv) {
            ga[jv] = *reinterpret_cast<const float4*>(gscale + (size_t)b * C_in + 4 * (q + 4 * jv));
            gs[jv] = *reinterpret_cast<const float4*>(gshift + (size_t)b * C_in + 4 * (q + 4 * jv));
        }
    }
    __syncthreads();
    for (int r0 = 0; r0 < TW; r0 += 64) {
        const int r = r0 + (threadIdx.x >> 2);
        if (r < TW) {  // (a quad is active or idle as a whole: the cross-lane sums below stay inside it)
            const int t = t0 - PAD + r;
            float acc[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) acc[i] = 0.f;
            if (t >= 0 && t < T) {
                const float* xr = x + ((size_t)b * T + t) * C_in;
#pragma unroll
                for (int jv = 0; jv < MAXV; ++jv) {
                    if (jv < nv) {
                        const int c4 = q + 4 * jv;
                        const float4 v4 = *reinterpret_cast<const float4*>(xr + 4 * c4);
                        float v[4] = {v4.x, v4.y, v4.z, v4.w};
                        if (gscale) {
                            const float a[4] = {ga[jv].x, ga[jv].y, ga[jv].z, ga[jv].w};
                            const float sh[4] = {gs[jv].x, gs[jv].y, gs[jv].z, gs[jv].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = silu_f(fmaf(a[e], v[e], sh[e]));
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float4* wr = reinterpret_cast<const float4*>(wl + (size_t)(4 * c4 + e) * NP);
#pragma unroll
                            for (int i4 = 0; i4 < NP / 4; ++i4) {
                                const float4 ww = wr[i4];
                                acc[4 * i4 + 0] = fmaf(ww.x, v[e], acc[4 * i4 + 0]);
                                acc[4 * i4 + 1] = fmaf(ww.y, v[e], acc[4 * i4 + 1]);
                                acc[4 * i4 + 2] = fmaf(ww.z, v[e], acc[4 * i4 + 2]);
                                acc[4 * i4 + 3] = fmaf(ww.w, v[e], acc[4 * i4 + 3]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                acc[i] += __shfl_xor(acc[i], 1);
                acc[i] += __shfl_xor(acc[i], 2);
            }
            if (q == 0) {
                float4* pr = reinterpret_cast<float4*>(part + (size_t)r * NP);
#pragma unroll
                for (int i4 = 0; i4 < NP / 4; ++i4) pr[i4] = make_float4(acc[4 * i4], acc[4 * i4 + 1], acc[4 * i4 + 2], acc[4 * i4 + 3]);
            }
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 128 * C_out; o += 256) {
        const int co = o >> 7, tl = o & 127;
        const int t = t0 + tl;
        if (tl < NOUT && t < T) {
            float v = bias ? bias[co] : 0.f;
#pragma unroll
            for (int k = 0; k < KT; ++k) v += part[(size_t)(tl + k) * NP + k * MAXCO + co];
            const size_t oo = ((size_t)b * C_out + co) * T + t;
            if (c_out) v = v * c_out[b] + c_skip[b] * skip_src[oo];
            y[oo] = v;
        }
    }
}
}  // namespace

// Round 4, second form (C_out <= 4, C_in in {16, 32, 64, 128}: the EDM / autoencoder heads).  The kernel above reads every weight
// from LDS once per input element and thread -- 5 ds_read_b128 per element for 20 FMAs, 22 us of LDS pipeline per B = 64 launch
// (41-46 us measured for 67 MB).  Here a workgroup owns 64 consecutive input rows of one sample:
//   1. all 256 threads stage the rows with fully coalesced 16-byte loads (a thread owns one 4-channel column: its folded GroupNorm
//      pair stays in registers), activate them and park them in LDS ([row][C_in + 4]: conflict-free for step 2's reads);
//   2. wave w takes channels [w C_in / 4, (w + 1) C_in / 4), lane = row: everything per channel -- the KT x C_out weights -- is
//      wave-uniform, i.e. scalar loads straight from the (C_out, C_in, KT) weight tensor feeding the FMAs' scalar operand (one
//      ds_read_b128 per 4 KT C_out FMAs instead of one per 4);
//   3. the four waves' KT x C_out partial sums per row meet in LDS, where the tap shift is an address offset:
//      output (j, co) = sum over waves and taps of partial[wave][k][co][row j + k]; a workgroup emits 64 - (KT - 1) outputs.
// (Intermediate forms, B = 64 | 16, us: one wave per row set, rows read straight from global memory with a 256-byte lane stride
// 39.2 | 22.8; the same with one wave per channel quarter 33.1 | 14.3; the LDS kernel 45.6 | 18.9.)
namespace {
template <int KT, int NCO>
__global__ __launch_bounds__(256) void head_conv_row_kernel(const float* __restrict__ x, const float* __restrict__ gscale,
                                                            const float* __restrict__ gshift, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ c_out,
                                                            const float* __restrict__ c_skip, const float* __restrict__ skip_src,
                                                            float* __restrict__ y, int T, int C_in, int ntiles, int co0, int C_tot) {
    // Round 5: NCO up to 8 output channels per launch (wave w emits channels w and w + 4) of a head with C_tot of them, starting at
    // co0 -- the 6-channel head of the reference's real data shape (MovingAverageEnvelope: 3 -> 6 channels) in one launch, the
    // 16-channel latent head in two; they used to fall through to the LDS kernel above (214 / 232 us against 30 for 3 channels, B = 64).
    constexpr int PAD = KT / 2, NOUT = 64 - (KT - 1), NP = KT * NCO;
    static_assert(NCO >= 1 && NCO <= 8, "a wave emits at most two output channels");
    extern __shared__ __attribute__((aligned(16))) float shm[];
    const int RS = C_in + 4;                    // floats per staged row
    float* tile = shm;                          // [64][RS]
    float* part = shm;                          // [4][NP][68]: takes the tile's place once every wave is done reading it
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x / ntiles, t0 = (blockIdx.x % ntiles) * NOUT;
    // the epilogue's skip-connection operand: requested now, used after the last barrier
    const int to = t0 + lane;
    const bool emit_t = lane < NOUT && to < T;
    size_t oo[2];
    float skipv[2] = {0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int co = wave + 4 * h;
        oo[h] = ((size_t)b * C_tot + co0 + (co < NCO ? co : 0)) * T + (to < T ? to : 0);
        if (c_out && emit_t && co < NCO) skipv[h] = skip_src[oo[h]];
    }
    // ---- 1. stage
    {
        const int ncol = C_in >> 2, rpp = 256 / ncol;   // 4-channel columns, rows per pass
        const int c4 = threadIdx.x % ncol, r0 = threadIdx.x / ncol;
        float4 a4 = make_float4(1.f, 1.f, 1.f, 1.f), s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gscale) {
            a4 = *reinterpret_cast<const float4*>(gscale + (size_t)b * C_in + 4 * c4);
            s4 = *reinterpret_cast<const float4*>(gshift + (size_t)b * C_in + 4 * c4);
        }
        const float* xb = x + (size_t)b * T * C_in + 4 * c4;
        for (int r = r0; r < 64; r += rpp) {
            const int t = t0 - PAD + r;
            float4 u = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding of the ACTIVATED input
            if (t >= 0 && t < T) {
                u = *reinterpret_cast<const float4*>(xb + (size_t)t * C_in);
                if (gscale) {
                    u.x = silu_f(fmaf(a4.x, u.x, s4.x)); u.y = silu_f(fmaf(a4.y, u.y, s4.y));
                    u.z = silu_f(fmaf(a4.z, u.z, s4.z)); u.w = silu_f(fmaf(a4.w, u.w, s4.w));
                }
            }
            *reinterpret_cast<float4*>(tile + r * RS + 4 * c4) = u;
        }
    }
    __syncthreads();
    // ---- 2. lane = row, wave = channel quarter
    const int cq = C_in >> 2;
    const float* ur = tile + lane * RS + wave * cq;
    float acc[KT][NCO];
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int co = 0; co < NCO; ++co) acc[k][co] = 0.f;
    const float* wc[NCO];   // per output channel: this wave's (C_in / 4, KT) weight block -- the 4 KT weights of a step are consecutive
#pragma unroll
    for (int co = 0; co < NCO; ++co) wc[co] = w + ((size_t)(co0 + co) * C_in + wave * cq) * KT;
    // ONE 4-channel step per loop trip, not unrolled: with several steps in the body hipcc hoists all their scalar loads to the top
    // of the trip and parks 240 scalars in VGPR lanes (210 v_readlane + 146 v_writelane per 120 packed FMAs)
#pragma unroll 1
    for (int c4 = 0; c4 < (cq >> 2); ++c4) {
        const float4 v4 = *reinterpret_cast<const float4*>(ur + 4 * c4);
        const float u[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int co = 0; co < NCO; ++co) {
            const float* wr = wc[co] + 4 * c4 * KT;   // wave-uniform: scalar loads
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k = 0; k < KT; ++k) acc[k][co] = fmaf(wr[e * KT + k], u[e], acc[k][co]);
        }
    }
    // ---- 3. partial sums -> LDS, tap shift = address offset
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int co = 0; co < NCO; ++co) {
            float* pr = part + ((wave * NP) + k * NCO + co) * 68;
            pr[lane] = acc[k][co];
            if (lane < 4) pr[64 + lane] = 0.f;   // (read by the outputs the workgroup does not emit)
        }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < (NCO > 4 ? 2 : 1); ++h) {
        const int co = wave + 4 * h;
        if (emit_t && co < NCO) {
            float v = bias ? bias[co0 + co] : 0.f;
#pragma unroll
            for (int k = 0; k < KT; ++k)
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) v += part[((wv * NP) + k * NCO + co) * 68 + lane + k];
            if (c_out) v = v * c_out[b] + c_skip[b] * skipv[h];
            y[oo[h]] = v;
        }
    }
}
}  // namespace

// dynamic LDS of head_conv_kernel for a shape, or 0 when the kernel is not built for it (the one place that knows the limits: the
// launcher below and the plan builder, through tq_head_conv_lds_bytes, both ask here)
static size_t head_conv_lds(int C_in, int C_out, int ktaps) {
    if (C_in < 16 || C_in % 16 || C_in > 128 || C_out < 1 || C_out > 16 || (ktaps != 1 && ktaps != 3 && ktaps != 5)) return 0;
    const int maxco = C_out <= 4 ? 4 : 16;
    const size_t sh = ((size_t)C_in * ktaps * maxco + (size_t)128 * ktaps * maxco) * sizeof(float);
    return sh > 64 * 1024 ? 0 : sh;
}

extern "C" size_t tq_head_conv_lds_bytes(int C_in, int C_out, int ktaps) { return head_conv_lds(C_in, C_out, ktaps); }

extern "C" int tq_head_conv_fwd(const float* x, const float* gscale, const float* gshift, const float* w, const float* bias,
                                const float* c_out, const float* c_skip, const float* skip_src, float* y, int B, int T,
                                int C_in, int C_out, int ktaps, hipStream_t stream) {
    if (!x || !w || !y) return TQ_ERR_ARG;
    if ((gscale == nullptr) != (gshift == nullptr)) return TQ_ERR_ARG;
    if (c_out && (!c_skip || !skip_src)) return TQ_ERR_ARG;
    const size_t sh = head_conv_lds(C_in, C_out, ktaps);
    if (B <= 0 || T <= 0 || sh == 0) return TQ_ERR_SHAPE;
    // TQDNE_HEAD_FWD=lds: the round-3 kernel for every shape (A/B switch)
    static const int row_form = [] { const char* e = getenv("TQDNE_HEAD_FWD"); return (e && e[0] == 'l') ? 0 : 1; }();
    if (row_form && (C_out <= 4 || C_out == 6 || C_out == 8 || C_out == 16) && (C_in == 16 || C_in == 32 || C_in == 64 || C_in == 128)) {
        const int nt = (T + (64 - (ktaps - 1)) - 1) / (64 - (ktaps - 1));
        const int per = C_out <= 8 ? C_out : 8;   // output channels per launch (16 = two launches of 8: the input tile is staged twice)
        const size_t st_ = (size_t)64 * (C_in + 4), sp_ = (size_t)4 * ktaps * per * 68;
        const size_t shr = (st_ > sp_ ? st_ : sp_) * sizeof(float);   // <= 54 KB (the partial sums reuse the staged tile's space)
#define TQ_HEADR(K, N, CO0) hipLaunchKernelGGL((head_conv_row_kernel<K, N>), dim3(B * nt), dim3(256), shr, stream, x, gscale, gshift, w, bias, \
                                               c_out, c_skip, skip_src, y, T, C_in, nt, CO0, C_out)
#define TQ_HEADRK(K) { if (per == 1) TQ_HEADR(K, 1, 0); else if (per == 2) TQ_HEADR(K, 2, 0); else if (per == 3) TQ_HEADR(K, 3, 0); \
                       else if (per == 4) TQ_HEADR(K, 4, 0); else if (per == 6) TQ_HEADR(K, 6, 0); \
                       else { for (int c0 = 0; c0 < C_out; c0 += 8) TQ_HEADR(K, 8, c0); } }
        if (ktaps == 5) TQ_HEADRK(5)
        else if (ktaps == 3) TQ_HEADRK(3)
        else TQ_HEADRK(1)
#undef TQ_HEADRK
#undef TQ_HEADR
        TQ_CHECK_LAUNCH();
        return 0;
    }
    const int nout = 128 - (ktaps - 1);   // output positions per workgroup (head_conv_kernel's NOUT)
    const int ntiles = (T + nout - 1) / nout;
    const int maxco = C_out <= 4 ? 4 : 16;
#define TQ_HEAD(K)                                                                                          \
    {                                                                                                       \
        auto kern = (maxco == 4) ? head_conv_kernel<K, 4> : head_conv_kernel<K, 16>;                        \
        hipLaunchKernelGGL(kern, dim3(B * ntiles), dim3(256), sh, stream, x, gscale, gshift, w, bias, c_out, c_skip, skip_src, y, T, C_in, C_out, ntiles); \
    }
    if (ktaps == 5) TQ_HEAD(5)
    else if (ktaps == 3) TQ_HEAD(3)
    else if (ktaps == 1) TQ_HEAD(1)
    else return TQ_ERR_SHAPE;
#undef TQ_HEAD
    TQ_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================
// Embedding: Fourier features -> time MLP (+ cond MLP).  One workgroup per sample; each wave owns
// outputs o = wave, wave+4, ... and its 64 lanes split the input dimension (coalesced weight rows).
// hidden[b][0][:] = pre-activation of time_mlp.0, hidden[b][1][:] = pre-activation of cond_mlp.0.
// =================================================================================================
namespace {
__device__ __forceinline__ void gemv_rows(const float* __restrict__ w, const float* __restrict__ bias,
                                          const float* in_sh, int n_in, int n_out, float* out_sh, bool accumulate) {
    // out[o] = bias[o] + sum_i w[o][i] * in[i] for a 256-thread workgroup.  LPO lanes share an output (1 for short rows, 4 for long
    // ones) and walk its row in interleaved 16-byte pieces, so every output's dependent-load chain is n_in / (4 LPO) deep and all
    // 256 / LPO outputs of a pass are in flight together (a wave-per-output reduction kept 3/4 of the lanes idle on 64-long rows
    // and paid 6 shuffles per output: 78 us for the embedding MLPs).
    const int tid = threadIdx.x;
    const int LPO = (n_in > 64 && (n_in & 15) == 0) ? 4 : 1;
    const int r = tid % LPO, o_local = tid / LPO;
    const int per_pass = 256 / LPO;
    const bool vec = (n_in & 3) == 0;
    for (int o0 = 0; o0 < n_out; o0 += per_pass) {
        const int o = o0 + o_local;
        float a = 0.f;
        if (o < n_out) {
            const float* wr = w + (size_t)o * n_in;
            if (vec) {
                const int n4 = n_in >> 2;
#pragma unroll 4
                for (int q = r; q < n4; q += LPO) {
                    const float4 wv = *reinterpret_cast<const float4*>(wr + 4 * q);
                    const float4 xv = *reinterpret_cast<const float4*>(in_sh + 4 * q);
                    a += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
                }
            } else {
                for (int i = r; i < n_in; i += LPO) a = fmaf(wr[i], in_sh[i], a);
            }
        }
        for (int sft = 1; sft < LPO; sft <<= 1) a += __shfl_xor(a, sft);
        if (o < n_out && r == 0) {
            const float v = a + bias[o];
            out_sh[o] = accumulate ? out_sh[o] + v : v;
        }
    }
}

__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ t, const float* __restrict__ cond,
                                                    const float* __restrict__ fw, const float* __restrict__ w0,
                                                    const float* __restrict__ b0, const float* __restrict__ w2,
                                                    const float* __restrict__ b2, const float* __restrict__ cw0,
                                                    const float* __restrict__ cb0, const float* __restrict__ cw2,
                                                    const float* __restrict__ cb2, float* __restrict__ emb,
                                                    float* __restrict__ silu_emb, float* __restrict__ hidden, int mc,
                                                    int ncond) {
    // grid (B, 4): every workgroup recomputes the cheap first layers (mc -> E and ncond -> E) and owns one quarter of the
    // outputs of the two E x E layers, so the 2 * E * E weight reads of a sample are spread over four CUs
    extern __shared__ __attribute__((aligned(16))) float shm[];
    const int E = 4 * mc;
    float* four = shm;        // [mc]
    float* h = four + mc;     // [E]  silu(time hidden)
    float* c = h + E;         // [E]  silu(cond hidden)
    float* e = c + E;         // [E/4] this workgroup's outputs
    float* cs = e + E / 4;    // [ncond]
    const int b = blockIdx.x, q = blockIdx.y;
    const float tv = t[b];
    const int half = mc >> 1;
    for (int i = threadIdx.x; i < half; i += 256) {
        // blocks.py:23: ((x * W) * 2) * pi in fp32, then sin | cos
        const float arg = ((tv * fw[i]) * 2.0f) * 3.14159265358979323846f;
        four[i] = sinf(arg);
        four[half + i] = cosf(arg);
    }
    for (int i = threadIdx.x; i < ncond; i += 256) cs[i] = cond[(size_t)b * ncond + i];
    __syncthreads();
    gemv_rows(w0, b0, four, mc, E, h, false);
    if (ncond > 0) gemv_rows(cw0, cb0, cs, ncond, E, c, false);
    __syncthreads();
    for (int i = threadIdx.x; i < E; i += 256) {
        if (hidden && q == 0) {
            hidden[((size_t)b * 2 + 0) * E + i] = h[i];
            if (ncond > 0) hidden[((size_t)b * 2 + 1) * E + i] = c[i];
        }
        h[i] = silu_f(h[i]);
        if (ncond > 0) c[i] = silu_f(c[i]);
    }
    __syncthreads();
    const int o0 = q * (E / 4);
    gemv_rows(w2 + (size_t)o0 * E, b2 + o0, h, E, E / 4, e, false);
    if (ncond > 0) {
        __syncthreads();
        gemv_rows(cw2 + (size_t)o0 * E, cb2 + o0, c, E, E / 4, e, true);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < E / 4; i += 256) {
        emb[(size_t)b * E + o0 + i] = e[i];
        silu_emb[(size_t)b * E + o0 + i] = silu_f(e[i]);
    }
}

// out (B, N) = x (B, E) W^T (N, E) + bias.  Workgroup = 64 outputs x 16 samples; W tile in LDS with an odd
// leading dimension (lanes = consecutive outputs -> conflict-free), x rows broadcast from LDS.
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out, int B, int E,
                                                     int N) {
    extern __shared__ float shm[];
    const int LDW = E + 1;
    float* ws = shm;            // [64][LDW]
    float* xs = ws + 64 * LDW;  // [16][E]
    const int n0 = blockIdx.x * 64;
    const int b0 = blockIdx.y * 16;
    for (int i = threadIdx.x; i < 64 * E; i += 256) {
        const int r = i / E, c = i % E;
        ws[r * LDW + c] = (n0 + r < N) ? w[(size_t)(n0 + r) * E + c] : 0.f;
    }
    for (int i = threadIdx.x; i < 16 * E; i += 256) {
        const int r = i / E, c = i % E;
        xs[i] = (b0 + r < B) ? x[(size_t)(b0 + r) * E + c] : 0.f;
    }
    __syncthreads();
    const int j = threadIdx.x & 63;
    const int bq = threadIdx.x >> 6;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < E; ++i) {
        const float wv = ws[j * LDW + i];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = fmaf(wv, xs[(bq + 4 * q) * E + i], a[q]);
    }
    if (n0 + j < N) {
        const float bv = bias ? bias[n0 + j] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b = b0 + bq + 4 * q;
            if (b < B) out[(size_t)b * N + n0 + j] = a[q] + bv;
        }
    }
}
}  // namespace

extern "C" int tq_embed_fwd(const float* t, const float* cond, const float* fourier_w, const float* w0, const float* b0,
                            const float* w2, const float* b2, const float* cw0, const float* cb0, const float* cw2,
                            const float* cb2, float* emb, float* silu_emb, float* hidden, int B, int mc, int ncond,
                            hipStream_t stream) {
    if (!t || !fourier_w || !w0 || !b0 || !w2 || !b2 || !emb || !silu_emb) return TQ_ERR_ARG;
    if (ncond > 0 && (!cond || !cw0 || !cb0 || !cw2 || !cb2)) return TQ_ERR_ARG;
    if (B <= 0 || mc <= 0 || mc % 2 || ncond < 0) return TQ_ERR_SHAPE;
    const size_t sh = (size_t)(mc + 9 * mc + ncond + 4) * sizeof(float);
    hipLaunchKernelGGL(embed_kernel, dim3(B, 4), dim3(256), sh, stream, t, cond, fourier_w, w0, b0, w2, b2, cw0, cb0, cw2, cb2,
                       emb, silu_emb, hidden, mc, ncond);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_linear_fwd(const float* x, const float* w, const float* bias, float* out, int B, int E, int N,
                             hipStream_t stream) {
    if (!x || !w || !out) return TQ_ERR_ARG;
    if (B <= 0 || E <= 0 || N <= 0) return TQ_ERR_SHAPE;
    const size_t sh = ((size_t)64 * (E + 1) + (size_t)16 * E) * sizeof(float);
    if (sh > 160 * 1024) return TQ_ERR_SHAPE;
    auto kern = linear_kernel;
    if (sh > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(kern, dim3((N + 63) / 64, (B + 15) / 16), dim3(256), sh, stream, x, w, bias, out, B, E, N);
    TQ_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================
// EDM scalar maps, noise injection, loss, Heun updates
// =================================================================================================
namespace {
__global__ void edm_scalars_kernel(const float* __restrict__ sigma, int sstride, float sd, float* c_in, float* c_out,
                                   float* c_skip, float* c_noise, float* lw, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float s = sigma[(size_t)b * sstride];
    const float sd2 = sd * sd;
    const float q = s * s + sd2;
    if (c_in) c_in[b] = 1.0f / sqrtf(q);              // edm.py:33-34
    if (c_out) c_out[b] = (s * sd) / sqrtf(q);        // edm.py:30-31
    if (c_skip) c_skip[b] = sd2 / q;                  // edm.py:27-28
    if (c_noise) c_noise[b] = 0.25f * logf(s);        // edm.py:36-37
    if (lw) { const float ssd = s * sd; lw[b] = q / (ssd * ssd); }  // edm.py:24-25
}

__global__ void cm_scalars_kernel(const float* __restrict__ sigma, int sstride, float sd, float smin, float* c_out,
                                  float* c_skip, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float s = sigma[(size_t)b * sstride];
    const float d = s - smin;
    c_skip[b] = (sd * sd) / (d * d + sd * sd);               // consistency_model.py:69
    c_out[b] = (sd * d) / sqrtf(sd * sd + s * s);            // consistency_model.py:71-73
}

__global__ void noise_inject_kernel(const float* __restrict__ y, const float* __restrict__ n, const float* __restrict__ eps,
                                    float pmean, float pstd, float* __restrict__ sigma, float* __restrict__ x, int per) {
    const int b = blockIdx.y;
    const float s = expf(eps[b] * pstd + pmean);  // edm.py:21-22
    if (blockIdx.x == 0 && threadIdx.x == 0) sigma[b] = s;
    const size_t base = (size_t)b * per;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x)
        x[base + i] = y[base + i] + n[base + i] * s;  // edm.py:128-129
}

__global__ __launch_bounds__(256) void edm_loss_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                       const float* __restrict__ lw, float* __restrict__ loss,
                                                       float* __restrict__ dpred, int per, float inv_n) {
    const int b = blockIdx.y;
    const float wgt = lw[b];
    const size_t base = (size_t)b * per;
    float a = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
        const float d = pred[base + i] - y[base + i];
        a += wgt * d * d;
        if (dpred) dpred[base + i] = 2.0f * wgt * d * inv_n;
    }
    a = wave_sum(a);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (ws[0] + ws[1] + ws[2] + ws[3]) * inv_n);
}

__global__ void heun_euler_kernel(const double* __restrict__ x, const float* __restrict__ den, const float* __restrict__ sig,
                                  const float* __restrict__ sign, double* __restrict__ dcur, double* __restrict__ xn,
                                  float* __restrict__ x32, size_t n) {
    const float s = *sig, s2 = *sign;
    const double sd = (double)s;
    const double dt = (double)(s2 - s);  // fp32 subtraction of two 0-dim fp32 tensors, then promoted (edm.py:183)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double xv = x[i];
        const double d = (xv - (double)den[i]) / sd;
        const double r = xv + d * dt;
        dcur[i] = d;
        xn[i] = r;
        x32[i] = (float)r;
    }
}

__global__ void heun_correct_kernel(const double* __restrict__ x, const double* __restrict__ xn,
                                    const float* __restrict__ den2, const double* __restrict__ dcur,
                                    const float* __restrict__ sig, const float* __restrict__ sign, double* __restrict__ xo,
                                    float* __restrict__ x32, size_t n) {
    const float s = *sig, s2 = *sign;
    const double sn = (double)s2;
    const double dt = (double)(s2 - s);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double dp = (xn[i] - (double)den2[i]) / sn;          // edm.py:193
        const double r = x[i] + dt * (0.5 * dcur[i] + 0.5 * dp);   // edm.py:194
        xo[i] = r;
        x32[i] = (float)r;
    }
}

__global__ void sampler_init_kernel(const double* __restrict__ z, const float* __restrict__ s0, double* __restrict__ x,
                                    float* __restrict__ x32, size_t n) {
    const double s = (double)*s0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double r = z[i] * s;
        x[i] = r;
        x32[i] = (float)r;
    }
}

// Stochastic ("churned") sampler, the temporary noise increase of edm.py:205-208:  x_hat = x + (n * S_noise) * c, with the fp64
// state x, the fp64 unit draw n and c = sqrt(sigma_hat^2 - sigma^2) formed by the caller in fp32 exactly as the reference does
// with its 0-dim fp32 tensors (a device scalar, promoted to fp64 here like torch's type promotion does).
__global__ void heun_churn_kernel(const double* __restrict__ x, const double* __restrict__ unit, const float* __restrict__ coef,
                                  double s_noise, double* __restrict__ xh, float* __restrict__ x32, size_t n) {
    const double c = (double)*coef;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double r = x[i] + (unit[i] * s_noise) * c;
        xh[i] = r;
        x32[i] = (float)r;
    }
}

// out[b, :] = x[b, :] + noise[b, :] * sigma[b]   (consistency_model.py:150-160: the two noised copies of the iCT step; also
// edm.py:128-129 with a given sigma)
__global__ void axpy_sigma_kernel(const float* __restrict__ x, const float* __restrict__ nz, const float* __restrict__ sigma,
                                  float* __restrict__ out, int per) {
    const int b = blockIdx.y;
    const float s = sigma[b];
    const size_t base = (size_t)b * per;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x)
        out[base + i] = x[base + i] + nz[base + i] * s;
}

// Weighted pseudo-Huber distance of the improved consistency training step (consistency_model.py:163-173):
//   loss = mean_{b,i} w_b * (sqrt((pred - target)^2 + c^2) - c);   dpred = w_b * (pred - target) / sqrt(.) / n
__global__ __launch_bounds__(256) void pseudo_huber_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                           const float* __restrict__ wgt, float c, float* __restrict__ loss,
                                                           float* __restrict__ dpred, int per, float inv_n) {
    const int b = blockIdx.y;
    const float w = wgt[b];
    const size_t base = (size_t)b * per;
    const float c2 = c * c;
    float a = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
        const float d = pred[base + i] - target[base + i];
        const float root = sqrtf(d * d + c2);
        a += w * (root - c);
        if (dpred) dpred[base + i] = d / root * w * inv_n;
    }
    a = wave_sum(a);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (ws[0] + ws[1] + ws[2] + ws[3]) * inv_n);
}

// Mean squared error and its gradient (autoencoder.py:61-63): loss += sum (a - b)^2 / n, d = 2 (a - b) / n
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ loss,
                                                  float* __restrict__ d, size_t n, float inv_n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float df = a[i] - b[i];
        acc += df * df;
        if (d) d[i] = 2.0f * df * inv_n;
    }
    acc = wave_sum(acc);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (ws[0] + ws[1] + ws[2] + ws[3]) * inv_n);
}

// VAE bottleneck (autoencoder.py:37-43, 64-66; blocks.py:233-260): enc (B, 2L, T) = [mean | log_std] on the channel axis.
//   forward : z = mean + eps * exp(log_std);  kl += mean_{b,t} 0.5 * sum_c (mean^2 + exp(2 log_std) - 2 log_std - 1)
//   backward: d mean = dz + kw * mean,  d log_std = dz * eps * std + kw * (std^2 - 1),  kw = kl_weight / (B * T)
__global__ __launch_bounds__(256) void vae_reparam_fwd_kernel(const float* __restrict__ enc, const float* __restrict__ eps,
                                                              float* __restrict__ z, float* __restrict__ kl, int L, int T,
                                                              size_t n, float inv_bt) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t lt = (size_t)L * T;
        const size_t b = i / lt, r = i - b * lt;
        const float m = enc[b * 2 * lt + r], ls = enc[b * 2 * lt + lt + r];
        const float sd = expf(ls);
        z[i] = m + eps[i] * sd;
        acc += 0.5f * (m * m + sd * sd - 2.0f * ls - 1.0f);
    }
    if (kl) {
        acc = wave_sum(acc);
        __shared__ float ws[4];
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(kl, (ws[0] + ws[1] + ws[2] + ws[3]) * inv_bt);
    }
}

__global__ void vae_reparam_bwd_kernel(const float* __restrict__ enc, const float* __restrict__ eps, const float* __restrict__ dz,
                                       float* __restrict__ denc, int L, int T, size_t n, float kw) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t lt = (size_t)L * T;
        const size_t b = i / lt, r = i - b * lt;
        const float m = enc[b * 2 * lt + r], ls = enc[b * 2 * lt + lt + r];
        const float sd = expf(ls);
        const float g = dz[i];
        denc[b * 2 * lt + r] = g + kw * m;
        denc[b * 2 * lt + lt + r] = g * eps[i] * sd + kw * (sd * sd - 1.0f);
    }
}

// The stem input of a model conditioned on a signal (edm.py:108-109): out (B, C0 + C1, T) = [x * scale_b | cond_signal]
__global__ void concat_scale_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ cs,
                                    float* __restrict__ out, int C0, int C1, int T, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t ct = (size_t)(C0 + C1) * T;
        const size_t b = i / ct, r = i - b * ct;
        const size_t c = r / T, t = r - c * T;
        out[i] = (c < (size_t)C0) ? x[(b * C0 + c) * T + t] * (scale ? scale[b] : 1.0f) : cs[(b * C1 + (c - C0)) * T + t];
    }
}

inline unsigned ew_grid(size_t n) {
    size_t g = (n + 255) / 256;
    return (unsigned)(g > 2048 ? 2048 : (g ? g : 1));
}
}  // namespace

extern "C" int tq_edm_scalars(const float* sigma, int sigma_stride, float sigma_data, float* c_in, float* c_out,
                              float* c_skip, float* c_noise, float* lweight, int B, hipStream_t stream) {
    if (!sigma || B <= 0) return TQ_ERR_ARG;
    hipLaunchKernelGGL(edm_scalars_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, sigma, sigma_stride, sigma_data, c_in,
                       c_out, c_skip, c_noise, lweight, B);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_cm_scalars(const float* sigma, int sigma_stride, float sigma_data, float sigma_min, float* c_out,
                             float* c_skip, int B, hipStream_t stream) {
    if (!sigma || !c_out || !c_skip || B <= 0) return TQ_ERR_ARG;
    hipLaunchKernelGGL(cm_scalars_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, sigma, sigma_stride, sigma_data, sigma_min,
                       c_out, c_skip, B);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_edm_noise_inject(const float* y, const float* unit_noise, const float* eps, float P_mean, float P_std,
                                   float* sigma, float* x_noisy, int B, int n_per_sample, hipStream_t stream) {
    if (!y || !unit_noise || !eps || !sigma || !x_noisy || B <= 0 || n_per_sample <= 0) return TQ_ERR_ARG;
    const unsigned gx = (unsigned)((n_per_sample + 255) / 256 > 64 ? 64 : (n_per_sample + 255) / 256);
    hipLaunchKernelGGL(noise_inject_kernel, dim3(gx, B), dim3(256), 0, stream, y, unit_noise, eps, P_mean, P_std, sigma,
                       x_noisy, n_per_sample);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_edm_loss(const float* pred, const float* y, const float* lweight, float* loss_out, float* dpred, int B,
                           int n_per_sample, hipStream_t stream) {
    if (!pred || !y || !lweight || !loss_out || B <= 0 || n_per_sample <= 0) return TQ_ERR_ARG;
    hipError_t e = hipMemsetAsync(loss_out, 0, sizeof(float), stream);
    if (e != hipSuccess) return (int)e;
    const unsigned gx = (unsigned)((n_per_sample + 255) / 256 > 16 ? 16 : (n_per_sample + 255) / 256);
    const float inv_n = (float)(1.0 / ((double)B * (double)n_per_sample));
    hipLaunchKernelGGL(edm_loss_kernel, dim3(gx, B), dim3(256), 0, stream, pred, y, lweight, loss_out, dpred, n_per_sample,
                       inv_n);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_heun_euler(const double* x, const float* denoised, const float* sigma, const float* sigma_next,
                             double* d_cur, double* x_next, float* x32, size_t n, hipStream_t stream) {
    if (!x || !denoised || !sigma || !sigma_next || !d_cur || !x_next || !x32 || n == 0) return TQ_ERR_ARG;
    hipLaunchKernelGGL(heun_euler_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, x, denoised, sigma, sigma_next, d_cur,
                       x_next, x32, n);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_heun_correct(const double* x, const double* x_next, const float* denoised_next, const double* d_cur,
                               const float* sigma, const float* sigma_next, double* x_out, float* x32, size_t n,
                               hipStream_t stream) {
    if (!x || !x_next || !denoised_next || !d_cur || !sigma || !sigma_next || !x_out || !x32 || n == 0) return TQ_ERR_ARG;
    hipLaunchKernelGGL(heun_correct_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, x, x_next, denoised_next, d_cur, sigma,
                       sigma_next, x_out, x32, n);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_sampler_init(const double* unit_noise, const float* sigma0, double* x, float* x32, size_t n,
                               hipStream_t stream) {
    if (!unit_noise || !sigma0 || !x || !x32 || n == 0) return TQ_ERR_ARG;
    hipLaunchKernelGGL(sampler_init_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, unit_noise, sigma0, x, x32, n);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_heun_churn(const double* x, const double* unit_noise, const float* coef, double s_noise, double* x_hat,
                             float* x32, size_t n, hipStream_t stream) {
    if (!x || !unit_noise || !coef || !x_hat || !x32 || n == 0) return TQ_ERR_ARG;
    hipLaunchKernelGGL(heun_churn_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, x, unit_noise, coef, s_noise, x_hat, x32, n);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_axpy_sigma(const float* x, const float* noise, const float* sigma, float* out, int B, int n_per_sample,
                             hipStream_t stream) {
    if (!x || !noise || !sigma || !out || B <= 0 || n_per_sample <= 0) return TQ_ERR_ARG;
    const int gx = (n_per_sample + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(axpy_sigma_kernel, dim3(gx, B), dim3(256), 0, stream, x, noise, sigma, out, n_per_sample);
    TQ_CHECK_LAUNCH();
    return 0;
}

// ---- DDPM (tqdne/diffusion.py; the scheduler arithmetic is the published algorithm of Ho et al. 2020, section 3) -----------------
namespace {
// out[b, :] = a[b] x[b, :] + c[b] y[b, :]   (forward process q(x_t | x_0): a = sqrt(abar_t), c = sqrt(1 - abar_t), diffusion.py:98)
__global__ void scale_add2_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ a,
                                  const float* __restrict__ c, float* __restrict__ out, int per) {
    const int b = blockIdx.y;
    const float fa = a[b], fc = c[b];
    const size_t base = (size_t)b * per;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x)
        out[base + i] = fa * x[base + i] + fc * y[base + i];
}
// one ancestral step x_t -> x_{t-1} (diffusion.py:77): x0 = epsilon-prediction ? (x - sqrt(1 - abar_t) eps) / sqrt(abar_t) : model output,
// clipped to +-clip (clip <= 0: off); out = c0 x0 + ct x + sigma z
__global__ void ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ mo, const float* __restrict__ z,
                                 float* __restrict__ out, size_t n, int eps_pred, float sqrt_1m_abar, float inv_sqrt_abar,
                                 float clip, float c0, float ct, float sigma) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float xv = x[i];
        float x0 = eps_pred ? (xv - sqrt_1m_abar * mo[i]) * inv_sqrt_abar : mo[i];
        if (clip > 0.f) x0 = fminf(fmaxf(x0, -clip), clip);
        float r = c0 * x0 + ct * xv;
        if (z) r += sigma * z[i];
        out[i] = r;
    }
}
}  // namespace

extern "C" int tq_scale_add2(const float* x, const float* y, const float* a, const float* c, float* out, int B, int n_per_sample,
                             hipStream_t stream) {
    if (!x || !y || !a || !c || !out || B <= 0 || n_per_sample <= 0) return TQ_ERR_ARG;
    const int gx = (n_per_sample + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(scale_add2_kernel, dim3(gx, B), dim3(256), 0, stream, x, y, a, c, out, n_per_sample);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_ddpm_step(const float* x, const float* model_out, const float* noise, float* out, size_t n, int epsilon_prediction,
                            double sqrt_one_minus_abar, double inv_sqrt_abar, double clip, double coef_x0, double coef_xt,
                            double sigma, hipStream_t stream) {
    if (!x || !model_out || !out || n == 0) return TQ_ERR_ARG;
    size_t g = (n + 256 * 4 - 1) / (256 * 4);
    if (g > 65535) g = 65535;
    hipLaunchKernelGGL(ddpm_step_kernel, dim3((unsigned)g), dim3(256), 0, stream, x, model_out, noise, out, n, epsilon_prediction,
                       (float)sqrt_one_minus_abar, (float)inv_sqrt_abar, (float)clip, (float)coef_x0, (float)coef_xt, (float)sigma);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_pseudo_huber_loss(const float* pred, const float* target, const float* weight, float c, float* loss_out,
                                    float* dpred, int B, int n_per_sample, hipStream_t stream) {
    if (!pred || !target || !weight || !loss_out || B <= 0 || n_per_sample <= 0) return TQ_ERR_ARG;
    hipError_t e = hipMemsetAsync(loss_out, 0, sizeof(float), stream);
    if (e != hipSuccess) return (int)e;
    const int gx = (n_per_sample + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(pseudo_huber_kernel, dim3(gx, B), dim3(256), 0, stream, pred, target, weight, c, loss_out, dpred,
                       n_per_sample, 1.0f / ((float)B * (float)n_per_sample));
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_mse_loss(const float* a, const float* b, float* loss_out, float* d, size_t n, hipStream_t stream) {
    if (!a || !b || !loss_out || n == 0) return TQ_ERR_ARG;
    hipError_t e = hipMemsetAsync(loss_out, 0, sizeof(float), stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mse_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, a, b, loss_out, d, n, 1.0f / (float)n);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_vae_reparam_fwd(const float* enc, const float* eps, float* z, float* kl_out, int B, int L, int T,
                                  hipStream_t stream) {
    if (!enc || !eps || !z || B <= 0 || L <= 0 || T <= 0) return TQ_ERR_ARG;
    if (kl_out) {
        hipError_t e = hipMemsetAsync(kl_out, 0, sizeof(float), stream);
        if (e != hipSuccess) return (int)e;
    }
    const size_t n = (size_t)B * L * T;
    hipLaunchKernelGGL(vae_reparam_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, enc, eps, z, kl_out, L, T, n,
                       1.0f / ((float)B * (float)T));
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_vae_reparam_bwd(const float* enc, const float* eps, const float* dz, float* denc, float kl_weight, int B, int L,
                                  int T, hipStream_t stream) {
    if (!enc || !eps || !dz || !denc || B <= 0 || L <= 0 || T <= 0) return TQ_ERR_ARG;
    const size_t n = (size_t)B * L * T;
    hipLaunchKernelGGL(vae_reparam_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, enc, eps, dz, denc, L, T, n,
                       kl_weight / ((float)B * (float)T));
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_concat_scale(const float* x, const float* scale, const float* cond_signal, float* out, int B, int C0, int C1,
                               int T, hipStream_t stream) {
    if (!x || !cond_signal || !out || B <= 0 || C0 <= 0 || C1 <= 0 || T <= 0) return TQ_ERR_ARG;
    const size_t n = (size_t)B * (C0 + C1) * T;
    hipLaunchKernelGGL(concat_scale_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, x, scale, cond_signal, out, C0, C1, T, n);
    TQ_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Fused Adam + EMA over the whole model (HBM-bound: reads p, g, m, v [, ema], writes p, m, v [, ema] once; 10 x 62 MB for the
// paper net).  torch.optim.Adam spends ~8 multi-tensor passes on the same update.
// ------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float step_size, float omb1, float b2, float omb2,
                                      float eps, float ibc2, float gscale, float decay) {
    g *= gscale;
    p *= decay;  // AdamW's decoupled weight decay (1 - lr * wd); 1 for plain Adam
    m = fmaf(omb1, g - m, m);
    v = fmaf(omb2, g * g, b2 * v);
    p -= step_size * (m / (sqrtf(v) * ibc2 + eps));
}

__global__ __launch_bounds__(256) void adam_ema_kernel(const TqAdamChunk* __restrict__ chunks, float step_size, float omb1, float b2,
                                                       float omb2, float eps, float ibc2, float ema_w, float gscale,
                                                       float decay, const int* __restrict__ skip_flag) {
    if (skip_flag && *skip_flag != 0) return;   // (uniform scalar load: the whole launch drops out together)
    const TqAdamChunk c = chunks[blockIdx.x];
    const int n4 = c.n & ~3;
    for (int i = threadIdx.x * 4; i < n4; i += 256 * 4) {
        float4 p = *reinterpret_cast<const float4*>(c.p + i);
        const float4 g = *reinterpret_cast<const float4*>(c.g + i);
        float4 m = *reinterpret_cast<const float4*>(c.m + i);
        float4 v = *reinterpret_cast<const float4*>(c.v + i);
        adam1(p.x, g.x, m.x, v.x, step_size, omb1, b2, omb2, eps, ibc2, gscale, decay);
        adam1(p.y, g.y, m.y, v.y, step_size, omb1, b2, omb2, eps, ibc2, gscale, decay);
        adam1(p.z, g.z, m.z, v.z, step_size, omb1, b2, omb2, eps, ibc2, gscale, decay);
        adam1(p.w, g.w, m.w, v.w, step_size, omb1, b2, omb2, eps, ibc2, gscale, decay);
        *reinterpret_cast<float4*>(c.p + i) = p;
        *reinterpret_cast<float4*>(c.m + i) = m;
        *reinterpret_cast<float4*>(c.v + i) = v;
        if (c.ema) {
            float4 e = *reinterpret_cast<const float4*>(c.ema + i);
            e.x = fmaf(ema_w, p.x - e.x, e.x); e.y = fmaf(ema_w, p.y - e.y, e.y);
            e.z = fmaf(ema_w, p.z - e.z, e.z); e.w = fmaf(ema_w, p.w - e.w, e.w);
            *reinterpret_cast<float4*>(c.ema + i) = e;
        }
    }
    const int i = n4 + threadIdx.x;
    if (i < c.n) {
        float p = c.p[i], m = c.m[i], v = c.v[i];
        adam1(p, c.g[i], m, v, step_size, omb1, b2, omb2, eps, ibc2, gscale, decay);
        c.p[i] = p; c.m[i] = m; c.v[i] = v;
        if (c.ema) c.ema[i] = fmaf(ema_w, p - c.ema[i], c.ema[i]);
    }
}
}  // namespace

extern "C" int tq_adam_ema_step_guarded(const TqAdamChunk* chunks, int n_chunks, double step_size, double beta1, double beta2,
                                        double eps, double inv_bias2_sqrt, double ema_weight, double grad_scale,
                                        double decay_factor, const int32_t* skip_flag, hipStream_t stream) {
    if (!chunks || n_chunks < 0) return TQ_ERR_ARG;
    if (n_chunks == 0) return 0;
    hipLaunchKernelGGL(adam_ema_kernel, dim3(n_chunks), dim3(256), 0, stream, chunks, (float)step_size, (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)eps, (float)inv_bias2_sqrt, (float)ema_weight, (float)grad_scale,
                       (float)decay_factor, reinterpret_cast<const int*>(skip_flag));
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_adam_ema_step(const TqAdamChunk* chunks, int n_chunks, double step_size, double beta1, double beta2, double eps,
                                double inv_bias2_sqrt, double ema_weight, double grad_scale, double decay_factor,
                                hipStream_t stream) {
    return tq_adam_ema_step_guarded(chunks, n_chunks, step_size, beta1, beta2, eps, inv_bias2_sqrt, ema_weight, grad_scale,
                                    decay_factor, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------
// MovingAverageEnvelope representation (representation.py:41-60).  HBM-bound (4 B in, 8 B out per sample): one workgroup per
// 1024 positions of one (sample, channel) row; |x| of the tile + window halo is staged in LDS, each thread forms the first
// window sum of its 4 outputs in float64 and slides it.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int ENV_TILE = 1024;

__global__ __launch_bounds__(256) void envelope_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int T,
                                                           int W, double log_eps, double eps, int n_tiles) {
    extern __shared__ float env_lds[];
    // (row, tile) folded into grid.x: grid.y would cap N * C at 65535 rows
    const int row = blockIdx.x / n_tiles, n = row / C, c = row % C;
    const int t0 = (blockIdx.x % n_tiles) * ENV_TILE;
    const int lo = W / 2, hi = (W - 1) / 2;
    const float* xr = x + (size_t)row * T;
    for (int i = threadIdx.x; i < ENV_TILE + W; i += 256) {
        const int t = t0 - lo + i;
        env_lds[i] = (t >= 0 && t < T) ? fabsf(xr[t]) : 0.f;
    }
    __syncthreads();
    float* o_scaled = out + ((size_t)n * 2 * C + c) * T;
    float* o_log = out + ((size_t)n * 2 * C + C + c) * T;
    const int l0 = threadIdx.x * 4;  // local index of the first output; its window is env_lds[l0 .. l0 + lo + hi]
    if (t0 + l0 >= T) return;
    double s = 0.0;
    for (int j = 0; j <= lo + hi; ++j) s += (double)env_lds[l0 + j];
    const double half_log = log(log_eps) * 0.5, inv_w = 1.0 / (double)W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = t0 + l0 + k;
        if (t >= T) break;
        const double env = s * inv_w;
        o_scaled[t] = (float)((double)xr[t] / (env + eps));
        o_log[t] = (float)(log(env + log_eps) - half_log);
        s += (double)env_lds[l0 + k + lo + hi + 1] - (double)env_lds[l0 + k];
    }
}

__global__ __launch_bounds__(256) void envelope_inv_kernel(const float* __restrict__ r, float* __restrict__ out, int C, int T,
                                                           double log_eps, double eps, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int t = (int)(i % T);
    const size_t row = i / T;
    const int c = (int)(row % C);
    const size_t n = row / C;
    const double scaled = r[(n * 2 * C + c) * T + t];
    const double log_env = r[(n * 2 * C + C + c) * T + t];
    out[i] = (float)(scaled * (exp(log_env + log(log_eps) * 0.5) + eps));
}
}  // namespace

extern "C" int tq_envelope_fwd(const float* x, float* out, int N, int C, int T, int window, double log_eps, double eps,
                               hipStream_t stream) {
    if (!x || !out) return TQ_ERR_ARG;
    if (N <= 0 || C <= 0 || T <= 0 || window <= 0 || window > T || window > 4096 || !(log_eps > 0.0)) return TQ_ERR_SHAPE;
    const int n_tiles = (T + ENV_TILE - 1) / ENV_TILE;
    if ((size_t)n_tiles * N * C > 0x7fffffffull) return TQ_ERR_SHAPE;
    const dim3 grid((unsigned)((size_t)n_tiles * N * C));
    hipLaunchKernelGGL(envelope_fwd_kernel, grid, dim3(256), (ENV_TILE + window + 1) * sizeof(float), stream, x, out, C, T,
                       window, log_eps, eps, n_tiles);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_envelope_inv(const float* repr, float* out, int N, int C, int T, double log_eps, double eps,
                               hipStream_t stream) {
    if (!repr || !out) return TQ_ERR_ARG;
    if (N <= 0 || C <= 0 || T <= 0 || !(log_eps > 0.0)) return TQ_ERR_SHAPE;
    const size_t total = (size_t)N * C * T;
    hipLaunchKernelGGL(envelope_inv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, repr, out, C, T, log_eps,
                       eps, total);
    TQ_CHECK_LAUNCH();
    return 0;
}
