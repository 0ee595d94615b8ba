// GroupNorm32 statistics -> folded per-(b, c) scale / shift of ONE sample, by the threads of one workgroup.
// Shared by gn_finalize_kernel (small_ops.hip: one workgroup per sample, its own launch) and the fused form in the conv epilogue
// (conv1d_mfma.hip: the workgroup that completes a sample's statistics folds them, "last arriver") so that both produce
// bit-identical coefficients: same per-thread summation order, same fp64 combine.
#pragma once
#include "common.hpp"

namespace tq {

// One {sum, sum of squares} pair of the partial statistics.  COHERENT = true: the pair may have been written by another workgroup
// of THIS launch (as one 8-byte agent-scope atomic store, behind that workgroup's drain + barrier + arrival ticket): read it with
// an 8-byte agent-scope atomic load (global_load_dwordx2 sc1: served by L2 / memory, never by this CU's L1) -- the "8-byte agent
// atomics on both sides" form of cdna_hip_programming.md, Guideline 16.
template <bool COHERENT>
__device__ __forceinline__ float2 gn_load_pair(const float* p) {
    if constexpr (COHERENT) {
        const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
    } else {
        return *reinterpret_cast<const float2*>(p);
    }
}

// sh: LDS, (2 C + 64) doubles.  st0 / st1: statistics (B, nslots0 | nslots1, C0 | C1, 2) of the (up to two, concatenated) source tensors;
// COH0 / COH1: see gn_load_pair.  Every thread of the workgroup calls this (it contains barriers); nthreads = blockDim.x.
template <bool COH0, bool COH1>
__device__ __forceinline__ void gn_fold_sample(double* sh, int b, const float* __restrict__ st0, int C0, const float* __restrict__ st1,
                                               int C1, int T, int nslots0, int nslots1, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, float* __restrict__ gscale,
                                               float* __restrict__ gshift, float* __restrict__ mean_rstd,
                                               float* lds_scale = nullptr, float* lds_shift = nullptr) {
    // lds_scale / lds_shift (optional, C floats each, outside ``sh``): the coefficients of this sample ALSO go to these LDS tables (the
    // consuming conv's prologue reads them from there instead of a global round trip behind the stores below)
    // Latency-bound (a few KB per sample): three dependent steps, so every global load is issued as early as possible -- a thread's
    // slot loads all together before the first add.
    const int C = C0 + C1;
    const int tid = threadIdx.x, nthr = blockDim.x;
    double* csum = sh;            // [C][2] channel sums
    double* gstat = sh + 2 * C;   // [32][2] group mean / rstd
    // (round 5) the affine parameters of this thread's channels are requested FIRST: they used to be loaded in the last step, a third
    // dependent global round trip behind the statistics and the group reduction
    constexpr int NPRE = 4;
    const bool pre = C <= NPRE * nthr;
    float gam[NPRE], bet[NPRE];
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
        const int c = tid + j * nthr;
        const int cc = (pre && c < C) ? c : 0;
        gam[j] = gamma[cc];
        bet[j] = beta[cc];
    }
    // thread = (channel, slot part): PARTS threads share one channel's slots so the dependent-load chain is short
    const int PARTS = (C <= 64) ? 4 : ((C <= 128) ? 2 : 1);
    for (int idx = tid; idx < C * PARTS; idx += nthr) {
        const int c = idx / PARTS, part = idx % PARTS;
        const float* st;
        int cs, cc;
        const bool first = c < C0;
        if (first) { st = st0; cs = C0; cc = c; } else { st = st1; cs = C1; cc = c - C0; }
        const int nslots = first ? nslots0 : nslots1;   // (the two sources may have been written with different position tiles)
        const float* pp = st + ((size_t)b * nslots * cs + cc) * 2;
        auto ld = [&](int s) -> float2 __attribute__((always_inline)) {
            const float* q = pp + (size_t)s * cs * 2;
            return first ? gn_load_pair<COH0>(q) : gn_load_pair<COH1>(q);
        };
        double s1 = 0.0, s2 = 0.0;
        int s = part;
        for (; s + 7 * PARTS < nslots; s += 8 * PARTS) {   // 8 loads in flight
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld(s + u * PARTS);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s1 += (double)v[u].x; s2 += (double)v[u].y; }
        }
#ifdef TQ_ABL_FOLD_REMAINDER   // (A/B build: the remainder batch always issued, as in rounds 3-5)
        if (true) {
#else
        if (s < nslots) {
#endif
            // up to 7 left: again all loads first (clamped index, masked add).  (Round 6: skipped when nothing is left -- the
            // clamped loads of an empty remainder were one more dependent global round trip in every fold.)
            float2 v[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int su = s + u * PARTS;
                v[u] = ld(su < nslots ? su : (nslots - 1));
            }
#pragma unroll
            for (int u = 0; u < 7; ++u)
                if (s + u * PARTS < nslots) { s1 += (double)v[u].x; s2 += (double)v[u].y; }
        }
        if (PARTS == 1) { csum[2 * c] = s1; csum[2 * c + 1] = s2; }
        else {
            // combine the parts of one channel (adjacent lanes)
            for (int o = 1; o < PARTS; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            if (part == 0) { csum[2 * c] = s1; csum[2 * c + 1] = s2; }
        }
    }
    __syncthreads();
    const int G = C / GN_GROUPS;
    if (tid < GN_GROUPS) {
        const int g = tid;
        double s1 = 0.0, s2 = 0.0;
        for (int j = 0; j < G; ++j) { s1 += csum[2 * (g * G + j)]; s2 += csum[2 * (g * G + j) + 1]; }
        const double n = (double)G * (double)T;
        const double mean = s1 / n;
        double var = fma(-mean, mean, s2 / n);   // (explicit contraction: the same bits in every kernel this is inlined into)
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)GN_EPS);
        gstat[2 * g] = mean;
        gstat[2 * g + 1] = rstd;
        if (mean_rstd) {
            mean_rstd[((size_t)b * GN_GROUPS + g) * 2] = (float)mean;
            mean_rstd[((size_t)b * GN_GROUPS + g) * 2 + 1] = (float)rstd;
        }
    }
    __syncthreads();
    if (pre) {
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int c = tid + j * nthr;
            if (c < C) {
                const int g = c / G;
                const float mean = (float)gstat[2 * g], rstd = (float)gstat[2 * g + 1];
                const float a = gam[j] * rstd;
                const float sft = fmaf(-mean, a, bet[j]);   // (explicit: the same rounding in every kernel this is inlined into)
                gscale[(size_t)b * C + c] = a;
                gshift[(size_t)b * C + c] = sft;
                if (lds_scale) { lds_scale[c] = a; lds_shift[c] = sft; }
            }
        }
        return;
    }
    for (int c = tid; c < C; c += nthr) {
        const int g = c / G;
        const float mean = (float)gstat[2 * g], rstd = (float)gstat[2 * g + 1];
        const float a = gamma[c] * rstd;
        const float sft = fmaf(-mean, a, beta[c]);
        gscale[(size_t)b * C + c] = a;
        gshift[(size_t)b * C + c] = sft;
        if (lds_scale) { lds_scale[c] = a; lds_shift[c] = sft; }
    }
}

}  // namespace tq
