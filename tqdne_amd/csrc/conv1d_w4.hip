// Fused 1-D convolution, ONE WAVE PER SIMD (4 waves x 512 registers) -- the stride-1 forward launches of the fp16 + MX-fp6 scheme.
//
// Same arithmetic, operand formats, LDS image and accumulation order as conv1d_mfma.hip's scheme-2 kernel (results are
// bit-identical to it); what changes is the execution structure.  There (two waves per SIMD, 255 registers each) a chunk of 64
// input channels is {MFMA stream} -> {GroupNorm / SiLU / fp16 + fp6 conversion of the next chunk} -> barrier, with both waves of a
// SIMD in lock-step, so the matrix pipe idles during the conversion and the halo pass (25-30 % of the chunk loop, phase stamps in
// DESIGN.md section 5).  Here a wave owns its SIMD and the whole register file:
//   * wave tile 64 output channels x 128 positions = 128 accumulator registers (AccVGPRs); workgroup = 4 waves as 4 x 1
//     (256 channels x 128 positions) or 2 x 2 (128 x 256);
//   * the conversion of chunk c + 1 is cut into per-element micro-operations that are spread over the (tap, t-block) steps of
//     chunk c's MFMA stream: ~16 VALU per step of 12 MFMAs, i.e. inside the 8 of 16 cycles per MFMA in which the SIMD can issue
//     vector instructions (MI355X_MICROARCH.md, "vector-instruction ISSUE cost"); scalar fp32 ops only (packed fp32 VALU is an
//     anti-lever beside MFMAs, same table);
//   * the fp32 rows of chunk c + 2 are requested as soon as a task of chunk c + 1 has been converted (its registers are free):
//     a load has about one chunk phase (4 us) to arrive;
//   * weight fragments of a tap: 4 channel blocks x 4 x 16 bytes = 64 registers, two buffers, refilled two taps ahead from L2;
//   * the halo rows (KT - 1 of them) are one more task that every thread runs on rows NT + (its row & (KT - 2)): lanes that do
//     not own a halo row recompute one (same loads: L1 hits; same LDS words rewritten with the same values) -- no predicate,
//     no branch in the stream.
// Replaces GroupNorm32 -> SiLU -> Conv1d -> (+emb) -> (+skip) of the reference's ResBlock at inference (tqdne/unet.py:86-102,131-143).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "conv_args.hpp"
#include "../../include/tqdne_hip.h"

using namespace tq;

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int KT, int WM, int WN>
struct W4 {
    static_assert(WM * WN == 4, "four waves");
    static constexpr int NT = 128 * WN;          // positions per workgroup
    static constexpr int MT = 64 * WM;           // output channels per workgroup
    static constexpr int HALO = KT - 1;
    static constexpr int ROWS = NT + HALO;
    static constexpr int NFULL = NT * 4 / 256;   // full staging tasks per thread and chunk (a task = 16 channels of one row)
    static constexpr int NTASK = NFULL + (HALO > 0 ? 1 : 0);
    static constexpr int PLANE = ROWS * 128;     // bytes: rows of 64 fp16 (main plane) / 4 fp6 block fragments (correction plane)
    static constexpr int BUF = 2 * PLANE;
    static constexpr int LDS_BYTES = 2 * BUF;    // double buffered
    static constexpr int PAD = KT / 2;
    static constexpr int NS = KT * 8;            // (tap, t-block) steps per chunk
    static constexpr int NOPS = NTASK * 17;      // micro-operations per chunk: 16 elements + 1 finish per task
    // a task's fp32 row (16 registers) is requested AHEAD micro-operations before its first element is converted -- in the previous
    // chunk phase where that reaches back beyond this one's start: ~2 tasks in flight + the one being converted, whatever NTASK is
#ifndef TQ_W4_AHEAD
#define TQ_W4_AHEAD 34
#endif
    static constexpr int AHEAD = TQ_W4_AHEAD;
    static constexpr int load_pos(int k) { return ((17 * k - AHEAD) % NOPS + NOPS) % NOPS; }   // micro-op slot of task k's load
    static constexpr bool load_prev(int k) { return 17 * k - AHEAD < 0; }                     // issued in the previous phase
};

__device__ __forceinline__ float f4c(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

#ifdef TQ_W4_STAMP
// diagnostic build only: per-workgroup clock stamps around the whole kernel body -- [0] / [1] s_memtime (shader
// clock) at entry / exit, [2] / [3] s_memrealtime (100 MHz) at entry / exit; in-kernel clock = d[0,1] / d[2,3] * 100 MHz
static __device__ unsigned long long tq_w4_stamps[4096 * 4];
#define TQ_W4_T(i, j) if (threadIdx.x == 0 && blockIdx.x < 4096) { tq_w4_stamps[blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memtime(); \
                                                                    tq_w4_stamps[blockIdx.x * 4 + (j)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define TQ_W4_T(i, j)
#endif

// ACT: 0 none, 1 folded GroupNorm, 2 GroupNorm + SiLU.   FUSE: the ResBlock's 1x1 skip conv rides as extra single-tap chunks.
template <int KT, int WM, int WN, int ACT, bool FUSE>
__global__ __launch_bounds__(256, 1) void conv1d_w4_kernel(const ConvArgs p) {
    using C = W4<KT, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    TQ_W4_T(0, 2)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;

    const int n_ttiles = (p.T_out + C::NT - 1) / C::NT;
    const int n_ctiles = p.C_out / C::MT;
    // channel tiles of one (b, t-tile) re-read the same rows: ids 8 apart = one XCD (workgroups are dealt round-robin over the XCDs)
    const int bid = blockIdx.x;
    int ct, tile;
    const int ntile = p.B * n_ttiles;
    if (n_ctiles > 1 && (ntile & 7) == 0) {
        const int grp = bid / (8 * n_ctiles), within = bid % (8 * n_ctiles);
        ct = within >> 3;
        tile = grp * 8 + (within & 7);
    } else {
        ct = bid % n_ctiles;
        tile = bid / n_ctiles;
    }
    const int tt = tile % n_ttiles, b = tile / n_ttiles;
    const int t0 = tt * C::NT;
    const int co_wave = ct * C::MT + wm * 64;

    const int Cin = p.C0 + p.C1;
    const int nchunks = Cin >> 6;
    const int nskip = FUSE ? ((p.sC0 + p.sC1) >> 6) : 0;
    const int nstages = nchunks + nskip;
    const int T = p.T_in;

    // ---- staging bookkeeping: task it of this thread = 16 channels (m2) of row (tid >> 2) + 64 it; the halo task: see the header
    const int m2 = tid & 3;
    int rowv[C::NTASK];      // LDS row of the task
    int srow[C::NTASK];      // source row (clamped into the signal)
    float pad1[C::NTASK];    // 1 inside the signal; +inf (ACT 2: zero through the sigmoid) or 0 (mask multiply) outside
#pragma unroll
    for (int it = 0; it < C::NTASK; ++it) {
        const int row = it < C::NFULL ? (tid >> 2) + 64 * it : C::NT + ((tid >> 2) & (C::HALO - 1));
        const int pos = t0 - C::PAD + row;
        rowv[it] = row;
        srow[it] = pos < 0 ? 0 : (pos >= T ? T - 1 : pos);
        pad1[it] = (pos >= 0 && pos < T) ? 1.0f : (ACT >= 2 ? __builtin_inff() : 0.0f);
    }
    float* gtab = reinterpret_cast<float*>(lds + C::LDS_BYTES);   // [Cin] scale, [Cin] shift of sample b

    auto chunk_src = [&](int stage, int& cs) -> const float* __attribute__((always_inline)) {
        const bool sk = FUSE && stage >= nchunks;
        const int cb = (sk ? stage - nchunks : stage) << 6;
        const float* a0 = sk ? p.sx0 : p.x0;
        const float* a1 = sk ? p.sx1 : p.x1;
        const int c0 = sk ? p.sC0 : p.C0, c1 = sk ? p.sC1 : p.C1;
        const bool first = cb < c0;
        cs = first ? c0 : c1;
        return (first ? a0 : a1) + (size_t)b * T * cs + (first ? cb : cb - c0) + 16 * m2;
    };

    float4 raw[C::NTASK][4];
    auto load_task = [&](int stage, auto it_c) __attribute__((always_inline)) {
        constexpr int it = decltype(it_c)::value;
        const int st = stage < nstages ? stage : nstages - 1;   // (past the end: a harmless re-read)
        int cs;
        const float* base = chunk_src(st, cs);
        const float4* q = reinterpret_cast<const float4*>(base + (size_t)srow[it] * cs);
        raw[it][0] = q[0]; raw[it][1] = q[1]; raw[it][2] = q[2]; raw[it][3] = q[3];
    };

    // ---- conversion state of the task in progress
    f32x16 xl, xf;
    f16x8 h0, h1;
    float mx = 0.f;
    float4 ga[2], gs[2];   // folded GroupNorm coefficients of 4 channels, double buffered by group parity
    const float nlog2e = -1.4426950408889634f;

    auto coef_load = [&](int stage, int g, int par) __attribute__((always_inline)) {
        if constexpr (ACT >= 1) {
            const int st = stage < nchunks ? stage : nchunks - 1;
            const float* ptr = gtab + (st << 6) + 16 * m2 + 4 * g;
            ga[par] = *reinterpret_cast<const float4*>(ptr);
            gs[par] = *reinterpret_cast<const float4*>(ptr + Cin);
        }
    };

    // element j of task k of the chunk `stage` (being staged into LDS buffer `buf`)
    auto conv_elem = [&](int stage, auto k_c, auto j_c) __attribute__((always_inline)) {
        constexpr int k = decltype(k_c)::value, j = decltype(j_c)::value;
        const bool act = !(FUSE && (stage < nstages ? stage : nstages - 1) >= nchunks);
        if constexpr (ACT >= 1 && (j & 3) == 0) {
            // request the next group's coefficients (this task's, or group 0 of the next task: the same 16 channels; after the
            // last task of the chunk: the next chunk's)
            constexpr int gn = ((j >> 2) + 1) & 3;
            coef_load((k == C::NTASK - 1 && j == 12) ? stage + 1 : stage, gn, gn & 1);
        }
        float u = f4c(raw[k][j >> 2], j & 3);
        if (ACT >= 1 && act) u = fmaf(f4c(ga[(j >> 2) & 1], j & 3), u, f4c(gs[(j >> 2) & 1], j & 3));
        if (ACT >= 2 && act) {
            // u * sigmoid(u) = u / (1 + 2^(-u log2 e)); rows outside the signal: 1 / (inf + e) = 0
            const float e = __builtin_amdgcn_exp2f(u * nlog2e) + pad1[k];
            u = u * __builtin_amdgcn_rcpf(e);
        } else {
            u = u * (ACT >= 2 ? (pad1[k] == 1.0f ? 1.0f : 0.0f) : pad1[k]);
        }
        // ONE fp16 rounding of the fp32 value u, used by the main plane AND by the remainder below.  Without the barrier hipcc
        // contracts "u = a * b; (half)u" into v_fma_mixlo_f16(a, b, 0) -- the fp16 rounding of the EXACT product -- for one of the two
        // uses only; when fl32(a * b) is an exact tie between two fp16 numbers the two roundings differ, the remainder is then
        // taken against the wrong neighbour and that element is off by a whole fp16 ulp (found as 3e-5 output errors on ~1 input
        // row in 200, tools/w4_diag.py).
        asm volatile("" : "+v"(u));
        const _Float16 hh = (_Float16)u;   // |x| > 65504 -> inf, NaN stays NaN: out-of-range inputs surface in the output
        if constexpr (j < 8) h0[j] = hh; else h1[j - 8] = hh;
        const float r = fmaf((float)hh, -4096.f, u * 4096.f);   // (x - fp16(x)) * 2^12, exact
        xl[j] = r; xf[j] = u;
        mx = fmaxf(fmaxf(mx, fabsf(r)), fabsf(u));
    };
    // finish task k: block scale, fp6 pack, LDS stores; then its registers take the same task of chunk stage + 1
    auto conv_fin = [&](int stage, int buf, auto k_c) __attribute__((always_inline)) {
        constexpr int k = decltype(k_c)::value;
        unsigned char* hi_plane = lds + buf * C::BUF;
        unsigned char* lo_plane = hi_plane + C::PLANE;
        const unsigned bb = e8m0_block_scale(mx);
        const u32x6 pk = cvt_2xpk16_fp6(xl, xf, __uint_as_float(bb << 23));
        const int row = rowv[k];
        const int sw = row & 7, ro = row * 128;
        *reinterpret_cast<uint4*>(hi_plane + ro + (((2 * m2) ^ sw) << 4)) = __builtin_bit_cast(uint4, h0);
        *reinterpret_cast<uint4*>(hi_plane + ro + (((2 * m2 + 1) ^ sw) << 4)) = __builtin_bit_cast(uint4, h1);
        *reinterpret_cast<uint4*>(lo_plane + ro + ((m2 ^ sw) << 4)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        // the lane's E8M0 byte for the MFMA, with the 2^-12 of both correction products folded in
        *reinterpret_cast<uint4*>(lo_plane + ro + (((4 + m2) ^ sw) << 4)) = make_uint4(pk[4], pk[5], bb - 12u, 0u);
        mx = 0.f;
    };
    // micro-operation m (0 .. NOPS - 1) of the conversion of chunk `stage`; the row loads ride in the same slots
    auto conv_op = [&](int stage, int buf, auto m_c) __attribute__((always_inline)) {
        constexpr int m = decltype(m_c)::value;
        constexpr int k = m / 17, j = m % 17;
        static_for<0, C::NTASK>([&](auto kl_c) __attribute__((always_inline)) {
            constexpr int kl = decltype(kl_c)::value;
            if constexpr (C::load_pos(kl) == m) load_task(C::load_prev(kl) ? stage + 1 : stage, kl_c);
        });
        if constexpr (j < 16) conv_elem(stage, std::integral_constant<int, k>{}, std::integral_constant<int, j>{});
        else conv_fin(stage, buf, std::integral_constant<int, k>{});
    };

    // ---- accumulators and the MFMA stream
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int kq = lane >> 4;
    const int tl_lane = wn * 128 + (lane & 15);
    const uint4* wbase = p.wpk + ((size_t)(co_wave >> 4) * 4) * 64 + lane;
    const size_t wstep = (size_t)p.ncob_pad * 4 * 64;   // uint4 per (chunk, tap)
    const int last_step = nchunks * KT + nskip - 1;
    // Weight fragments of one tap, per 16-channel block: fp16 (channels 0..31), fp16 (32..63), the 6-dword e2m3 block and its E8M0
    // byte.  The MFMAs are issued from inline asm with register-class constraints (below): accumulators and weight fragments live
    // in AccVGPRs ("a"), activation fragments and the two scale words in VGPRs ("v").  Hence the load shapes: the fp6 block as a
    // 16-byte + an 8-byte load that together ARE its 6-register tuple (no spare component the allocator could treat as dead and
    // reuse, no component that has to live in the other register file), the scale word as a load of its own.
    struct WBuf { f16x8 h0[4], h1[4]; u32x6 c6[4]; int sc[4]; };
    WBuf wa, wb;
    auto load_w = [&](int step, WBuf& w) __attribute__((always_inline)) {
#ifdef TQ_W4_ABL_NOW   // (diagnostic build: weights stay L1-resident; wrong results)
        const int st = step & 1;
#else
        const int st = step < last_step ? step : last_step;   // (past the end: a harmless re-read)
#endif
        const uint4* wp = wbase + (size_t)st * wstep;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            w.h0[cb] = __builtin_bit_cast(f16x8, wp[(cb * 4 + 0) * 64]);
            w.h1[cb] = __builtin_bit_cast(f16x8, wp[(cb * 4 + 1) * 64]);
            const uint4 lo = wp[(cb * 4 + 2) * 64];
            const uint2 hi = *reinterpret_cast<const uint2*>(wp + (cb * 4 + 3) * 64);
            w.c6[cb] = u32x6{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y};
            w.sc[cb] = reinterpret_cast<const int*>(wp + (cb * 4 + 3) * 64)[2];
        }
    };
    auto tap_base = [&](int k) -> int __attribute__((always_inline)) {
        const int rowk = tl_lane + k;
        return rowk * 128 + ((kq ^ (rowk & 7)) << 4);
    };
    struct BFrag { f16x8 h0, h1; uint4 lo, hi; };
    auto read_b = [&](const unsigned char* hi_plane, const unsigned char* lo_plane, int b0, int tb, BFrag& f) __attribute__((always_inline)) {
#ifdef TQ_W4_ABL_NOLDSMOVE   // (diagnostic build: every t-block reads the same LDS rows; wrong results)
        const int toff = 0;
        (void)tb;
#else
        const int toff = tb * 16 * 128;
#endif
        f.h0 = *reinterpret_cast<const f16x8*>(hi_plane + b0 + toff);          // fp16, channels 8 kq ...
        f.h1 = *reinterpret_cast<const f16x8*>(hi_plane + (b0 ^ 64) + toff);   // fp16, channels 32 + 8 kq ...
        f.lo = *reinterpret_cast<const uint4*>(lo_plane + b0 + toff);          // fp6 block, dwords 0..3
        f.hi = *reinterpret_cast<const uint4*>(lo_plane + (b0 ^ 64) + toff);   // dwords 4, 5, E8M0 byte
    };
    // The twelve MFMAs of a (tap, t-block) step as three asm statements of four (one per channel block).  Why asm: with the
    // builtins hipcc renamed accumulators between MFMAs (destination != addend register), moved them through VGPRs
    // (v_accvgpr_read behind an s_nop 7) and kept weight fragments in VGPRs while VALU operands commuted through AccVGPRs -- the
    // bare MFMA / weight / LDS stream ran at 2.4x its matrix-pipe time (tools/w4_time.py, -DTQ_W4_ABL_NOCONV).  An in/out "+a"
    // operand cannot be renamed, and "a" puts the weights where MFMA reads them for free.  Per accumulator the order is that of
    // conv1d_mfma.hip: fp16 (channels 0..31), fp16 (32..63), block-scaled corrections.  hipcc inserts the s_waitcnt for operands
    // that are pending loads; wait states it does not insert for asm: VALU-written operand -> MFMA (the two v_mov that complete
    // the activation side's 6-register tuple): s_nop 1 in front of the third statement.
    auto mma_f16 = [&](const f16x8 (&w)[4], const f16x8& bv, auto tb_c) __attribute__((always_inline)) {
        constexpr int tb = decltype(tb_c)::value;
        f32x4 &a0 = acc[0][tb], &a1 = acc[1][tb], &a2 = acc[2][tb], &a3 = acc[3][tb];   // (asm operands alone do not capture)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %8, %0\n\t"
                     "v_mfma_f32_16x16x32_f16 %1, %5, %8, %1\n\t"
                     "v_mfma_f32_16x16x32_f16 %2, %6, %8, %2\n\t"
                     "v_mfma_f32_16x16x32_f16 %3, %7, %8, %3"
                     : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3)
                     : "a"(w[0]), "a"(w[1]), "a"(w[2]), "a"(w[3]), "v"(bv));
    };
    auto mma_fp6 = [&](const WBuf& w, const BFrag& f, auto tb_c) __attribute__((always_inline)) {
        constexpr int tb = decltype(tb_c)::value;
        const u32x6 bc = {f.lo.x, f.lo.y, f.lo.z, f.lo.w, f.hi.x, f.hi.y};
        const int sb = (int)f.hi.z;
        f32x4 &a0 = acc[0][tb], &a1 = acc[1][tb], &a2 = acc[2][tb], &a3 = acc[3][tb];
        asm volatile("s_nop 1\n\t"
                     "v_mfma_scale_f32_16x16x128_f8f6f4 %0, %4, %8, %0, %9, %13 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
                     "v_mfma_scale_f32_16x16x128_f8f6f4 %1, %5, %8, %1, %10, %13 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
                     "v_mfma_scale_f32_16x16x128_f8f6f4 %2, %6, %8, %2, %11, %13 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
                     "v_mfma_scale_f32_16x16x128_f8f6f4 %3, %7, %8, %3, %12, %13 op_sel_hi:[0,0,0] cbsz:2 blgp:2"
                     : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3)
                     : "a"(w.c6[0]), "a"(w.c6[1]), "a"(w.c6[2]), "a"(w.c6[3]), "v"(bc), "v"(w.sc[0]), "v"(w.sc[1]), "v"(w.sc[2]),
                       "v"(w.sc[3]), "v"(sb));
    };

    // One chunk phase on LDS buffer `buf`: NTAPS x 8 (tap, t-block) steps; with STAGE the conversion of chunk `stage + 1` into the
    // other buffer and the row loads that follow it are spread over the steps.  Weight buffers: taps alternate a, b, a, ...; after tap
    // g its buffer is refilled with tap g + 2 of the (chunk, tap) sequence.  g0 = weight step index of this chunk's first tap.
    // (KT is odd, so the refill for the next chunk's tap 0 only starts when this chunk's last tap is done: that one load is exposed
    // once per chunk -- see DESIGN.md for the variants that hide it and what they cost in registers.)
    auto phase = [&](int stage, int buf, int g0, auto ntaps_c, auto first_tap_c, auto stage_c) __attribute__((always_inline)) {
        constexpr int NTAPS = decltype(ntaps_c)::value, K0 = decltype(first_tap_c)::value;
        constexpr bool STAGE = decltype(stage_c)::value;
        constexpr int NU = NTAPS * 8;
        constexpr int NG = NU * 3;   // MFMA groups of four: the conversion's micro-operations are dealt out behind them
        const unsigned char* hi_plane = lds + buf * C::BUF;
        const unsigned char* lo_plane = hi_plane + C::PLANE;
        BFrag bf[2];
        int b0 = tap_base(K0), b0n = b0;
        read_b(hi_plane, lo_plane, b0, 0, bf[0]);
        static_for<0, NU>([&](auto u_c) __attribute__((always_inline)) {
            constexpr int u = decltype(u_c)::value;
            constexpr int kk = u >> 3, tb = u & 7;
            const auto TB = std::integral_constant<int, tb>{};
            if constexpr (tb == 0 && kk + 1 < NTAPS) b0n = tap_base(K0 + kk + 1);
            if constexpr (u + 1 < NU) read_b(hi_plane, lo_plane, ((u + 1) >> 3) == kk ? b0 : b0n, (u + 1) & 7, bf[(u + 1) & 1]);
            auto ops = [&](auto g_c) __attribute__((always_inline)) {
#ifndef TQ_W4_ABL_NOCONV   // (diagnostic build: the MFMA / weight / LDS-read stream alone; wrong results)
                if constexpr (STAGE) {
                    constexpr int g = 3 * u + decltype(g_c)::value;
                    constexpr int lo = g * C::NOPS / NG, hi = (g + 1) * C::NOPS / NG;
                    static_for<lo, hi>([&](auto m_c) __attribute__((always_inline)) { conv_op(stage + 1, buf ^ 1, m_c); });
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            };
            if constexpr (kk & 1) mma_f16(wb.h0, bf[u & 1].h0, TB); else mma_f16(wa.h0, bf[u & 1].h0, TB);
            ops(std::integral_constant<int, 0>{});
            if constexpr (kk & 1) mma_f16(wb.h1, bf[u & 1].h1, TB); else mma_f16(wa.h1, bf[u & 1].h1, TB);
            ops(std::integral_constant<int, 1>{});
            if constexpr (kk & 1) mma_fp6(wb, bf[u & 1], TB); else mma_fp6(wa, bf[u & 1], TB);
            ops(std::integral_constant<int, 2>{});
            if constexpr (tb == 7) {   // tap done: its buffer takes the weights of the step two ahead
                if constexpr (NTAPS > 1) {
                    const int nxt = (kk + 2 < NTAPS) ? (g0 + kk + 2) : ((kk & 1) ? (g0 + NTAPS + 1) : (g0 + NTAPS));
                    if constexpr (kk & 1) load_w(nxt, wb); else load_w(nxt, wa);
                }
                b0 = b0n;
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };

    // ---- prologue: coefficient table, first chunk staged with nothing to hide under
    if constexpr (ACT >= 1) {
        const float4* gsrc = reinterpret_cast<const float4*>(p.gscale + (size_t)b * Cin);
        const float4* hsrc = reinterpret_cast<const float4*>(p.gshift + (size_t)b * Cin);
        float4* g4 = reinterpret_cast<float4*>(gtab);
        for (int i = tid; i < (Cin >> 2); i += 256) { g4[i] = gsrc[i]; g4[(Cin >> 2) + i] = hsrc[i]; }
    }
    static_for<0, C::NTASK>([&](auto it_c) __attribute__((always_inline)) { load_task(0, it_c); });
    if constexpr (KT > 1) {
        load_w(0, wa);
        load_w(1, wb);
    }
    __syncthreads();   // (the coefficient table)
    coef_load(0, 0, 0);
    static_for<0, C::NOPS>([&](auto m_c) __attribute__((always_inline)) {
        constexpr int m = decltype(m_c)::value;
        constexpr int k = m / 17, j = m % 17;
        if constexpr (j < 16) conv_elem(0, std::integral_constant<int, k>{}, std::integral_constant<int, j>{});
        else {
            conv_fin(0, 0, std::integral_constant<int, k>{});
            // the state the first phase expects: the rows of chunk 1 for the tasks whose load belongs to the phase before
            if constexpr (C::load_prev(k)) load_task(1, std::integral_constant<int, k>{});
        }
    });
    __syncthreads();

    // ---- main loop over the stages: main chunks (KT taps), then the fused skip conv's chunks (centre tap)
    {
        const auto KTc = std::integral_constant<int, KT>{};
        const auto Z = std::integral_constant<int, 0>{};
        if constexpr (!FUSE) {
            for (int c = 0; c + 1 < nstages; ++c) {
                phase(c, c & 1, c * KT, KTc, Z, std::true_type{});
                __syncthreads();
            }
            phase(nstages - 1, (nstages - 1) & 1, (nstages - 1) * KT, KTc, Z, std::false_type{});
        } else {
            const auto ONE = std::integral_constant<int, 1>{};
            const auto PADc = std::integral_constant<int, C::PAD>{};
            for (int c = 0; c < nchunks; ++c) {   // (the last main chunk stages the first skip chunk)
                phase(c, c & 1, c * KT, KTc, Z, std::true_type{});
                __syncthreads();
            }
            // skip chunks: buffer a holds this step's weights, b the next one's; b is shifted into a and refilled
            for (int j = 0; j < nskip; ++j) {
                const int st = nchunks + j;
                if (j + 1 < nskip) phase(st, st & 1, 0, ONE, PADc, std::true_type{});
                else phase(st, st & 1, 0, ONE, PADc, std::false_type{});
                wa = wb;
                load_w(nchunks * KT + j + 2, wb);
                if (j + 1 < nskip) __syncthreads();
            }
        }
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (the last MFMAs' results: hipcc pads nothing behind asm)
    // ---- epilogue (as conv1d_mfma.hip, EPI == 0): + bias + emb (+ skip bias), store, GroupNorm partial sums, range guard
    const bool poly = p.flags & TQ_CONV_POLY2;
    const int Cr = poly ? (p.C_out >> 1) : p.C_out;
    const int ph = (poly && co_wave >= Cr) ? 1 : 0;
    const int co_real = co_wave - ph * Cr;
    const int slot = poly ? 2 * ((t0 >> 7) + wn) + ph : (t0 >> 7) + wn;
    const float* emb_b = (p.flags & TQ_CONV_EMB) ? p.emb + (size_t)b * p.emb_stride : nullptr;
    float4 add[4];
    float s1[4][4], s2[4][4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int co = co_real + cb * 16 + 4 * (lane >> 4);
        add[cb] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) add[cb] = *reinterpret_cast<const float4*>(p.bias + co);
        if (emb_b) {
            const float4 e = *reinterpret_cast<const float4*>(emb_b + co);
            add[cb].x += e.x; add[cb].y += e.y; add[cb].z += e.z; add[cb].w += e.w;
        }
        if (FUSE && p.sbias) {
            const float4 e = *reinterpret_cast<const float4*>(p.sbias + co);
            add[cb].x += e.x; add[cb].y += e.y; add[cb].z += e.z; add[cb].w += e.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[cb][j] = 0.f; s2[cb][j] = 0.f; }
    }
#pragma unroll
    for (int tb = 0; tb < 8; ++tb) {
        const int t = t0 + wn * 128 + tb * 16 + (lane & 15);
        if (t < p.T_out) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {   // (a wave's 64 channels = two whole 128-byte lines per row, stored back to back)
                const int co = co_real + cb * 16 + 4 * (lane >> 4);
                const size_t o = poly ? ((size_t)b * 2 * p.T_out + 2 * t + ph) * Cr + co : ((size_t)b * p.T_out + t) * p.C_out + co;
                float4 v = make_float4(acc[cb][tb][0] + add[cb].x, acc[cb][tb][1] + add[cb].y,
                                       acc[cb][tb][2] + add[cb].z, acc[cb][tb][3] + add[cb].w);
                if (p.flags & TQ_CONV_RES) {
                    const float4 r = *reinterpret_cast<const float4*>(p.res + o);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                *reinterpret_cast<float4*>(p.y + o) = v;
                s1[cb][0] += v.x; s1[cb][1] += v.y; s1[cb][2] += v.z; s1[cb][3] += v.w;
                s2[cb][0] += v.x * v.x; s2[cb][1] += v.y * v.y; s2[cb][2] += v.z * v.z; s2[cb][3] += v.w * v.w;
            }
        }
    }
    if (p.flags & TQ_CONV_STATS) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int co = co_real + cb * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    s1[cb][j] += __shfl_xor(s1[cb][j], o);
                    s2[cb][j] += __shfl_xor(s2[cb][j], o);
                }
            }
            if ((lane & 15) == 0 && slot < p.nslots) {
                float* st = p.stats + (((size_t)b * p.nslots + slot) * Cr + co) * 2;
                *reinterpret_cast<float4*>(st) = make_float4(s1[cb][0], s2[cb][0], s1[cb][1], s2[cb][1]);
                *reinterpret_cast<float4*>(st + 4) = make_float4(s1[cb][2], s2[cb][2], s1[cb][3], s2[cb][3]);
                // range guard: max|y| <= sqrt(sum of squares); (65504 / 2)^2 = 1.0727e9.  NaN / inf fail the comparison too
                const float q = fmaxf(fmaxf(s2[cb][0], s2[cb][1]), fmaxf(s2[cb][2], s2[cb][3]));
                if (p.range_flag && !(q < 1.0727e9f)) *p.range_flag = 1;
            }
        }
    }
    TQ_W4_T(1, 3)
}

template <int KT, int WM, int WN, int ACT, bool FUSE>
int launch_w4(const ConvArgs& a, hipStream_t stream) {
    using C = W4<KT, WM, WN>;
    auto kern = conv1d_w4_kernel<KT, WM, WN, ACT, FUSE>;
    constexpr int LDS_BYTES = C::LDS_BYTES + (ACT >= 1 ? 2 * 4 * 1024 : 0);   // + the coefficient table (C_in <= 1024)
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static std::atomic<uint64_t> attr_done{0};
    int dev_ord = 0;
    (void)hipGetDevice(&dev_ord);
    const uint64_t dev_bit = 1ull << (dev_ord & 63);
    if (!(attr_done.load(std::memory_order_acquire) & dev_bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done.fetch_or(dev_bit, std::memory_order_release);
    }
    const int n_ttiles = (a.T_out + C::NT - 1) / C::NT;
    const unsigned grid = (unsigned)(a.B * n_ttiles * (a.C_out / C::MT));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, stream, a);
    TQ_CHECK_LAUNCH();
    return 0;
}

template <int KT, int ACT, bool FUSE>
int pick_tile(const ConvArgs& a, hipStream_t s) {
    if (a.C_out % 256 == 0) return launch_w4<KT, 4, 1, ACT, FUSE>(a, s);
    if (a.C_out % 128 == 0) return launch_w4<KT, 2, 2, ACT, FUSE>(a, s);
    return TQ_ERR_SHAPE;
}

}  // namespace

#ifdef TQ_W4_STAMP
extern "C" int tq_debug_read_w4_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tq_w4_stamps), sizeof(unsigned long long) * 4 * (n < 4096 ? n : 4096));
}
#endif

namespace tq {
int conv1d_w4_launch(const ConvArgs& a, int ktaps, hipStream_t stream) {
    // Off by default: measured 7-25 % SLOWER per layer than the two-waves-per-SIMD kernel in its present, compiler-scheduled form
    // ("one wave per SIMD"); TQDNE_CONV_W4=1 selects it (tools/w4_check.py A/Bs the two).
    static const int enabled = [] { const char* e = getenv("TQDNE_CONV_W4"); return (e && e[0] == '1') ? 1 : 0; }();
    if (!enabled) return TQ_ERR_SHAPE;
    if (a.wfmt != TQ_WFMT_F16_MX6 || a.kv || a.gf_counters || (a.flags & TQ_CONV_DROPOUT) || ktaps != 5) return TQ_ERR_SHAPE;
    if (a.t_tile) return TQ_ERR_SHAPE;   // (a small-tile plan sized its statistics for 32-position slots: not this kernel's 128)
    if (a.C0 % 64 || a.C1 % 64 || a.sC0 % 64 || a.sC1 % 64 || a.C0 + a.C1 > 1024 || a.C_out % 128) return TQ_ERR_SHAPE;
    if ((a.flags & TQ_CONV_POLY2) || a.T_in != a.T_out) return TQ_ERR_SHAPE;
    const bool gn = a.flags & TQ_CONV_GN, silu = a.flags & TQ_CONV_SILU;
    if (!gn || !silu) return TQ_ERR_SHAPE;   // built for the ResBlocks' GN + SiLU convs
    if (a.sx0) {
        static const int fuse = [] { const char* e = getenv("TQDNE_CONV_W4_FUSE"); return (e && e[0] == '1') ? 1 : 0; }();
        if (!fuse) return TQ_ERR_SHAPE;   // (the fused-skip instantiations still spill: off until they do not)
        return pick_tile<5, 2, true>(a, stream);
    }
    return pick_tile<5, 2, false>(a, stream);
}
}  // namespace tq
