// Fused 1-D convolution for the tqdne UNet on gfx950 (MI355X).
//
//   y[b,t,co] = bias[co] + emb[b,co] + res[b,t,co]
//             + sum_{k,ci} W[co,ci,k] * f(x[b, t*stride + k - pad, ci])
//   f(v) = dropout(SiLU(gscale[b,ci]*v + gshift[b,ci]))      (each stage optional)
//
// replaces, per call, the reference's GroupNorm32 -> SiLU -> Dropout -> Conv1d -> (+emb) -> (+skip)
// chain (tqdne/unet.py:86-102,131-143; tqdne/blocks.py:56-66,92-101,127-145), the channel concat
// of unet.py:396 (two source tensors, never materialised) and F.interpolate(nearest, x2) of
// blocks.py:63 (folded into the gather index).  It also emits per-channel partial sums
// (sum, sum of squares) of y per 128-position slot so the *next* GroupNorm needs no pass over y.
//
// Mapping to CDNA4:
//   * implicit GEMM D[co][t] += W[co][ci,k] * X[ci,k][t] on v_mfma_f32_16x16x32_bf16, fp32 operands split
//     into bf16 hi/lo and multiplied as hi*hi + hi*lo + lo*hi (fp32 accumulate): gfx950 has no TF32/xf32,
//     and exact-f32 MFMA is 16x slower than bf16 MFMA.
//   * A operand (weights) is pre-packed per lane (tq_pack_conv_weight) and streamed straight from L2 into
//     registers: waves of a workgroup are split along co, so no wave re-reads another wave's weights.
//   * B operand (activations) is staged through LDS once per 32-channel chunk: 16-byte coalesced fp32
//     loads of channels-last rows, GN/SiLU/dropout/split in registers, 8-byte LDS stores into an
//     XOR-swizzled [row][4 x 16B] image that the ds_write_b64 stores and the ds_read_b128 fragment reads hit
//     conflict-free for any tap shift.  The K taps are row shifts of the same LDS image (no im2col).
//   * double-buffered LDS, global loads for chunk c+1 issued before the MFMAs of chunk c.
//   * epilogue: accumulator lane = 4 consecutive co at one t -> one 16-byte store per 16x16 tile.
//
// This header holds the kernel template and its tile dispatch; it is compiled into several translation units (one per family of
// instantiations: conv1d_fwd_k5a.hip, conv1d_fwd_k5b.hip, conv1d_fwd_k13.hip, conv1d_resample.hip, conv1d_dgrad.hip) so that the
// library builds in parallel; conv1d_mfma.hip holds the C ABI and the weight packers.
#pragma once
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "conv_args.hpp"
#include "gn_fold.hpp"
#include <atomic>
#include "../../include/tqdne_hip.h"

using namespace tq;

#ifdef TQ_STAMP
// diagnostic build only: per-phase wave-cycle sums (never compiled into the shipped library)
static __device__ unsigned long long tq_stamps[12];
#ifdef TQ_STAMP_OWNER
extern "C" int tq_debug_read_stamps(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(tq_stamps), sizeof(unsigned long long) * 12);
    if (e != hipSuccess) return (int)e;
    if (reset) {
        unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(tq_stamps), z, sizeof(z));
    }
    return (int)e;
}
#endif
// per-workgroup timeline (thread 0; plain stores, no atomics): [entry, loop begin, loop end, exit] as s_memrealtime (100 MHz)
// in [0..3] and s_memtime (shader clock) in [4..7] -- in-kernel clock = d s_memtime / d s_memrealtime
static __device__ unsigned long long tq_timeline[4096 * 8];
#ifdef TQ_STAMP_OWNER
extern "C" int tq_debug_read_timeline(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tq_timeline), sizeof(unsigned long long) * 8 * n);
}
#endif
#define TQ_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#else
#define TQ_T(x)
#endif

namespace {

#ifndef TQ_STAGE_PRE
#define TQ_STAGE_PRE 5
#endif

// SCH (contraction scheme of the fp32 product x * w; both operands arrive as fp32):
//   0  "bf16x3": x = xh + xl, w = wh + wl in bf16; xh*wh + xh*wl + xl*wh on v_mfma_f32_16x16x32_bf16, 32-channel chunks.
//   1  "f16+mx8": per 64-channel chunk two v_mfma_f32_16x16x32_f16 on xh = fp16(x), wh = fp16(w), plus ONE block-scaled fp8 MFMA
//      (v_mfma_scale_f32_16x16x128_f8f6f4) for both first-order corrections: a lane's bytes 0..15 carry fp8(xl * 2^12) against
//      fp8(w), bytes 16..31 fp8(x) against fp8(wl * 2^12), uniform E8M0 scales 2^0 (A) and 2^-12 (B).  The corrections are 2^-12
//      of the product, so fp8's 2^-4 leaves ~2^-15 like bf16x3, at 2/3 of its MFMA cycles.  fp16 RANGE applies to x (|x| > 65504
//      becomes inf in the output, relative precision is lost below 6e-5): forward activations only.
//   2  "f16+mx6": as 1, but the correction operands are e2m3 (fp6) with a per-lane E8M0 block scale (one per 16 channels x
//      {x_l * 2^12, x}), produced by v_cvt_scalef32_2xpk16_fp6_f32: the block-scaled MFMA then takes 4 passes instead of 8 (12
//      instead of 16 per 64 channels) and the corrections no longer clamp (fp8 with uniform scales saturates beyond |x| = 448 and
//      |x_l| = 0.109): measured 5e-5 against 7e-4 of the output scale on heavy-tailed data (tools/micro/fp6_scheme_probe.hip).
//      Staging layout: a thread owns 16 consecutive channels of a row (= exactly one lane fragment of the correction operand).
// TBW: 16-position blocks per wave along t (8, or 4 for the "slim" 64-channel tile: half the accumulators, so that three or four
// workgroups share a CU and their load / MFMA / store phases interleave -- see dispatch_tile)
// NCB: 16-channel blocks per wave (2; 4 for the round-4 tile whose waves are 64 channels x 64 positions, see conv1d_mfma_kernel)
template <int KT, int STRIDE, int UPS, int WM, int WN, int SCH = 0, int TBW = 8, int NCB = 2>
struct Cfg {
    static constexpr int CH = SCH ? 64 : 32;     // channels per chunk
    static constexpr int ROWB = 2 * CH;          // bytes per row of one LDS plane
    static constexpr int TPR = (SCH == 2) ? 4 : CH / 4;  // staging threads per row (4 channels each; scheme 2: 16 channels each)
    static constexpr int NWB = SCH ? 4 : 2;      // 16-byte weight fragments per lane, 16-channel block and (chunk, tap)
    static constexpr int NW = NWB * NCB;         // ... per wave
    static constexpr int NBF = SCH ? 4 : 2;      // 16-byte activation fragments per lane and (tap, t-block)
    static constexpr int NTHR = 64 * WM * WN;
    static constexpr int WT = 16 * TBW;  // output positions per wave
    static constexpr int NT = WT * WN;   // output positions per workgroup
    static constexpr int MT = 16 * NCB * WM;   // output channels per workgroup
    static constexpr int ROWS = (STRIDE == 1) ? (NT + KT - 1) : (2 * NT + 1);
    static constexpr int NIT = (ROWS * TPR + NTHR - 1) / NTHR;
    // staging iterations prefetched into registers across the MFMA phase.  The 4-wave tile of scheme 1 stages twice as many rows
    // per thread as the 8-wave one and has no registers left for them: with 5 in flight it spilled 36 registers into the chunk
    // loop, with 1 (the rest loaded + written in batches after the MFMAs, covered by the co-resident workgroup) none: -4 ... -10 %
    static constexpr int PRE_MAX = (SCH == 1 && WM == 4) ? 1 : TQ_STAGE_PRE;
    static constexpr int PRE = NIT < PRE_MAX ? NIT : PRE_MAX;
    static constexpr int SYNC_BATCH = 4;           // the rest is loaded+written synchronously in batches
    static constexpr int ITERS = (NIT <= PRE) ? NIT : PRE + ((NIT - PRE + SYNC_BATCH - 1) / SYNC_BATCH) * SYNC_BATCH;
    // scheme 2 stages rows [0, NT) in NFULL full iterations of NTHR 16-channel tasks and the KT - 1 halo rows in one predicated step
    static constexpr int NFULL = (SCH == 2) ? (NT * 4) / NTHR : 0;
    static constexpr int HALO_TASKS = (SCH == 2) ? (ROWS - NT) * 4 : 0;
    static constexpr int ROWS_PAD = (SCH == 2) ? ROWS : (ITERS * NTHR + TPR - 1) / TPR;  // every staging task lands in-bounds: no predicate
    static constexpr int PLANE = (ROWS_PAD > ROWS ? ROWS_PAD : ROWS) * ROWB;  // bytes per plane
    static constexpr int BUF = 2 * PLANE;         // hi + lo
    static constexpr int LDS_BYTES = 2 * BUF;     // double buffered
    static constexpr int PAD = (STRIDE == 1) ? (KT / 2) : 1;
};

// ACT (compile time): prologue applied while staging -- 0 none, 1 folded GN, 2 GN + SiLU, 3 GN + SiLU + dropout
// FUSE: the ResBlock's 1x1 skip convolution (unet.py:112,143) is accumulated into the same MFMA accumulators as extra
// 32-channel stages read from the block input (plain, centre tap), instead of a separate launch + residual round trip
// PW ("pointwise, input-stationary"; 1x1 convs of the attention block, blocks.py:127-145): the whole input tile (all <= 4 chunks)
// is staged ONCE into its own LDS buffers with every load in flight together, then the workgroup runs over all output-channel
// tiles: no per-chunk barrier / load round trip (a 1x1 chunk has 1/5 of the MFMA work to hide one under) and no re-staging of
// the same rows by 3 channel-tile workgroups (qkv).
template <int KT, int STRIDE, int UPS, int WM, int WN, int EPI, int ACT, bool FUSE, int SCH, bool PW = false, int TBW = 8, int NCB = 2>
__global__ __launch_bounds__(64 * WM * WN, ((TBW == 4 && NCB == 2 && SCH == 0) ? 3 : 2)) void conv1d_mfma_kernel(const ConvArgs p) {
    static_assert(SCH == 0 || STRIDE == 1, "the fp16-range schemes serve stride-1 launches");
    static_assert(EPI != 1 || SCH == 0 || (SCH == 2 && ACT == 0 && !FUSE && !PW), "data gradients: bf16x3, or fp16 + MX-fp6 on a dy scaled into the fp16 range");
    // NCB == 4 (round 4): 256 channels x 128 positions as 4 x 2 waves of 64 channels x 64 positions -- every activation fragment read
    // from LDS feeds twelve MFMAs instead of six (tools/micro/mfma_shape_power.hip: at full load the conv's 4 ds_read_b128 per 6 MFMAs
    // cost a quarter of the matrix rate; 16 reads + 16 weight loads per 48 MFMAs run 13 % faster than 32 + 8)
    static_assert(NCB == 2 || (NCB == 4 && TBW == 4 && WM == 4 && WN == 2 && SCH == 2 && KT > 1 && STRIDE == 1 && UPS == 0 && EPI == 0 && !FUSE && !PW),
                  "64-channel waves: the fp16 + MX-fp6 forward tile of 256 x 128");
    // C64 (round 6): the fp16 + MX-fp6 tile of the 64-channel layers -- 64 channels x 128 positions as 2 x 2 waves of 32 channels x 64
    // positions, two workgroups per CU (the 64-channel ResBlock convs at T = 4096 and the data gradients whose input-channel count is a
    // multiple of 64 but not of 128 ran in bf16x3: twice the matrix work)
    constexpr bool C64 = TBW == 4 && NCB == 2 && SCH == 2 && WM == 2 && WN == 2;
    static_assert(!C64 || (STRIDE == 1 && UPS == 0 && !PW && EPI != 2), "64-channel fp16 + MX-fp6 tile: stride-1 forward and data-gradient launches");
    static_assert(TBW == 8 || (TBW == 4 && NCB == 4) || C64 || (TBW == 4 && SCH == 0 && STRIDE == 1 && UPS == 0 && EPI == 0 && WN == 2 && !PW) ||
                  (TBW == 2 && (SCH == 0 || (SCH == 2 && WM == 4)) && STRIDE == 1 && UPS == 0 && EPI == 0 && WN == 1 && !PW),
                  "slim tile: bf16x3 forward, 2 x 2 waves; small tile (32 positions per workgroup): stride-1 forward, one wave column");
    static_assert(!PW || (KT == 1 && STRIDE == 1 && UPS == 0 && SCH >= 1 && !FUSE && WN == 1 && EPI != 1), "PW: 1x1, fp16-range schemes");
    static_assert(SCH != 2 || WN == 1 || NCB == 4 || C64 || (WN == 2 && WM == 4 && TBW == 8 && STRIDE == 1 && UPS == 0 && EPI == 0 && !PW),
                  "scheme 2 tiles are 128 positions wide (experiment: 128 channels x 256 positions, 8 waves)");
    using C = Cfg<KT, STRIDE, UPS, WM, WN, SCH, TBW, NCB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef TQ_STAMP
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
    const unsigned long long r_entry = __builtin_amdgcn_s_memrealtime();
#endif

#ifdef TQ_EXP_STAGGER
    // experiment (round 5): the workgroups of one launch run in lock-step -- every CU in its load burst, then in its MFMA loop, then in
    // its store burst (DESIGN_LOG.md).  Start them apart: workgroup i sleeps (hash(i) % 8) * p.exp_stagger * 64 cycles first, so that the
    // bursts of one round spread over the round (first-round workgroups only matter: later ones start when a CU frees up).
    if (p.exp_stagger > 0 && blockIdx.x < 256u * (64 * WM * WN == 512 ? 1u : 2u)) {
        const unsigned k = (blockIdx.x * 2654435761u) >> 29;
        for (unsigned i = 0; i < k * (unsigned)p.exp_stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM;
    const int wn = wave / WM;

    const int n_ttiles = (p.T_out + C::NT - 1) / C::NT;
    const int n_ctiles = (p.C_out + C::MT - 1) / C::MT;
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs, so ids i and i + 8 share an L2.  The channel tiles
    // of one (b, t-tile) re-read the same input rows: give them ids 8 apart (same XCD) instead of adjacent (different XCDs).
    int bid = blockIdx.x;
    int ct, tile;
    const int ntile = p.B * n_ttiles;
    if constexpr (PW) {
        ct = 0;  // the workgroup visits every channel tile itself
        tile = bid;
    } else if (n_ctiles > 1 && (ntile & 7) == 0) {
        const int grp = bid / (8 * n_ctiles), within = bid % (8 * n_ctiles);
        ct = within >> 3;
        tile = grp * 8 + (within & 7);
    } else {
        ct = bid % n_ctiles;
        tile = bid / n_ctiles;
    }
    const int tt = tile % n_ttiles;
    const int b = tile / n_ttiles;
    const int t0 = tt * C::NT;
    // Data gradient in the fp16 + MX-fp6 scheme (EPI == 1, SCH == 2): gradients are tiny (1e-6 ... 1e-3 at the bench's loss scale)
    // and fp16 starts losing bits below 6e-5, so dy is staged times an exact power of two 2^k that puts the tensor's largest
    // magnitude (p.in_amax: bit pattern of max|dy|, written by the column-sum pass that precedes every data gradient) into
    // [2^13, 2^14), and the accumulators are multiplied by 2^-k in the epilogue.  Both products are exact (no rounding, |k| <= 126).
    float dy_scale = 1.f, dy_unscale = 1.f;
    if constexpr (EPI == 1 && SCH == 2) {
        if (p.in_amax) {
            uint32_t amx = 0u;   // the maximum arrives spread over TQ_AMAX_WAYS words, one per 128-byte line (see tq_colsum)
#pragma unroll
            for (int w = 0; w < TQ_AMAX_WAYS; ++w) amx = max(amx, p.in_amax[w * TQ_AMAX_STRIDE]);
            const int e = (int)((amx >> 23) & 0xFFu);            // biased exponent of max|dy| (0: zero / denormal, 255: inf / NaN)
            int k = 140 - e;                                      // 2^13 <= max|dy| * 2^k < 2^14
            k = k > 126 ? 126 : (k < -126 ? -126 : k);
            dy_scale = __uint_as_float((unsigned)(127 + k) << 23);
            dy_unscale = __uint_as_float((unsigned)(127 - k) << 23);
        }
    }
    // per-sample key of the dropout hash (scalar unit; only the dropout prologue / the data gradient's dropout chain read it)
    const DropKey dkey = (ACT == 3 || EPI == 1) ? drop_key(p.drop_seed, p.drop_site, (uint32_t)b) : DropKey{0u, 0u};
    int co_wave = ct * C::MT + wm * (16 * NCB);
    const bool wave_active = co_wave < p.C_out;

    const int Cin = p.C0 + p.C1;
    const int nchunks = Cin / C::CH;
    const int nskip = FUSE ? ((p.sC0 + p.sC1) / C::CH) : 0;
    const int nstages = nchunks + nskip;  // stage s < nchunks: main chunk (KT taps); else skip chunk (centre tap)
    const int T_src = UPS ? 2 * p.T_in : p.T_in;  // extent of the (virtually upsampled) input

    // ---- staging bookkeeping: thread owns 4 consecutive channels (m) of rows i = (tid + it*NTHR) / TPR
    const int m = tid % C::TPR;
    const int wslot = m >> 1, whalf = m & 1;  // 16-byte slot (k quarter) and 8-byte half owned by this thread
    // (the fused-skip launch of the 8-wave scheme-1 tile prefetches 4 of its 5 staging iterations and loads the last one after the
    // MFMAs: one float4 less in flight across the chunk loop is what it takes to keep that loop free of spills)
    // (the dropout prologue of the training forward needs more still: 1 + 4)
    constexpr bool TIGHT = SCH == 1 && WM == 8 && C::PRE == 5 && C::NIT == 5;
    constexpr int PRE = (TIGHT && ACT == 3) ? 1 : (TIGHT && FUSE) ? 4 : C::PRE;
    constexpr int SYNC_BATCH = (TIGHT && ACT == 3) ? 4 : (TIGHT && FUSE) ? 1 : C::SYNC_BATCH;
    float4 raw[PRE];
    float4 g_a, g_s;

    auto src_pos = [&](int i) -> int __attribute__((always_inline)) {
        if (STRIDE == 1) return t0 - C::PAD + i;
        // de-interleaved image: rows [0, NT] hold even offsets v=2*idx, rows [NT+1, 2NT] odd offsets
        const int par = (i > C::NT) ? 1 : 0;
        const int idx = i - par * (C::NT + 1);
        return 2 * t0 - C::PAD + 2 * idx + par;
    };

    auto chunk_base = [&](int stage, int& cs) -> const float* __attribute__((always_inline)) {
        const bool sk = FUSE && stage >= nchunks;
        const int cb = (sk ? stage - nchunks : stage) * C::CH;
        const float* a0 = sk ? p.sx0 : p.x0;
        const float* a1 = sk ? p.sx1 : p.x1;
        const int c0 = sk ? p.sC0 : p.C0, c1 = sk ? p.sC1 : p.C1;
        const bool first = cb < c0;
        const float* src = first ? a0 : a1;
        cs = first ? c0 : c1;
        const int coff = first ? cb : cb - c0;
        return src + (size_t)b * p.T_in * cs + coff + 4 * m;
    };

    // branch-free: out-of-range rows load a clamped (valid) row and are zeroed in write_one -- an exec-masked load would put
    // control flow in front of the MFMA phase and make hipcc drain every outstanding load there (s_waitcnt vmcnt(0))
    auto load_one = [&](const float* base, int cs, int it) -> float4 __attribute__((always_inline)) {
        const int i = (tid + it * C::NTHR) / C::TPR;
        int pos = src_pos(i);
        pos = pos < 0 ? 0 : (pos >= T_src ? T_src - 1 : pos);
        const int srow = UPS ? (pos >> 1) : pos;
        return *reinterpret_cast<const float4*>(base + (size_t)srow * cs);
    };

    // transform + split + LDS store of one staged float4: ~12 VALU per element, no branches (rows past the tile go to padding
    // rows of the LDS image; rows outside the signal are multiplied by 0 = the conv's zero padding of the ACTIVATED input)
    auto write_one = [&](int chunk, int buf, int it, const float4& rv) __attribute__((always_inline)) {
        unsigned char* hi_plane = lds + buf * C::BUF;
        unsigned char* lo_plane = hi_plane + C::PLANE;
        const int i = (tid + it * C::NTHR) / C::TPR;
        const int pos = src_pos(i);
        const float msk = (pos >= 0 && pos < T_src) ? 1.f : 0.f;
        float u[4] = {rv.x, rv.y, rv.z, rv.w};
        const bool act = !(FUSE && chunk >= nchunks);  // skip stages stage the raw block input
        if (ACT >= 1 && act) {
            u[0] = fmaf(g_a.x, u[0], g_s.x); u[1] = fmaf(g_a.y, u[1], g_s.y);
            u[2] = fmaf(g_a.z, u[2], g_s.z); u[3] = fmaf(g_a.w, u[3], g_s.w);
        }
        if (ACT >= 2 && act) {
#pragma unroll
            for (int j = 0; j < 4; ++j)  // u * sigmoid(u) = u / (1 + 2^(-u*log2 e)): v_mul, v_exp, v_add, v_rcp, v_mul
                u[j] = u[j] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u[j] * -1.4426950408889634f));
        }
        if (ACT == 3 && act) {
            const int cb = chunk * C::CH;
            const int pc = pos < 0 ? 0 : pos;
            const uint32_t e0 = (uint32_t)pc * (uint32_t)Cin + (uint32_t)(cb + 4 * m);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                u[j] = (drop_hash(dkey, e0 + j) >= p.drop_thresh) ? u[j] * p.drop_scale : 0.f;
        }
        if constexpr (SCH == 0) {
            uint32_t hb[4];
            float lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u[j] *= msk;
                hb[j] = __float_as_uint(u[j]) & 0xFFFF0000u;  // hi = x truncated to bf16; lo (rounded) carries the remainder
                lo[j] = u[j] - __uint_as_float(hb[j]);
            }
            uint2 hv;
            hv.x = (hb[0] >> 16) | hb[1];
            hv.y = (hb[2] >> 16) | hb[3];
            bf16x4 l;
            l[0] = (__bf16)lo[0]; l[1] = (__bf16)lo[1]; l[2] = (__bf16)lo[2]; l[3] = (__bf16)lo[3];
            const int off = i * 64 + ((wslot ^ (((i >> 2) & 1) << 1)) << 4) + (whalf << 3);
            *reinterpret_cast<uint2*>(hi_plane + off) = hv;
            *reinterpret_cast<bf16x4*>(lo_plane + off) = l;
        } else {
            // 128-byte rows of 8 x 16-byte units, unit' = unit ^ (row & 7): conflict-free for ds_read_b128 at every row offset.
            // Main plane: fp16(x) (round to nearest), unit = channel / 8.  Correction plane: unit g (= channel / 16) holds
            // fp8(xl * 2^12) of channels 16g..16g+15 and unit 4 + g holds fp8(x) of the same channels, so a lane group's two
            // reads sit at byte offsets b and b ^ 64 of the row in both planes.
            f16x4 hv;
            float xl[4], xc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u[j] *= msk;
                hv[j] = (_Float16)u[j];  // |x| > 65504 -> inf, NaN stays NaN: out-of-range inputs surface in the output
                xl[j] = __builtin_amdgcn_fmed3f(u[j] - (float)hv[j], -0.109375f, 0.109375f);  // * 2^12 stays inside e4m3 (448)
                xc[j] = __builtin_amdgcn_fmed3f(u[j], -448.f, 448.f);                           // (the conversions do not saturate)
            }
            i16x2 c8 = {0, 0};
            c8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(c8, xl[0], xl[1], 0.000244140625f, false);  // divides by the scale
            c8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(c8, xl[2], xl[3], 0.000244140625f, true);
            int x8 = __builtin_amdgcn_cvt_pk_fp8_f32(xc[0], xc[1], 0, false);
            x8 = __builtin_amdgcn_cvt_pk_fp8_f32(xc[2], xc[3], x8, true);
            const int sw = i & 7;
            const int row = i * 128;
            *reinterpret_cast<f16x4*>(hi_plane + row + ((wslot ^ sw) << 4) + (whalf << 3)) = hv;
            const int g = m >> 2, byte = (m & 3) << 2;
            *reinterpret_cast<i16x2*>(lo_plane + row + ((g ^ sw) << 4) + byte) = c8;
            *reinterpret_cast<int*>(lo_plane + row + (((4 + g) ^ sw) << 4) + byte) = x8;
        }
    };

    // ---- scheme 2 staging: a thread owns the 16 consecutive channels 16 m2 .. 16 m2 + 15 of a row (one lane fragment of the
    // block-scaled operand); folded GroupNorm coefficients come from an LDS table filled once per workgroup (32 of them per
    // thread would not fit in registers across the MFMA phase)
    const int m2 = tid & 3;
    float4 raw2[4];
    float nlog2e = -1.4426950408889634f, n4096 = -4096.f;
    asm volatile("" : "+s"(nlog2e), "+s"(n4096));  // kept in an SGPR: a literal would split the packed multiply into two scalar-literal ones

    float* gtab = reinterpret_cast<float*>(lds + (PW ? 4 * C::BUF : C::LDS_BYTES));   // [Cin] scale, [Cin] shift of sample b
    // bf16x3 small tile with the consumer-side GroupNorm fold (TqConvDesc.gn_fold): the fold runs behind the first chunk's loads and hands
    // that chunk's coefficients over through the table (see FOLD_LATE below); later chunks read what it wrote to global memory
#ifdef TQ_ABL_FOLD_EARLY
    constexpr bool FOLD_LATE0 = false;
#else
    constexpr bool FOLD_LATE0 = SCH == 0 && TBW == 2 && ACT >= 1 && !PW && EPI == 0;
#endif
    const bool fold0_pending = FOLD_LATE0 && p.cf_st0 != nullptr;
    auto chunk_base2 = [&](int stage, int& cs) -> const float* __attribute__((always_inline)) {
        const bool sk = FUSE && stage >= nchunks;
        const int cb = (sk ? stage - nchunks : stage) * C::CH;
        const float* a0 = sk ? p.sx0 : p.x0;
        const float* a1 = sk ? p.sx1 : p.x1;
        const int c0 = sk ? p.sC0 : p.C0, c1 = sk ? p.sC1 : p.C1;
        const bool first = cb < c0;
        cs = first ? c0 : c1;
        return (first ? a0 : a1) + (size_t)b * p.T_in * cs + (first ? cb : cb - c0) + 16 * m2;
    };
    auto load16 = [&](const float* base, int cs, int row, float4 (&v)[4]) __attribute__((always_inline)) {
        int pos = src_pos(row);
        pos = pos < 0 ? 0 : (pos >= T_src ? T_src - 1 : pos);
        const float4* q = reinterpret_cast<const float4*>(base + (size_t)(UPS ? (pos >> 1) : pos) * cs);
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
    };
    auto write16 = [&](int chunk, int buf, int row, const float4 (&v)[4]) __attribute__((always_inline)) {
        unsigned char* hi_plane = lds + buf * C::BUF;
        unsigned char* lo_plane = hi_plane + C::PLANE;
        const int pos = src_pos(row);
        const float msk = (pos >= 0 && pos < T_src) ? 1.f : 0.f;
        float u[16] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w,
                       v[2].x, v[2].y, v[2].z, v[2].w, v[3].x, v[3].y, v[3].z, v[3].w};
        const bool act = !(FUSE && chunk >= nchunks);  // skip stages stage the raw block input
        if (ACT >= 1 && act) {
            const int cb = chunk * C::CH + 16 * m2;
            const float4* ga = reinterpret_cast<const float4*>(gtab + cb);
            const float4* gs = reinterpret_cast<const float4*>(gtab + Cin + cb);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 a = ga[q4], sh = gs[q4];
                u[4 * q4 + 0] = fmaf(a.x, u[4 * q4 + 0], sh.x); u[4 * q4 + 1] = fmaf(a.y, u[4 * q4 + 1], sh.y);
                u[4 * q4 + 2] = fmaf(a.z, u[4 * q4 + 2], sh.z); u[4 * q4 + 3] = fmaf(a.w, u[4 * q4 + 3], sh.w);
            }
        }
        // Per element (this conversion runs in lock-step on both waves of a SIMD with the matrix pipe idle -- tools/stamps.py:
        // 18 % of the chunk loop of the 512 -> 256 layer -- so every VALU instruction counts): packed fp32 ops wherever two
        // elements share an operation; rows outside the signal are zeroed through the sigmoid (1 / (inf + e) = 0) instead of a
        // multiplication by the mask; the fp16 remainder comes from ONE mixed-precision fma on the packed half.
        if (ACT >= 2 && act) {
            const float one = (pos >= 0 && pos < T_src) ? 1.0f : __builtin_inff();
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                f32x2 uu = {u[j], u[j + 1]};
                f32x2 e = uu * nlog2e;   // u * sigmoid(u) = u / (1 + 2^(-u log2 e))
                e = f32x2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)} + one;
                uu = uu * f32x2{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
                u[j] = uu.x; u[j + 1] = uu.y;
            }
        }
        const float mm = (ACT >= 2 && act) ? 1.f : msk * dy_scale;  // (skip stages and un-activated inputs: the plain mask; data gradients: x 2^k)
        if (ACT == 3 && act) {
            const int pc = pos < 0 ? 0 : pos;
            const uint32_t e0 = (uint32_t)pc * (uint32_t)Cin + (uint32_t)(chunk * C::CH + 16 * m2);
            // the 16 keep decisions into a bit mask first (four hash chains at a time: 16 interleaved ones are what made the 64-bit
            // hash of rounds 1-3 spill in this variant), then applied with static indices
            unsigned keep = 0;
#pragma unroll 4
            for (int j = 0; j < 16; ++j)
                keep |= (drop_hash(dkey, e0 + j) >= p.drop_thresh ? 1u : 0u) << j;
#pragma unroll
            for (int j = 0; j < 16; ++j) u[j] = ((keep >> j) & 1u) ? u[j] * p.drop_scale : 0.f;
        }
        f16x8 h0, h1;
        f32x16 xl, xf;
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            f32x2 x = {u[j], u[j + 1]};
            if constexpr (ACT < 2 || FUSE) x = x * mm;
            // |x| > 65504 -> inf, NaN stays NaN: out-of-range inputs surface in the output
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            const f16x2 hp = {(_Float16)x.x, (_Float16)x.y};
            if (j < 8) { h0[j] = hp.x; h0[j + 1] = hp.y; } else { h1[j - 8] = hp.x; h1[j - 7] = hp.y; }
            const f32x2 x4k = x * 4096.f;
            // (x - fp16(x)) * 2^12 = x * 2^12 - fp16(x) * 2^12, exact (both products and their difference are representable):
            // one v_fma_mix_f32 per element straight from the packed half (hipcc's own choice is two conversions + a packed fma)
            float r0, r1;
            const unsigned hpu = __builtin_bit_cast(unsigned, hp);
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hpu), "s"(n4096), "v"(x4k.x));
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hpu), "s"(n4096), "v"(x4k.y));
            xl[j] = r0; xl[j + 1] = r1; xf[j] = x.x; xf[j + 1] = x.y;
            mx = fmaxf(fmaxf(mx, fabsf(r0)), fabsf(x.x));
            mx = fmaxf(fmaxf(mx, fabsf(r1)), fabsf(x.y));
        }
        const unsigned bb = e8m0_block_scale(mx);
        const u32x6 pk = cvt_2xpk16_fp6(xl, xf, __uint_as_float(bb << 23));
        const int sw = row & 7;
        const int ro = row * 128;
        *reinterpret_cast<uint4*>(hi_plane + ro + (((2 * m2) ^ sw) << 4)) = __builtin_bit_cast(uint4, h0);
        *reinterpret_cast<uint4*>(hi_plane + ro + (((2 * m2 + 1) ^ sw) << 4)) = __builtin_bit_cast(uint4, h1);
        *reinterpret_cast<uint4*>(lo_plane + ro + ((m2 ^ sw) << 4)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        // the lane's E8M0 byte for the MFMA, with the 2^-12 of both correction products folded in
        *reinterpret_cast<uint4*>(lo_plane + ro + (((4 + m2) ^ sw) << 4)) = make_uint4(pk[4], pk[5], bb - 12u, 0u);
    };

    // phase 1 (before the MFMAs of the previous chunk): issue the first PRE iterations' loads
    auto stage_load = [&](int chunk) __attribute__((always_inline)) {
        if constexpr (SCH == 2) {
            int cs2;
            const float* base2 = chunk_base2(chunk, cs2);
            load16(base2, cs2, tid >> 2, raw2);
            return;
        }
        int cs;
        const float* base = chunk_base(chunk, cs);
#pragma unroll
        for (int it = 0; it < PRE; ++it) raw[it] = load_one(base, cs, it);
        if (ACT >= 1 && !(fold0_pending && chunk == 0)) {  // folded GroupNorm coefficients of this thread's 4 channels (skip stages: clamped, unused)
            const int cb = (chunk < nchunks ? chunk : nchunks - 1) * C::CH;
            g_a = *reinterpret_cast<const float4*>(p.gscale + (size_t)b * Cin + cb + 4 * m);
            g_s = *reinterpret_cast<const float4*>(p.gshift + (size_t)b * Cin + cb + 4 * m);
        }
    };

    // phase 2 (after them): transform + LDS write; iterations beyond PRE are loaded here in small batches
    auto stage_write = [&](int chunk, int buf) __attribute__((always_inline)) {
        if constexpr (SCH == 2 && C::NFULL == 0) {
            // small tile (32 positions + halo = ROWS <= NTHR / 4 rows): every row, halo included, is one task of the single pass
            static_assert(C::NFULL != 0 || C::ROWS * 4 <= C::NTHR, "small tile: one staging task per thread");
            if ((tid >> 2) < C::ROWS) write16(chunk, buf, tid >> 2, raw2);
            return;
        } else if constexpr (SCH == 2) {
            int cs2;
            const float* base2 = chunk_base2(chunk, cs2);
            // the KT - 1 halo rows: loads issued first by every thread (tasks past the halo re-read its last row: an L1 hit),
            // converted last, by the HALO_TASKS threads that own them -- their latency hides behind the main conversion
            float4 hv4[4];
            if constexpr (C::HALO_TASKS > 0) {
                const int hr = tid >> 2;
                load16(base2, cs2, C::NT + (hr < C::ROWS - C::NT ? hr : C::ROWS - C::NT - 1), hv4);
            }
            write16(chunk, buf, tid >> 2, raw2);
#pragma unroll 1
            for (int it = 1; it < C::NFULL; ++it) {   // (4-wave tile: second half of the rows, loaded here; prefetching it too: neutral)
                float4 t4[4];
                load16(base2, cs2, (tid >> 2) + it * (C::NTHR / 4), t4);
                write16(chunk, buf, (tid >> 2) + it * (C::NTHR / 4), t4);
            }
            if constexpr (C::HALO_TASKS > 0) {
                if (tid < C::HALO_TASKS) write16(chunk, buf, C::NT + (tid >> 2), hv4);
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < PRE; ++it)
            write_one(chunk, buf, it, raw[it]);
        if (C::NIT > PRE) {
            int cs;
            const float* base = chunk_base(chunk, cs);
#pragma unroll 1
            for (int it0 = PRE; it0 < C::NIT; it0 += SYNC_BATCH) {
                float4 tmp[SYNC_BATCH];
#pragma unroll
                for (int j = 0; j < SYNC_BATCH; ++j) tmp[j] = load_one(base, cs, it0 + j);
#pragma unroll
                for (int j = 0; j < SYNC_BATCH; ++j) write_one(chunk, buf, it0 + j, tmp[j]);
            }
        }
    };

    f32x4 acc[NCB][TBW];
    constexpr bool RES_EARLY = !PW && EPI == 0 && !FUSE && KT == 1;
    if constexpr (RES_EARLY) {
        // 1x1 convs (the attention block's x + proj(a): 256 single-round workgroups of load -> 0.4 us of MFMA -> residual load -> store):
        // the residual (TQ_CONV_RES) is the accumulators' START value, so its load round trip hides behind the main loop instead of
        // standing between the last MFMA and the stores: 31 -> 28 us per launch (B = 64, T = 512).  Measured neutral for the k = 5
        // ResBlock convs (their prologue burst grows by what the epilogue saves), which keep the add in the epilogue.  Branch-free -- a
        // wave-uniform branch here would end in s_waitcnt vmcnt(0) at its join: launches without a residual read one valid address.
        const bool has_res = (p.flags & TQ_CONV_RES) && !(p.flags & TQ_CONV_POLY2) && (ct * C::MT + wm * 32) < p.C_out;
        const float* rbase = has_res ? p.res : p.x0;
#pragma unroll
        for (int j = 0; j < TBW; ++j) {
            const int t = t0 + wn * C::WT + j * 16 + (lane & 15);
            const int tc = t < p.T_out ? t : p.T_out - 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int co = ct * C::MT + wm * 32 + i * 16 + 4 * (lane >> 4);
                const size_t o = has_res ? ((size_t)b * p.T_out + tc) * p.C_out + co : 0;
                const float4 r = *reinterpret_cast<const float4*>(rbase + o);
                acc[i][j] = has_res ? f32x4{r.x, r.y, r.z, r.w} : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    } else if constexpr (!PW) {
#pragma unroll
        for (int i = 0; i < NCB; ++i)
#pragma unroll
            for (int j = 0; j < TBW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int pw_c0 = 0;  // PW: first chunk of the pair the MFMA stream works on ("taps" of that stream = chunks = LDS buffers)

    const int kq = lane >> 4;
    const int tl_lane = wn * C::WT + (lane & 15);
    const uint4* wbase = p.wpk + ((size_t)(co_wave >> 4) * C::NWB) * 64 + lane;  // (PW: advanced per channel tile)
    const size_t wstep = (size_t)p.ncob_pad * C::NWB * 64;  // uint4 per (chunk, tap)

    // ---- MFMA phase of one chunk.  Written as straight-line code (no branches inside: hipcc's waitcnt insertion falls back to
    // s_waitcnt vmcnt(0) / lgkmcnt(0) at every control-flow join, which serialises each prefetch with its consumer) and pinned
    // with sched_barrier so that (1) the weight fragments of tap k+2 are requested before the MFMAs of tap k+1 and (2) the
    // LDS reads of t-block tb+1 are in flight under the MFMAs of t-block tb.
    const int last_step = nchunks * KT + nskip - 1;
    // w[cbk * NW/2 + q]: scheme 0: q = 0 hi, 1 lo; scheme 1: q = 0, 1 fp16 fragments of channels [0,32), [32,64) of the chunk,
    // q = 2 | 3 the 32 correction bytes (fp8(w) | fp8(wl * 2^12)) of channels 16 * (lane >> 4) ...
    auto load_w = [&](int step, Frag (&w)[C::NW]) __attribute__((always_inline)) {
        const int st = step < last_step ? step : last_step;  // clamped: the final refills re-read the last fragments
#ifdef TQ_ABL_NOW
        const uint4* wp = wbase + (size_t)(st & 1) * wstep;  // ablation: weights stay L1-resident
#else
        const uint4* wp = wbase + (size_t)st * wstep;
#endif
#pragma unroll
        for (int q = 0; q < C::NW; ++q) w[q].u = wp[q * 64];
    };

    // scheme 0 LDS image: row = 64 B = 4 slots of 16 B (one k-quarter each), slot' = kq ^ (2 * ((row >> 2) & 1)): conflict-free for
    // ds_read_b128 at every row offset (tap shift) and for the 8-byte staging stores.  Row of t-block tb is rowk + 16*tb, which
    // leaves the swizzle term unchanged (both schemes): one address per tap, t-blocks are immediate offsets.
    auto tap_base = [&](int k) -> int __attribute__((always_inline)) {
        int tl = tl_lane;
        if constexpr (SCH >= 1) {
            // recomputed per tap (4 VALU): hoisted out of the chunk loop the per-tap addresses are live across it, get spilled,
            // and each reload waits for vmcnt(0), i.e. for the staging loads in flight
            asm volatile("" : "+v"(tl));
        }
        if constexpr (PW) return tl * 128 + ((kq ^ (tl & 7)) << 4) + (pw_c0 + k) * C::BUF;
        const int rowk = (STRIDE == 1) ? (tl + k) : ((k & 1) * (C::NT + 1) + tl + (k >> 1));
        if constexpr (SCH == 0) return rowk * 64 + ((kq ^ (((rowk >> 2) & 1) << 1)) << 4);
        else return rowk * 128 + ((kq ^ (rowk & 7)) << 4);
    };
    auto read_b = [&](const unsigned char* hi_plane, const unsigned char* lo_plane, int b0, int tb, Frag (&f)[C::NBF])
        __attribute__((always_inline)) {
#ifdef TQ_ABL_NOLDS
        const int toff = 0;
        (void)tb;
#else
        const int toff = tb * 16 * C::ROWB;
#endif
        if constexpr (SCH == 0) {
            f[0].u = *reinterpret_cast<const uint4*>(hi_plane + b0 + toff);
            f[1].u = *reinterpret_cast<const uint4*>(lo_plane + b0 + toff);
        } else {
            f[0].u = *reinterpret_cast<const uint4*>(hi_plane + b0 + toff);          // fp16, channels 8 kq ...
            f[1].u = *reinterpret_cast<const uint4*>(hi_plane + (b0 ^ 64) + toff);   // fp16, channels 32 + 8 kq ...
            f[2].u = *reinterpret_cast<const uint4*>(lo_plane + b0 + toff);          // fp8(xl * 2^12), channels 16 kq ...
            f[3].u = *reinterpret_cast<const uint4*>(lo_plane + (b0 ^ 64) + toff);   // fp8(x), same channels
        }
    };
    // the MFMAs of one (tap, t-block) step for both 16-channel blocks of the wave
    auto mma_step = [&](const Frag (&w)[C::NW], const Frag (&f)[C::NBF], int tb) __attribute__((always_inline)) {
#ifdef TQ_ABL_NOMMA   // (timing ablation, wrong numerics: no matrix instruction; the operands stay live through an empty asm so that the weight loads
                      // and LDS fragment reads that feed them are still issued and waited for)
#pragma unroll
        for (int q = 0; q < C::NW; ++q) asm volatile("" :: "v"(w[q].u.x), "v"(w[q].u.w));
#pragma unroll
        for (int q = 0; q < C::NBF; ++q) asm volatile("" :: "v"(f[q].u.x), "v"(f[q].u.w));
        return;
#endif
        if constexpr (SCH == 0) {
#if defined(TQ_ABL_SCH0_UNITS)   // (timing ablation, wrong numerics: the forward bf16x3 launches with 1 or 2 of their 3 products -- what a
                                 // cheaper contraction scheme for the 64-channel layers could save at most)
            if constexpr (EPI == 0) {
#pragma unroll
                for (int cbk = 0; cbk < 2; ++cbk) {
                    if (TQ_ABL_SCH0_UNITS >= 2) acc[cbk][tb] = mfma_bf16(w[cbk * 2].v, f[1].v, acc[cbk][tb]);
                    acc[cbk][tb] = mfma_bf16(w[cbk * 2].v, f[0].v, acc[cbk][tb]);
                }
                return;
            }
#endif
#pragma unroll
            for (int cbk = 0; cbk < 2; ++cbk)
                acc[cbk][tb] = mfma_x3(w[cbk * 2].v, w[cbk * 2 + 1].v, f[0].v, f[1].v, acc[cbk][tb]);
        } else if constexpr (SCH == 2) {
            // fp6 fragments: dwords 0..5 of the i32x8 (the last two are ignored for 6-bit operands); the lane's E8M0 byte travels
            // in dword 6 of its own fragment (weights: from the packer; activations: from write16, with the 2^-12 folded in).  (Reading
            // dwords 4, 5 and the scale with an 8- and a 4-byte LDS read makes the six operand registers contiguous without the
            // two v_mov per step hipcc inserts here, but those reads are 2-way bank-conflicted in this image: measured equal.)
            const i32x8 bc = {(int)f[2].u.x, (int)f[2].u.y, (int)f[2].u.z, (int)f[2].u.w,
                              (int)f[3].u.x, (int)f[3].u.y, (int)f[3].u.z, (int)f[3].u.w};
            const f16x8 b0v = __builtin_bit_cast(f16x8, f[0].u), b1v = __builtin_bit_cast(f16x8, f[1].u);
#pragma unroll
            for (int cbk = 0; cbk < NCB; ++cbk)
                acc[cbk][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[cbk * 4 + 0].u), b0v, acc[cbk][tb], 0, 0, 0);
#pragma unroll
            for (int cbk = 0; cbk < NCB; ++cbk)
                acc[cbk][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[cbk * 4 + 1].u), b1v, acc[cbk][tb], 0, 0, 0);
#pragma unroll
            for (int cbk = 0; cbk < NCB; ++cbk) {
                const Frag& c0 = w[cbk * 4 + 2];
                const Frag& c1 = w[cbk * 4 + 3];
                const i32x8 ac = {(int)c0.u.x, (int)c0.u.y, (int)c0.u.z, (int)c0.u.w, (int)c1.u.x, (int)c1.u.y, (int)c1.u.z, (int)c1.u.w};
                acc[cbk][tb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ac, bc, acc[cbk][tb], 2, 2, 0, (int)c1.u.z, 0, (int)f[3].u.z);
            }
        } else {
            const i32x8 bc = {(int)f[2].u.x, (int)f[2].u.y, (int)f[2].u.z, (int)f[2].u.w,
                              (int)f[3].u.x, (int)f[3].u.y, (int)f[3].u.z, (int)f[3].u.w};
            const f16x8 b0v = __builtin_bit_cast(f16x8, f[0].u), b1v = __builtin_bit_cast(f16x8, f[1].u);
            // Issue order: the four short fp16 MFMAs first (dependent ones two apart), the two long block-scaled ones last.  With
            // the scaled MFMA first, the second accumulator's fp16 MFMA had to wait for it (16 passes against 4 of cover).
#pragma unroll
            for (int cbk = 0; cbk < 2; ++cbk)
                acc[cbk][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[cbk * 4 + 0].u), b0v, acc[cbk][tb], 0, 0, 0);
#pragma unroll
            for (int cbk = 0; cbk < 2; ++cbk)
                acc[cbk][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[cbk * 4 + 1].u), b1v, acc[cbk][tb], 0, 0, 0);
#pragma unroll
            for (int cbk = 0; cbk < 2; ++cbk) {
                const Frag& c0 = w[cbk * 4 + 2];
                const Frag& c1 = w[cbk * 4 + 3];
                const i32x8 ac = {(int)c0.u.x, (int)c0.u.y, (int)c0.u.z, (int)c0.u.w, (int)c1.u.x, (int)c1.u.y, (int)c1.u.z, (int)c1.u.w};
                // E8M0 scales: A 127 (2^0), B 115 (2^-12) on every lane: both correction products carry 2^-12
#ifdef TQ_ABL_FP6TIME
                // ablation (wrong numerics): the same stream with the block-scaled MFMA at its fp6 issue rate (4 passes instead of 8)
                acc[cbk][tb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ac, bc, acc[cbk][tb], 2, 2, 0, 127, 0, 115);
#else
                acc[cbk][tb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ac, bc, acc[cbk][tb], 0, 0, 0, 127, 0, 115);
#endif
            }
        }
    };

#ifndef TQ_LDS_DEPTH
#define TQ_LDS_DEPTH 2
#endif
#ifndef TQ_LDS_DEPTH1
#define TQ_LDS_DEPTH1 1
#endif
    // Weight fragments live in two register buffers (taps alternate a, b, a, ...); after tap k its buffer is refilled with
    // tap k+2 of this chunk or, wrapping, with the next chunk's tap of the same parity, so every chunk starts with a = tap 0,
    // b = tap 1 already in flight.
    Frag wa[C::NW], wb[NCB == 4 ? 1 : C::NW];   // (64-channel waves: ONE buffer, see mma_stream1)
#ifndef TQ_LDS_DEPTH2
#define TQ_LDS_DEPTH2 1
#endif
    // scheme 1: 128 MFMA cycles per step, and registers are tight; scheme 2: 96 cycles per step, ~225 registers
    constexpr int LDS_DEP = SCH == 2 ? TQ_LDS_DEPTH2 : (SCH ? TQ_LDS_DEPTH1 : TQ_LDS_DEPTH);

    // MFMA phase of one chunk = ONE stream of NTAPS x 8 (tap, t-block) steps.  B fragments are read LDS_DEP steps ahead of the
    // MFMAs that consume them, across tap boundaries too (a per-tap restart exposed the LDS latency KT times per chunk).
    auto mma_stream = [&](const unsigned char* hi_plane, const unsigned char* lo_plane, int s0, auto ntaps_c, auto first_tap_c)
        __attribute__((always_inline)) {
        if constexpr (NCB == 2) {
        constexpr int NTAPS = decltype(ntaps_c)::value, K0 = decltype(first_tap_c)::value;
        constexpr int DEP = LDS_DEP, NB = LDS_DEP + 1, NS = NTAPS * TBW;
        Frag bf[NB][C::NBF];
        int b0 = tap_base(K0), b0n = b0;
#pragma unroll
        for (int t = 0; t < DEP; ++t) read_b(hi_plane, lo_plane, b0, t, bf[t]);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int kk = st / TBW, tb = st % TBW;
            if (tb == 0 && kk + 1 < NTAPS) b0n = tap_base(K0 + kk + 1);
            const int sn = st + DEP;  // step whose fragments are requested now
            if (sn < NS) read_b(hi_plane, lo_plane, (sn / TBW) == kk ? b0 : b0n, sn % TBW, bf[sn % NB]);
            __builtin_amdgcn_sched_barrier(0);
            if (kk & 1) mma_step(wb, bf[st % NB], tb); else mma_step(wa, bf[st % NB], tb);
            __builtin_amdgcn_sched_barrier(0);
            if (tb == TBW - 1) {  // tap done: refill its weight buffer two steps of the (chunk, tap) sequence ahead
                const int nxt = (NTAPS == 1) ? (s0 + 1) : ((kk + 2 < NTAPS) ? (s0 + kk + 2) : ((kk & 1) ? (s0 + NTAPS + 1) : (s0 + NTAPS)));
                if (kk & 1) load_w(nxt, wb); else load_w(nxt, wa);
                b0 = b0n;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
    };

    // 64-channel waves (NCB == 4): the KT x TBW steps of a chunk with the tap's sixteen weight fragments in ONE register buffer.  A
    // block's four fragments are replaced by the next (chunk, tap) step's as soon as the tap's last t-block has issued that block's
    // MFMAs; the wave then waits for them at the next tap's first step (hipcc's counted vmcnt) -- a stall of one L2 round trip per
    // tap that the SIMD's other wave covers with its own MFMAs (the older wave of a SIMD wins MFMA issue, so the two alternate by
    // themselves; a second buffer would be 64 more registers).  Activation fragments are read one step ahead.
    auto mma_block2 = [&](const Frag (&w)[C::NW], const Frag (&f)[C::NBF], int tb, int cbk) __attribute__((always_inline)) {
        const i32x8 bc = {(int)f[2].u.x, (int)f[2].u.y, (int)f[2].u.z, (int)f[2].u.w,
                          (int)f[3].u.x, (int)f[3].u.y, (int)f[3].u.z, (int)f[3].u.w};
        acc[cbk][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[cbk * 4 + 0].u), __builtin_bit_cast(f16x8, f[0].u), acc[cbk][tb], 0, 0, 0);
        acc[cbk][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[cbk * 4 + 1].u), __builtin_bit_cast(f16x8, f[1].u), acc[cbk][tb], 0, 0, 0);
        const Frag& c0 = w[cbk * 4 + 2];
        const Frag& c1 = w[cbk * 4 + 3];
        const i32x8 ac = {(int)c0.u.x, (int)c0.u.y, (int)c0.u.z, (int)c0.u.w, (int)c1.u.x, (int)c1.u.y, (int)c1.u.z, (int)c1.u.w};
        acc[cbk][tb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ac, bc, acc[cbk][tb], 2, 2, 0, (int)c1.u.z, 0, (int)f[3].u.z);
    };
    auto mma_stream1 = [&](const unsigned char* hi_plane, const unsigned char* lo_plane, int s0) __attribute__((always_inline)) {
        if constexpr (NCB == 4 && SCH == 2) {
            constexpr int NS = KT * TBW;
            Frag bf[2][C::NBF];
            int b0 = tap_base(0), b0n = b0;
            read_b(hi_plane, lo_plane, b0, 0, bf[0]);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const int kk = st / TBW, tb = st % TBW;
                if (tb == 0 && kk + 1 < KT) b0n = tap_base(kk + 1);
                if (st + 1 < NS) read_b(hi_plane, lo_plane, ((st + 1) / TBW) == kk ? b0 : b0n, (st + 1) % TBW, bf[(st + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (tb < TBW - 1) {
                    mma_step(wa, bf[st & 1], tb);
                } else {
                    const int nx = s0 + kk + 1 < last_step ? s0 + kk + 1 : last_step;   // (the final refill re-reads the last fragments)
                    const uint4* wp = wbase + (size_t)nx * wstep;
#pragma unroll
                    for (int cbk = 0; cbk < NCB; ++cbk) {
                        mma_block2(wa, bf[st & 1], tb, cbk);
#ifndef TQ_ABL_NCB4_NOREFILL   // (ablation, wrong numerics: the tap's weights are never replaced = a weight prefetch that costs nothing)
#pragma unroll
                        for (int q = 0; q < 4; ++q) wa[cbk * 4 + q].u = wp[(cbk * 4 + q) * 64];
#endif
                    }
                    b0 = b0n;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    auto compute = [&](int chunk, int buf) __attribute__((always_inline)) {
        const unsigned char* hi_plane = lds + buf * C::BUF;
        if constexpr (NCB == 4) mma_stream1(hi_plane, hi_plane + C::PLANE, chunk * KT);
        else mma_stream(hi_plane, hi_plane + C::PLANE, chunk * KT, std::integral_constant<int, KT>{}, std::integral_constant<int, 0>{});
    };

    // skip stage j: one (centre) tap.  Buffer a holds this step's weights and b the next one's (the last main chunk's
    // wrap-around refills fetched skip steps 0 / 1 as "next chunk, taps 0 / 1"); b is shifted into a afterwards and refilled:
    // every buffer access stays statically indexed (a pointer select between a and b would demote both to scratch)
    auto compute_skip = [&](int j, int buf) __attribute__((always_inline)) {
        const unsigned char* hi_plane = lds + buf * C::BUF;
        const unsigned char* lo_plane = hi_plane + C::PLANE;
        constexpr int DEP = LDS_DEP, NB = LDS_DEP + 1;
        Frag bf[NB][C::NBF];
        const int b0 = tap_base(C::PAD);
#pragma unroll
        for (int t = 0; t < DEP; ++t) read_b(hi_plane, lo_plane, b0, t, bf[t]);
#pragma unroll
        for (int tb = 0; tb < TBW; ++tb) {
            if (tb + DEP < TBW) read_b(hi_plane, lo_plane, b0, tb + DEP, bf[(tb + DEP) % NB]);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(wa, bf[tb % NB], tb);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NCB == 2) {
#pragma unroll
            for (int q = 0; q < C::NW; ++q) wa[q] = wb[q];
            load_w(nchunks * KT + j + 2, wb);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // Folded GroupNorm coefficients of sample b -> LDS table (read by write16): C_in <= 1024 floats each, i.e. at most ONE float4 of
    // either array per thread (scheme-2 tiles have >= 256 threads).  Round 5: the two loads are issued HERE and stored to LDS only after
    // the first chunk's staging loads and the first weight fragments have been requested (gtab_store below), so that the prologue
    // pays one global round trip, not two in a row (table, barrier, then the chunk).  Branch-free: threads past the table repeat
    // its last entry (same value to the same slot).
    static_assert(!(SCH == 2 && ACT >= 1) || C::NTHR >= 256, "one table entry per thread");
    // Round 6, consumer-side GroupNorm fold (TqConvDesc.gn_fold; small tile only): the workgroup forms the folded coefficients of ITS sample
    // from the source tensors' partial statistics here -- the arithmetic of tq_gn_finalize (gn_fold_sample: bit-identical coefficients for
    // any workgroup size) -- writes them where the prologue below (and, later, the backward) reads them, and goes on; every workgroup of
    // a sample writes the same bits.  Replaces the tq_gn_finalize launch in front of this one: a launch-bound plan (<= 4 samples: ~100
    // dependent launches of 5-30 us) loses 45 % of its launches for ~3 us more in the prologue of each conv.
#ifdef TQ_ABL_FOLD_EARLY   // (A/B build: the small tile of scheme 2 folds in front of its loads too, as until the end of round 6)
    constexpr bool FOLD_EARLY = TBW == 2 && ACT >= 1 && !PW && EPI == 0;
#else
    constexpr bool FOLD_EARLY = TBW == 2 && SCH != 2 && ACT >= 1 && !PW && EPI == 0 && !FOLD_LATE0;   // (both schemes of the small tile fold behind their first loads, below)
#endif
    if constexpr (FOLD_EARLY) {
        if (p.cf_st0) {   // (uniform over the launch)
            gn_fold_sample<false, false>(reinterpret_cast<double*>(lds), b, p.cf_st0, p.C0, p.cf_st1, p.C1, p.T_in, p.cf_ns0, p.cf_ns1, p.cf_gamma,
                                         p.cf_beta, const_cast<float*>(p.gscale), const_cast<float*>(p.gshift), p.cf_mean_rstd);
            __syncthreads();   // the coefficients (global) are read back by other threads; the LDS scratch becomes the staging buffers
        }
    }
    // ... and in the fp16 + MX-fp6 scheme (the small tile; the default tiles as an experiment, TQDNE_GN_FOLD=1): the fold runs AFTER the first
    // chunk's staging loads and the first weight fragments have been requested (gtab_store below), so that its statistics loads share
    // their round trip, and leaves the coefficients in the LDS table directly -- what it adds to a workgroup is its arithmetic and three
    // barriers, not a global round trip (the small tile folded in FRONT of its loads until the end of round 6: two round trips in a row
    // in launches that are one dependent chain of latencies).
    constexpr bool FOLD_LATE = SCH == 2 && ACT >= 1 && !PW && EPI == 0 && (TBW == 8 || TBW == 2) && !FOLD_EARLY;
    const bool fold_late = FOLD_LATE && p.cf_st0 != nullptr;
    float4 gt_a = make_float4(0.f, 0.f, 0.f, 0.f), gt_s = gt_a;
    int gt_i = 0;
    if constexpr (SCH == 2 && ACT >= 1) {
        const int n4 = Cin >> 2;
        gt_i = tid < n4 ? tid : n4 - 1;
        if (!fold_late) {
            gt_a = reinterpret_cast<const float4*>(p.gscale + (size_t)b * Cin)[gt_i];
            gt_s = reinterpret_cast<const float4*>(p.gshift + (size_t)b * Cin)[gt_i];
        }
    }
    auto gtab_store = [&]() __attribute__((always_inline)) {
        if constexpr (SCH == 2 && ACT >= 1) {
            if constexpr (FOLD_LATE) {
                if (fold_late) {   // (uniform over the launch)
                    gn_fold_sample<false, false>(reinterpret_cast<double*>(lds), b, p.cf_st0, p.C0, p.cf_st1, p.C1, p.T_in, p.cf_ns0, p.cf_ns1,
                                                 p.cf_gamma, p.cf_beta, const_cast<float*>(p.gscale), const_cast<float*>(p.gshift), p.cf_mean_rstd,
                                                 gtab, gtab + Cin);
                    __syncthreads();   // table complete; the scratch at the start of LDS becomes staging buffer 0
                    return;
                }
            }
            float4* g4 = reinterpret_cast<float4*>(gtab);
            g4[gt_i] = gt_a;
            g4[(Cin >> 2) + gt_i] = gt_s;
            __syncthreads();
        }
    };
    if constexpr (PW) gtab_store();   // (the input-stationary form stages all its chunks in one pass right below)
    const int npass = PW ? n_ctiles : 1;
    if constexpr (PW) {
        // every chunk's loads in flight together (64 + 32 registers, nothing else is live yet), then one transform + store pass
        static_assert(!PW || SCH == 2 || C::NIT == C::PRE, "PW stages a chunk in PRE iterations");
        static_assert(!PW || SCH != 2 || (C::NFULL == 1 && C::HALO_TASKS == 0), "PW, scheme 2: one 16-channel task per thread and chunk");
        if constexpr (SCH == 2) {
            float4 rr2[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                int cs2;
                const float* base2 = chunk_base2(c < nchunks ? c : nchunks - 1, cs2);
                load16(base2, cs2, tid >> 2, rr2[c]);
            }
            load_w(0, wa);
            load_w(1, wb);
#pragma unroll
            for (int c = 0; c < 4; ++c) write16(c < nchunks ? c : nchunks - 1, c, tid >> 2, rr2[c]);
            __syncthreads();
        } else {
        float4 rr[4][C::PRE], ga4[4], gs4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int cc = c < nchunks ? c : nchunks - 1;  // (fewer than 4 chunks: the spare buffers get a copy of the last one)
            int cs;
            const float* base = chunk_base(cc, cs);
#pragma unroll
            for (int it = 0; it < C::PRE; ++it) rr[c][it] = load_one(base, cs, it);
            if (ACT >= 1) {
                ga4[c] = *reinterpret_cast<const float4*>(p.gscale + (size_t)b * Cin + cc * C::CH + 4 * m);
                gs4[c] = *reinterpret_cast<const float4*>(p.gshift + (size_t)b * Cin + cc * C::CH + 4 * m);
            }
        }
        load_w(0, wa);
        load_w(1, wb);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (ACT >= 1) { g_a = ga4[c]; g_s = gs4[c]; }
#pragma unroll
            for (int it = 0; it < C::PRE; ++it) write_one(c < nchunks ? c : nchunks - 1, c, it, rr[c][it]);
        }
        __syncthreads();
        }
    }
    int pass = 0;
next_pass:  // (PW only: a loop statement here costs the other instantiations registers)
    if constexpr (PW) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j = 0; 2 * j < nchunks; ++j) {
            pw_c0 = 2 * j;
            mma_stream(lds, lds + C::PLANE, 2 * j, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
        }
        // the next channel tile's first weights are requested BEFORE this tile's stores: vmcnt retires in order, behind the
        // stores they would only arrive once the whole output tile has drained
        if (pass + 1 < npass) wbase += (size_t)(C::MT >> 4) * C::NWB * 64;
        load_w(0, wa);
        load_w(1, wb);
        __builtin_amdgcn_sched_barrier(0);
    } else {
        // ---- main loop over the stages (32-channel chunks of the conv input, then of the fused skip input)
        stage_load(0);
        if (wave_active) {
            load_w(0, wa);
            if constexpr (NCB == 2) {
                if (KT > 1 || nskip > 0) load_w(1, wb);
            }
        }
        gtab_store();
        if constexpr (FOLD_LATE0) {
            if (fold0_pending) {   // (uniform over the launch)
                gn_fold_sample<false, false>(reinterpret_cast<double*>(lds), b, p.cf_st0, p.C0, p.cf_st1, p.C1, p.T_in, p.cf_ns0, p.cf_ns1, p.cf_gamma,
                                             p.cf_beta, const_cast<float*>(p.gscale), const_cast<float*>(p.gshift), p.cf_mean_rstd, gtab, gtab + Cin);
                __syncthreads();   // table complete (it sits behind the staging buffers; the fold's scratch at the start of LDS is free again)
                g_a = *reinterpret_cast<const float4*>(gtab + 4 * m);
                g_s = *reinterpret_cast<const float4*>(gtab + Cin + 4 * m);
            }
        }
        stage_write(0, 0);
        __syncthreads();
#ifdef TQ_STAMP
        unsigned long long s_load = 0, s_mma = 0, s_write = 0, s_bar = 0;
        const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
        if (tid == 0 && blockIdx.x < 4096) {
            unsigned long long* tl = tq_timeline + blockIdx.x * 8;
            tl[0] = r_entry; tl[1] = __builtin_amdgcn_s_memrealtime(); tl[4] = t_entry; tl[5] = t_begin;
        }
#endif
#ifdef TQ_SKEW
        // Half-phase skew of the two waves of a SIMD (8-wave tile: waves w and w + 4 share one): waves 4-7 convert + store their
        // share of chunk c + 1 BEFORE their MFMA phase of chunk c (from loads issued one chunk earlier), waves 0-3 after theirs, so
        // that one of the two is in its matrix stream while the other one stages.  Same barriers, same LDS hand-over: at the start
        // of an iteration every wave has left the MFMA phase that read the buffer about to be overwritten.
        // Diagnostic build only (-DTQ_SKEW): parity-green and spill-free in the fp6 layout (232-242 registers), but measured no
        // faster (18-step sample 164.7 vs 165.7 ms, single layers 0-7 % slower): moving staging between the two waves of a SIMD
        // is zero-sum here, as MI355X_MICROARCH.md's two-waves-per-SIMD section predicts.
        if constexpr (!FUSE && SCH == 2 && WM == 8) {
            // (ONE copy of the MFMA stream: two copies behind a wave-uniform branch made hipcc spill ~60 registers)
            const bool skew = wave >= 4;
            if (skew && nstages > 1) stage_load(1);
            for (int c = 0; c + 1 < nstages; ++c) {
                if (skew) stage_write(c + 1, (c + 1) & 1);
                const int nxt = c + (skew ? 2 : 1);
                stage_load(nxt < nstages ? nxt : nstages - 1);
                if (wave_active) compute(c, c & 1);
                if (!skew) stage_write(c + 1, (c + 1) & 1);
                __syncthreads();
            }
            if (wave_active) compute(nstages - 1, (nstages - 1) & 1);
        } else
#endif
        if constexpr (!FUSE) {
            for (int c = 0; c + 1 < nstages; ++c) {
                TQ_T(tA)
#ifndef TQ_ABL_NOSTAGE
                stage_load(c + 1);  // issued before the MFMA phase; the phase's first weight waits concern older loads only
#endif
                TQ_T(tB)
                if (wave_active) {
                    if (!FUSE || c < nchunks) compute(c, c & 1);
                    else compute_skip(c - nchunks, c & 1);
                }
                TQ_T(tC)
#ifndef TQ_ABL_NOSTAGE
                stage_write(c + 1, (c + 1) & 1);
#endif
                TQ_T(tD)
                __syncthreads();
                TQ_T(tE)
#ifdef TQ_STAMP
                s_load += tB - tA; s_mma += tC - tB; s_write += tD - tC; s_bar += tE - tD;
#endif
            }
            if (wave_active) {
                if (!FUSE || nskip == 0) compute(nstages - 1, (nstages - 1) & 1);
                else compute_skip(nskip - 1, (nstages - 1) & 1);
            }
        } else {
            // Fused skip conv: main chunks and skip chunks run in SEPARATE loops (one loop with a per-stage branch between the
            // two MFMA streams cost ~50 registers: both streams' live ranges end up merged across the loop)
            for (int c = 0; c + 1 < nchunks; ++c) {
                stage_load(c + 1);
                if (wave_active) compute(c, c & 1);
                stage_write(c + 1, (c + 1) & 1);
                __syncthreads();
            }
            stage_load(nchunks);  // first skip chunk, under the last main chunk
            if (wave_active) compute(nchunks - 1, (nchunks - 1) & 1);
            stage_write(nchunks, nchunks & 1);
            __syncthreads();
            for (int j = 0; j + 1 < nskip; ++j) {
                const int st = nchunks + j;
                stage_load(st + 1);
                if (wave_active) compute_skip(j, st & 1);
                stage_write(st + 1, (st + 1) & 1);
                __syncthreads();
            }
            if (wave_active) compute_skip(nskip - 1, (nstages - 1) & 1);
        }
#ifdef TQ_STAMP
        if (lane == 0) {
            const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
            if (blockIdx.x == 7) {  // one workgroup's phase sums are enough (atomics from every wave perturb the kernel)
                atomicAdd(&tq_stamps[0], s_load); atomicAdd(&tq_stamps[1], s_mma); atomicAdd(&tq_stamps[2], s_write);
                atomicAdd(&tq_stamps[3], s_bar); atomicAdd(&tq_stamps[4], t_loop - t_begin); atomicAdd(&tq_stamps[5], 1ull);
            }
            if (tid == 0 && blockIdx.x < 4096) {
                unsigned long long* tl = tq_timeline + blockIdx.x * 8;
                tl[2] = __builtin_amdgcn_s_memrealtime(); tl[6] = t_loop;
            }
        }
#endif

    }
    // ---- epilogue
    if (!PW && !wave_active) return;
    if constexpr (EPI == 2) {
        // qkv projection feeding attention_fwd2 (blocks.py:139-145): q stays fp32 in the (B, T, 3 H D) tensor, k (scaled by
        // D^-1/4 like q inside the kernel) and v are written as the bf16 hi / lo planes kv[b][h][K hi, K lo, V hi, V lo][t][d]
        // the attention kernel streams -- instead of fp32 here plus a separate split pass over them (tq_attention_fwd)
        const int HD = p.kvH * p.kvD;
        const size_t plane = (size_t)p.kvTp * p.kvD * 2;  // bytes
        float4 add[2];
#pragma unroll
        for (int cbk = 0; cbk < 2; ++cbk) {
            const int co = co_wave + cbk * 16 + 4 * (lane >> 4);
            add[cbk] = p.bias ? *reinterpret_cast<const float4*>(p.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int sec = co_wave / HD;   // 0 q, 1 k, 2 v: uniform over the wave (its 32 channels lie inside one head: 32 | D)
        if (sec == 0) {
#pragma unroll
            for (int tb = 0; tb < TBW; ++tb) {
                const int t = t0 + wn * C::WT + tb * 16 + (lane & 15);
                if (t < p.T_out) {
#pragma unroll
                    for (int cbk = 0; cbk < 2; ++cbk) {
                        const int co = co_wave + cbk * 16 + 4 * (lane >> 4);
                        *reinterpret_cast<float4*>(p.y + ((size_t)b * p.T_out + t) * p.C_out + co) =
                            make_float4(acc[cbk][tb][0] + add[cbk].x, acc[cbk][tb][1] + add[cbk].y,
                                        acc[cbk][tb][2] + add[cbk].z, acc[cbk][tb][3] + add[cbk].w);
                    }
                }
            }
        } else {
            // k / v planes.  A lane holds 4 channels of each of the wave's two 16-channel blocks: stored as they are, that is two 8-byte
            // pieces per row, plane and lane group -- 32-byte runs, twice the write requests of the fp32 epilogue for the same bytes
            // (measured: this launch 42 us with fp32 stores, 55 us with the split ones).  Lane groups g and g ^ 1 swap one block first:
            // even groups then own channels 4g .. 4g + 7 of block 0, odd ones 16 + 4(g - 1) .. + 7 of block 1 -> ONE 16-byte store per
            // lane, row and plane, 64 contiguous bytes per row.
            const int g = lane >> 4, odd = g & 1;
            const int r = co_wave - sec * HD, h = r / p.kvD, d0 = r - h * p.kvD;
            const float sc = sec == 1 ? p.kvscale : 1.0f;
            const int dch = d0 + (odd ? 16 + 4 * (g - 1) : 4 * g);
            unsigned char* base0 = p.kv + (((size_t)b * p.kvH + h) * 4 + (sec == 1 ? 0 : 2)) * plane + (size_t)dch * 2;
            float vmax = 0.f;   // range guard of the fp16 V planes: max |v| this lane converts
#pragma unroll
            for (int tb = 0; tb < TBW; ++tb) {
                const int t = t0 + wn * C::WT + tb * 16 + (lane & 15);
                uint2 hi[2], lo[2];
#pragma unroll
                for (int cbk = 0; cbk < 2; ++cbk) {
                    const float v[4] = {acc[cbk][tb][0] + add[cbk].x, acc[cbk][tb][1] + add[cbk].y,
                                        acc[cbk][tb][2] + add[cbk].z, acc[cbk][tb][3] + add[cbk].w};
                    if (sec == 2 && p.kv_vf16) {
                        // V: fp16 hi / lo (attention_fwd2_kernel<D, true> multiplies both by ONE fp16 p); |v| > 65504 becomes inf and
                        // surfaces in the output, like out-of-range inputs of the fp16-range convs
                        f16x4 hh, ll;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            hh[j] = (_Float16)v[j];
                            ll[j] = (_Float16)(v[j] - (float)hh[j]);
                            vmax = fmaxf(vmax, fabsf(v[j]));
                        }
                        hi[cbk] = __builtin_bit_cast(uint2, hh);
                        lo[cbk] = __builtin_bit_cast(uint2, ll);
                    } else {
                        bf16x4 hh, ll;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            __bf16 a, c;
                            split_bf16(v[j] * sc, a, c);
                            hh[j] = a; ll[j] = c;
                        }
                        hi[cbk] = __builtin_bit_cast(uint2, hh);
                        lo[cbk] = __builtin_bit_cast(uint2, ll);
                    }
                }
                const uint2 sh = odd ? hi[0] : hi[1], sl = odd ? lo[0] : lo[1];   // the block this lane gives away
                uint2 rh, rl;
                rh.x = __shfl_xor((int)sh.x, 16); rh.y = __shfl_xor((int)sh.y, 16);
                rl.x = __shfl_xor((int)sl.x, 16); rl.y = __shfl_xor((int)sl.y, 16);
                const uint4 oh = odd ? make_uint4(rh.x, rh.y, hi[1].x, hi[1].y) : make_uint4(hi[0].x, hi[0].y, rh.x, rh.y);
                const uint4 ol = odd ? make_uint4(rl.x, rl.y, lo[1].x, lo[1].y) : make_uint4(lo[0].x, lo[0].y, rl.x, rl.y);
                if (t < p.T_out) {
                    unsigned char* dst = base0 + (size_t)t * p.kvD * 2;
                    *reinterpret_cast<uint4*>(dst) = oh;
                    *reinterpret_cast<uint4*>(dst + plane) = ol;
                }
            }
            // (TqConvDesc.range_flag: V within a factor two of the fp16 range, or not finite -> the caller repeats on bf16 planes)
            if (p.range_flag && !(vmax < 32752.f)) *p.range_flag = 1;
        }
    } else if constexpr (EPI == 0) {
    // TQ_CONV_POLY2: the upper half of the (virtual) output channels is phase 1 of an upsampling conv: real channel co - C,
    // real row 2t + 1 of a tensor with twice the rows and half the channels (a wave's 32 channels never straddle the phases)
    const bool poly = p.flags & TQ_CONV_POLY2;
    const int Cr = poly ? (p.C_out >> 1) : p.C_out;
    const int ph = (poly && co_wave >= Cr) ? 1 : 0;
    const int co_real = co_wave - ph * Cr;
    // statistics slot = the 128 positions of a wave column (TBW == 8), of the workgroup (slim tile), or the workgroup's 32 positions
    // (small tile: the host sized the statistics tensor, p.nslots, for 32-position slots)
    const int slot = poly ? 2 * ((t0 >> 7) + wn) + ph : (TBW == 2 ? (t0 >> 5) : (t0 >> 7) + (TBW == 8 ? wn : 0));
    const float* emb_b = (p.flags & TQ_CONV_EMB) ? p.emb + (size_t)b * p.emb_stride : nullptr;
    // The two 16-channel blocks of a wave are the two 64-byte halves of one 128-byte output line: they are stored back to back
    // (t-block outer, channel block inner).  With the channel block as the outer loop the halves reached L2 microseconds apart
    // and PMC showed 1.46x the output bytes written to HBM.
    float4 add[NCB];
    float s1[NCB][4], s2[NCB][4];
#pragma unroll
    for (int cbk = 0; cbk < NCB; ++cbk) {
        const int co = co_real + cbk * 16 + 4 * (lane >> 4);
        add[cbk] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) add[cbk] = *reinterpret_cast<const float4*>(p.bias + co);
        if (emb_b) {
            const float4 e = *reinterpret_cast<const float4*>(emb_b + co);
            add[cbk].x += e.x; add[cbk].y += e.y; add[cbk].z += e.z; add[cbk].w += e.w;
        }
        if (FUSE && p.sbias) {
            const float4 e = *reinterpret_cast<const float4*>(p.sbias + co);
            add[cbk].x += e.x; add[cbk].y += e.y; add[cbk].z += e.z; add[cbk].w += e.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[cbk][j] = 0.f; s2[cbk][j] = 0.f; }
    }
    // Residual (TQ_CONV_RES; RES_EARLY launches took it as the accumulators' start value): ALL of the wave's loads are issued first --
    // branch-free, rows past the tensor clamped to its last row -- and waited for once.  Round 6: inside the store loop each load was
    // followed by s_waitcnt vmcnt(0) (hipcc drains the counter at the loop's joins), which also waits for the PREVIOUS stores: sixteen
    // store-complete + load round trips in a row per wave.
    const bool res_epi = (!RES_EARLY || poly) && (p.flags & TQ_CONV_RES);
    float4 rq[TBW][NCB];
#ifdef TQ_ABL_EPI_SERIAL   // (A/B build: the loads inside the store loop, as in rounds 1-5)
    if (false) {
#else
    if (res_epi) {   // (wave-uniform)
#endif
#pragma unroll
        for (int tb = 0; tb < TBW; ++tb) {
            const int t = t0 + wn * C::WT + tb * 16 + (lane & 15);
            const int tc = t < p.T_out ? t : p.T_out - 1;
#pragma unroll
            for (int cbk = 0; cbk < NCB; ++cbk) {
                const int co = co_real + cbk * 16 + 4 * (lane >> 4);
                const size_t o = poly ? ((size_t)b * 2 * p.T_out + 2 * tc + ph) * Cr + co : ((size_t)b * p.T_out + tc) * p.C_out + co;
                rq[tb][cbk] = *reinterpret_cast<const float4*>(p.res + o);
            }
        }
    }
#pragma unroll
    for (int tb = 0; tb < TBW; ++tb) {
        int t = t0 + wn * C::WT + tb * 16 + (lane & 15);
        // (opaque to the optimiser: the store addresses are recomputed here -- shared with the load batch above they would be sixteen
        // 64-bit values live across it, which hipcc sends to scratch)
        asm volatile("" : "+v"(t));
        if (t < p.T_out) {
#pragma unroll
            for (int cbk = 0; cbk < NCB; ++cbk) {
                const int co = co_real + cbk * 16 + 4 * (lane >> 4);
                const size_t o = poly ? ((size_t)b * 2 * p.T_out + 2 * t + ph) * Cr + co : ((size_t)b * p.T_out + t) * p.C_out + co;
                float4 v = make_float4(acc[cbk][tb][0] + add[cbk].x, acc[cbk][tb][1] + add[cbk].y,
                                       acc[cbk][tb][2] + add[cbk].z, acc[cbk][tb][3] + add[cbk].w);
                if (res_epi) {
#ifdef TQ_ABL_EPI_SERIAL
                    const float4 r = *reinterpret_cast<const float4*>(p.res + o);
#else
                    const float4 r = rq[tb][cbk];
#endif
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
#ifdef TQ_ABL_NOEPI
                if (v.x == 1.2345e30f)  // ablation: keep the arithmetic, drop the stores
#endif
                *reinterpret_cast<float4*>(p.y + o) = v;
                s1[cbk][0] += v.x; s1[cbk][1] += v.y; s1[cbk][2] += v.z; s1[cbk][3] += v.w;
                s2[cbk][0] += v.x * v.x; s2[cbk][1] += v.y * v.y; s2[cbk][2] += v.z * v.z; s2[cbk][3] += v.w * v.w;
            }
        }
    }
    if (p.flags & TQ_CONV_STATS) {
#pragma unroll
        for (int cbk = 0; cbk < NCB; ++cbk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    s1[cbk][j] += __shfl_xor(s1[cbk][j], o);
                    s2[cbk][j] += __shfl_xor(s2[cbk][j], o);
                }
            }
        }
        if constexpr (TBW == 4) {
            // slim tile: the two waves of a channel half cover the two 64-position halves of ONE 128-position statistics slot; the
            // second one hands its sums over through LDS (the staging buffers are idle) and the first one stores the slot's total
            __syncthreads();
            float* red = reinterpret_cast<float*>(lds) + (wm * 4 + (lane >> 4)) * (8 * NCB);   // [wm][kq][cbk][j][2]
            if (wn == 1 && (lane & 15) == 0) {
#pragma unroll
                for (int cbk = 0; cbk < NCB; ++cbk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { red[(cbk * 4 + j) * 2] = s1[cbk][j]; red[(cbk * 4 + j) * 2 + 1] = s2[cbk][j]; }
            }
            __syncthreads();
            if (wn == 0 && (lane & 15) == 0) {
#pragma unroll
                for (int cbk = 0; cbk < NCB; ++cbk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s1[cbk][j] += red[(cbk * 4 + j) * 2]; s2[cbk][j] += red[(cbk * 4 + j) * 2 + 1]; }
            }
        }
#pragma unroll
        for (int cbk = 0; cbk < NCB; ++cbk) {
            const int co = co_real + cbk * 16 + 4 * (lane >> 4);
            if ((lane & 15) == 0 && slot < p.nslots && (TBW == 8 || wn == 0)) {
                float* st = p.stats + (((size_t)b * p.nslots + slot) * Cr + co) * 2;
#ifdef TQ_BUILD_EXPERIMENTS
                if (p.gf_counters) {
                    // fused finalisation: the last-arriving workgroup of sample b reads these pairs in THIS launch -- 8-byte
                    // agent-scope atomic stores (write-through, global_store_dwordx2 sc1), read back with the matching loads
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned long long u = (unsigned long long)__float_as_uint(s1[cbk][j]) |
                                                     ((unsigned long long)__float_as_uint(s2[cbk][j]) << 32);
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(st + 2 * j), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else
#endif
                {
                    *reinterpret_cast<float4*>(st) = make_float4(s1[cbk][0], s2[cbk][0], s1[cbk][1], s2[cbk][1]);
                    *reinterpret_cast<float4*>(st + 4) = make_float4(s1[cbk][2], s2[cbk][2], s1[cbk][3], s2[cbk][3]);
                }
                // range guard: max|y| <= sqrt(sum of squares); (65504 / 2)^2 = 1.0727e9.  NaN / inf fail the comparison too
                const float q = fmaxf(fmaxf(s2[cbk][0], s2[cbk][1]), fmaxf(s2[cbk][2], s2[cbk][3]));
                if (p.range_flag && !(q < 1.0727e9f)) *p.range_flag = 1;
            }
        }
    }
#ifdef TQ_BUILD_EXPERIMENTS
    if constexpr (!PW) {
        if (p.gf_counters) {   // (uniform over the launch; every wave of the workgroup is here: the host refuses ragged channel tiles)
            // GroupNorm finalisation by the last arriver (TqGnFuse): every storing wave drains its stores, the workgroup meets, ONE
            // lane takes the sample's ticket (agent-scope atomic add, returning); the workgroup whose ticket completes the sample
            // -- tickets count up for ever: (ticket + 1) % workgroups-per-sample == 0 -- folds the statistics of that sample.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* flag = reinterpret_cast<int*>(lds);
            if (tid == 0) {
                const unsigned long long t = __hip_atomic_fetch_add(p.gf_counters + b, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = ((t + 1ull) % (unsigned long long)p.gf_narrive) == 0ull ? 1 : 0;
            }
            __syncthreads();
            const bool last = *flag != 0;
            __syncthreads();   // (the flag word is part of the fold's scratch)
            if (last) {
                const int Cown = poly ? Cr : p.C_out;
                const int ns = p.nslots;
                double* sh = reinterpret_cast<double*>(lds);
                if (p.gf_partner_first)
                    gn_fold_sample<false, true>(sh, b, p.gf_partner, p.gf_Cp, p.stats, Cown, poly ? 2 * p.T_out : p.T_out, ns, ns, p.gf_gamma,
                                                p.gf_beta, p.gf_gscale, p.gf_gshift, p.gf_mean_rstd);
                else
                    gn_fold_sample<true, false>(sh, b, p.stats, Cown, p.gf_partner, p.gf_Cp, poly ? 2 * p.T_out : p.T_out, ns, ns, p.gf_gamma,
                                                p.gf_beta, p.gf_gscale, p.gf_gshift, p.gf_mean_rstd);
            }
        }
    }
#endif
    } else {
        // data gradient: acc = d loss / d xhat.  Chain through dropout, SiLU and the folded GroupNorm scale of the
        // FORWARD conv's prologue:  g = acc * mask/(1-p) * silu'(u), u = a*x + s  (the GN statistics' own dependence
        // on x is handled by tq_gn_bwd_finalize / tq_gn_bwd_apply from the partial sums (sum g, sum g*x) emitted here)
        const int slot = (t0 >> 7) + (TBW == 8 ? wn : 0);   // (TBW == 4: the workgroup's 128 positions are ONE slot, see below)
        const int Ctot = p.C_out;
        // as in the forward epilogue: t-block outer, channel block inner, so the two halves of a 128-byte line are stored together
        float* dst[2]; const float* fx[2]; int cs[2], cc[2];
        float4 ga[2], gs[2];
        float s1[2][4], s2[2][4];
#pragma unroll
        for (int cbk = 0; cbk < 2; ++cbk) {
            const int co = co_wave + cbk * 16 + 4 * (lane >> 4);
            if (co < p.OC0) { dst[cbk] = p.y; fx[cbk] = p.fx0; cs[cbk] = p.OC0; cc[cbk] = co; }
            else            { dst[cbk] = p.y1; fx[cbk] = p.fx1; cs[cbk] = Ctot - p.OC0; cc[cbk] = co - p.OC0; }
            ga[cbk] = make_float4(1.f, 1.f, 1.f, 1.f); gs[cbk] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bflags & TQ_BWD_GN) {
                ga[cbk] = *reinterpret_cast<const float4*>(p.fgs + (size_t)b * Ctot + co);
                gs[cbk] = *reinterpret_cast<const float4*>(p.fgh + (size_t)b * Ctot + co);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[cbk][j] = 0.f; s2[cbk][j] = 0.f; }
        }
        // Round 6: the loads of the epilogue -- the forward input x (chain through GN / SiLU, GN-backward sums) or, for accumulating
        // launches, the gradient already in place -- are issued for the whole wave tile FIRST (branch-free, clamped rows) and waited for
        // once.  Inside the store loop every load was followed by s_waitcnt vmcnt(0), i.e. by a wait for the previous stores as well:
        // sixteen dependent round trips per wave.  (A launch that needs both kinds keeps the in-loop loads for the second.)
        const bool need_x = p.bflags & (TQ_BWD_GN | TQ_BWD_SILU | TQ_BWD_STATS);
        const bool accum = p.bflags & TQ_BWD_ACCUM;
        f32x4 lq[TBW][2];   // (a vector type like the accumulators: as HIP float4 structs the batch stayed in scratch memory)
#ifdef TQ_ABL_EPI_SERIAL   // (A/B build: the loads inside the store loop, as in rounds 1-5)
        if (false) {
#else
        if (need_x || accum) {   // (wave-uniform)
#endif
#pragma unroll
            for (int tb = 0; tb < TBW; ++tb) {
                const int t = t0 + wn * C::WT + tb * 16 + (lane & 15);
                const int tc = t < p.T_out ? t : p.T_out - 1;
#pragma unroll
                for (int cbk = 0; cbk < 2; ++cbk) {
                    const size_t o = ((size_t)b * p.T_out + tc) * cs[cbk] + cc[cbk];
                    lq[tb][cbk] = *reinterpret_cast<const f32x4*>((need_x ? fx[cbk] : dst[cbk]) + o);
                }
            }
        }
#pragma unroll
        for (int tb = 0; tb < TBW; ++tb) {
            int t = t0 + wn * C::WT + tb * 16 + (lane & 15);
            asm volatile("" : "+v"(t));   // (store addresses recomputed, not kept live across the load batch: see the forward epilogue)
            if (t < p.T_out) {
#pragma unroll
                for (int cbk = 0; cbk < 2; ++cbk) {
                    const int co = co_wave + cbk * 16 + 4 * (lane >> 4);
                    const size_t o = ((size_t)b * p.T_out + t) * cs[cbk] + cc[cbk];
                    float v[4] = {acc[cbk][tb][0], acc[cbk][tb][1], acc[cbk][tb][2], acc[cbk][tb][3]};
                    if constexpr (SCH == 2) { v[0] *= dy_unscale; v[1] *= dy_unscale; v[2] *= dy_unscale; v[3] *= dy_unscale; }
                    float xv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (need_x) {
#ifdef TQ_ABL_EPI_SERIAL
                        const f32x4 x4 = *reinterpret_cast<const f32x4*>(fx[cbk] + o);
#else
                        const f32x4 x4 = lq[tb][cbk];
#endif
                        xv[0] = x4[0]; xv[1] = x4[1]; xv[2] = x4[2]; xv[3] = x4[3];
                    }
                    if (p.bflags & TQ_BWD_SILU) {
                        const float a4[4] = {ga[cbk].x, ga[cbk].y, ga[cbk].z, ga[cbk].w};
                        const float h4[4] = {gs[cbk].x, gs[cbk].y, gs[cbk].z, gs[cbk].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] *= dsilu_f(a4[j] * xv[j] + h4[j]);
                    }
                    if (p.bflags & TQ_BWD_DROPOUT) {
                        const uint32_t e0 = (uint32_t)t * (uint32_t)Ctot + (uint32_t)co;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            v[j] = (drop_hash(dkey, e0 + j) >= p.drop_thresh) ? v[j] * p.drop_scale : 0.f;
                    }
                    if (accum) {
#ifdef TQ_ABL_EPI_SERIAL
                        f32x4 r = *reinterpret_cast<const f32x4*>(dst[cbk] + o);
#else
                        f32x4 r = lq[tb][cbk];
                        if (need_x) r = *reinterpret_cast<const f32x4*>(dst[cbk] + o);
#endif
                        v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
                    }
                    *reinterpret_cast<float4*>(dst[cbk] + o) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s1[cbk][j] += v[j]; s2[cbk][j] += v[j] * xv[j]; }
                }
            }
        }
        if (p.bflags & TQ_BWD_STATS) {
#pragma unroll
            for (int cbk = 0; cbk < 2; ++cbk) {
                const int co = co_wave + cbk * 16 + 4 * (lane >> 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        s1[cbk][j] += __shfl_xor(s1[cbk][j], o);
                        s2[cbk][j] += __shfl_xor(s2[cbk][j], o);
                    }
                }
                if constexpr (TBW == 8) {
                    if ((lane & 15) == 0 && slot < p.nslots) {
                        float* st = p.stats + (((size_t)b * p.nslots + slot) * Ctot + co) * 2;
                        *reinterpret_cast<float4*>(st) = make_float4(s1[cbk][0], s2[cbk][0], s1[cbk][1], s2[cbk][1]);
                        *reinterpret_cast<float4*>(st + 4) = make_float4(s1[cbk][2], s2[cbk][2], s1[cbk][3], s2[cbk][3]);
                    }
                }
            }
            if constexpr (TBW == 4) {
                // 64-position waves: the two waves of a channel half cover the two halves of ONE 128-position slot; the second hands its
                // sums over through LDS (the staging buffers are idle once every wave has left its MFMA stream: first barrier)
                __syncthreads();
                float* red = reinterpret_cast<float*>(lds) + (wm * 4 + (lane >> 4)) * 16;   // [wm][kq][cbk][j][2]
                if (wn == 1 && (lane & 15) == 0) {
#pragma unroll
                    for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { red[(cbk * 4 + j) * 2] = s1[cbk][j]; red[(cbk * 4 + j) * 2 + 1] = s2[cbk][j]; }
                }
                __syncthreads();
                if (wn == 0 && (lane & 15) == 0 && slot < p.nslots) {
#pragma unroll
                    for (int cbk = 0; cbk < 2; ++cbk) {
                        const int co = co_wave + cbk * 16 + 4 * (lane >> 4);
                        float a1[4], a2[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { a1[j] = s1[cbk][j] + red[(cbk * 4 + j) * 2]; a2[j] = s2[cbk][j] + red[(cbk * 4 + j) * 2 + 1]; }
                        float* st = p.stats + (((size_t)b * p.nslots + slot) * Ctot + co) * 2;
                        *reinterpret_cast<float4*>(st) = make_float4(a1[0], a2[0], a1[1], a2[1]);
                        *reinterpret_cast<float4*>(st + 4) = make_float4(a1[2], a2[2], a1[3], a2[3]);
                    }
                }
            }
        }
    }
    if constexpr (PW) {
        co_wave += C::MT;
        if (++pass < npass) goto next_pass;
    }
#ifdef TQ_STAMP
    if (tid == 0 && blockIdx.x < 4096) {
        unsigned long long* tl = tq_timeline + blockIdx.x * 8;
        tl[3] = __builtin_amdgcn_s_memrealtime(); tl[7] = __builtin_amdgcn_s_memtime();
    }
#endif
}

template <int KT, int STRIDE, int UPS, int WM, int WN, int EPI, int ACT, bool FUSE, int SCH = 0, bool PW = false, int TBW = 8, int NCB = 2>
int launch(const ConvArgs& a, hipStream_t stream) {
    using C = Cfg<KT, STRIDE, UPS, WM, WN, SCH, TBW, NCB>;
    auto kern = conv1d_mfma_kernel<KT, STRIDE, UPS, WM, WN, EPI, ACT, FUSE, SCH, PW, TBW, NCB>;
    // scheme 2 keeps the folded GroupNorm coefficients of the workgroup's sample behind the staging buffers (2 x C_in floats)
    constexpr int GTAB_MAX = ((SCH == 2 && ACT >= 1) || (SCH == 0 && TBW == 2 && ACT >= 1 && !PW && EPI == 0)) ? 2 * 4 * 1024 : 0;   // room for C_in <= 1024
    constexpr int LDS_BYTES = (PW ? 4 * C::BUF : C::LDS_BYTES) + GTAB_MAX;
    if (GTAB_MAX && a.cf_st0 && a.C0 + a.C1 > 1024) return TQ_ERR_SHAPE;
    if (SCH == 2 && a.C0 + a.C1 > 1024) return TQ_ERR_SHAPE;
    if (a.cf_st0) {   // consumer-side GroupNorm fold: built into the small tile's forward launches; its scratch (2 C + 64 doubles) must fit the staging buffers
        constexpr bool built = (TBW == 2 && ACT >= 1 && !PW && EPI == 0) || (SCH == 2 && ACT >= 1 && !PW && EPI == 0 && TBW == 8);
        if (!built || (size_t)(2 * (a.C0 + a.C1) + 64) * sizeof(double) > (size_t)(LDS_BYTES - GTAB_MAX)) return TQ_ERR_SHAPE;   // (the table behind them is being written meanwhile)
    }
    // The dynamic-LDS limit is a per-device property of the kernel: remember, per device ordinal, that it has been raised
    // (idempotent call: two threads racing here both set the same value; the mask only saves the repeated runtime call).
    static std::atomic<uint64_t> attr_done{0};
    int dev_ord = 0;
    (void)hipGetDevice(&dev_ord);
    const uint64_t dev_bit = 1ull << (dev_ord & 63);
    if (!(attr_done.load(std::memory_order_acquire) & dev_bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done.fetch_or(dev_bit, std::memory_order_release);
    }
    const int n_ttiles = (a.T_out + C::NT - 1) / C::NT;
    const int n_ctiles = (a.C_out + C::MT - 1) / C::MT;
    const unsigned grid = (unsigned)(a.B * n_ttiles * (PW ? 1 : n_ctiles));
#ifdef TQ_BUILD_EXPERIMENTS
    if (a.gf_counters) {
        // fused GroupNorm finalisation: every workgroup of a sample takes a ticket -- all its waves must reach the epilogue (no ragged
        // channel tile) and the fold's scratch (2 C + 64 doubles) must fit the staging buffers it reuses
        const int Ctot = ((a.flags & TQ_CONV_POLY2) ? a.C_out / 2 : a.C_out) + a.gf_Cp;
        if (PW || EPI != 0 || (a.C_out % C::MT) || (size_t)(2 * Ctot + 64) * sizeof(double) > (size_t)LDS_BYTES) return TQ_ERR_SHAPE;
        ConvArgs a2 = a;
        a2.gf_narrive = n_ttiles * n_ctiles;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHR), LDS_BYTES, stream, a2);
        TQ_CHECK_LAUNCH();
        return 0;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHR), LDS_BYTES, stream, a);
    TQ_CHECK_LAUNCH();
    return 0;
}

template <int KT, int STRIDE, int UPS, int EPI, int ACT, bool FUSE = false>
int dispatch_tile(const ConvArgs& a, hipStream_t s) {
    if (a.t_tile == 32) {
        // Small tile for launch-bound batches (TqConvDesc.t_tile): 32 positions per workgroup, four times the workgroups of the 128-
        // position tiles at a quarter of the work each.  Built for the ResBlock convs (k = 5, GN + SiLU [+ dropout] prologue, with or
        // without the fused skip conv) in bf16x3 and in fp16 + MX-fp6 (128-channel tile).
        if constexpr (KT == 5 && STRIDE == 1 && UPS == 0 && EPI == 0 && ACT >= 2) {
            if (a.flags & TQ_CONV_POLY2) return TQ_ERR_SHAPE;
            if (a.wfmt == TQ_WFMT_F16_MX6) {
                if (a.C0 % 64 || a.C1 % 64 || a.sC0 % 64 || a.sC1 % 64 || a.C_out % 128) return TQ_ERR_SHAPE;
                return launch<KT, STRIDE, UPS, 4, 1, EPI, ACT, FUSE, 2, false, 2>(a, s);
            }
            if (a.wfmt != TQ_WFMT_BF16X3) return TQ_ERR_SHAPE;
            if (a.C_out % 128 == 0) return launch<KT, STRIDE, UPS, 4, 1, EPI, ACT, FUSE, 0, false, 2>(a, s);
            if (a.C_out % 64 == 0) return launch<KT, STRIDE, UPS, 2, 1, EPI, ACT, FUSE, 0, false, 2>(a, s);
            return launch<KT, STRIDE, UPS, 1, 1, EPI, ACT, FUSE, 0, false, 2>(a, s);
        }
        return TQ_ERR_SHAPE;
    }
    if (a.wfmt == TQ_WFMT_F16_MX6) {  // same shapes as TQ_WFMT_F16_MX8 (below), fp6 block-scaled corrections
        if constexpr (STRIDE == 1 && UPS == 0 && EPI == 1 && ACT == 0 && !FUSE) {   // data gradient (dy scaled by a power of two)
            if (a.C0 % 64 || a.C1) return TQ_ERR_SHAPE;
            // TQDNE_DGRAD_TILE256=0 (A/B switch, round 6): 256-channel outputs as two co-resident 4-wave workgroups -- does a neighbour's
            // MFMA stream cover the chain epilogue (x read back behind the last MFMA, +26 ... +73 % per launch)?
            static const int dg256 = [] { const char* e = getenv("TQDNE_DGRAD_TILE256"); return (e && e[0] == '0') ? 0 : 1; }();
            if (dg256 && a.C_out % 256 == 0) return launch<KT, STRIDE, UPS, 8, 1, EPI, ACT, FUSE, 2>(a, s);
            if (a.C_out % 128 == 0) return launch<KT, STRIDE, UPS, 4, 1, EPI, ACT, FUSE, 2>(a, s);
            if (a.C_out % 64 == 0) return launch<KT, STRIDE, UPS, 2, 2, EPI, ACT, FUSE, 2, false, 4>(a, s);   // (round 6: 64 / 192 input channels)
            return TQ_ERR_SHAPE;
        }
        if constexpr (STRIDE == 1 && EPI != 1) {
            if (a.C0 % 64 || a.C1 % 64 || a.sC0 % 64 || a.sC1 % 64) return TQ_ERR_SHAPE;
            if constexpr (KT == 1 && UPS == 0 && ACT <= 1) {
                const int cin = a.C0 + a.C1;
                // TQ_CONV_CH_TILES (round 6; TQDNE_QKV_PW=0 / 1 forces one form everywhere: A/B switch): launch-bound plans take the
                // channel-tiled form for the qkv projection too -- three times the workgroups of the input-stationary one
                static const int force = [] { const char* e = getenv("TQDNE_QKV_PW"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
                const bool pw = force >= 0 ? force == 1 : !(a.flags & TQ_CONV_CH_TILES);
                if (pw && a.C_out % 256 == 0 && a.C_out >= 512 && (cin == 128 || cin == 256))
                    return launch<KT, STRIDE, UPS, 8, 1, EPI, ACT, FUSE, 2, true>(a, s);
            }
            // (256-channel outputs as two co-resident 4-wave workgroups instead of one 8-wave one: measured 2-8 % slower per layer)
#ifdef TQ_EXP_NCB4
            if constexpr (KT > 1 && UPS == 0 && EPI == 0 && !FUSE) {
                // experiment (-DTQ_EXP_NCB4, TQDNE_CONV_NCB4=1): 4 x 2 waves of 64 channels x 64 positions instead of 8 x 1 of 32 x 128 --
                // half the LDS reads per MFMA, ONE weight buffer.  Bit-identical convolution, 10-14 % SLOWER (87 -> 99 us, 256 -> 256,
                // T = 1024): with one buffer a weight wait stands at every tap, and vmcnt retires in order, so the first of them also
                // waits for the chunk's staging loads issued at the start of the phase (tools/experiments/ncb4_ab.py)
                static const int ncb4 = [] { const char* e = getenv("TQDNE_CONV_NCB4"); return (e && e[0] == '1') ? 1 : 0; }();
                if (ncb4 && a.C_out % 256 == 0 && !(a.flags & TQ_CONV_POLY2))
                    return launch<KT, STRIDE, UPS, 4, 2, EPI, ACT, FUSE, 2, false, 4, 4>(a, s);
            }
#endif
            // TQDNE_CONV_TILE256=0 (A/B switch, round 5): 256-channel outputs as two co-resident 4-wave workgroups (128 channels each) instead
            // of one 8-wave one.  Per layer 2-8 % slower (round 2: co-resident workgroups of ONE launch run in lock-step, and every input
            // tile is staged twice); re-measured under the 4-lane sampler, where co-resident workgroups of different lanes are out of phase
            static const int tile256 = [] { const char* e = getenv("TQDNE_CONV_TILE256"); return (e && e[0] == '0') ? 0 : 1; }();
            if (tile256 && a.C_out % 256 == 0) return launch<KT, STRIDE, UPS, 8, 1, EPI, ACT, FUSE, 2>(a, s);
#ifdef TQ_EXP_WN2
            if constexpr (KT == 5 && UPS == 0 && EPI == 0 && ACT >= 2) {
                // experiment: 128 channels x 256 positions in ONE 8-wave workgroup instead of two co-resident 4-wave ones -- the weights
                // (C_in x 128 x 5 x 2.2 B per tile: 540 KB for 384 -> 128) are streamed from L2 once per 256 positions instead of 128
                static const int wn2 = [] { const char* e = getenv("TQDNE_CONV_WN2"); return (e && e[0] == '1') ? 1 : 0; }();
                if (wn2 && a.C_out % 128 == 0 && !(a.flags & TQ_CONV_POLY2) && a.T_out % 256 == 0)
                    return launch<KT, STRIDE, UPS, 4, 2, EPI, ACT, FUSE, 2>(a, s);
            }
#endif
            if (a.C_out % 128 == 0) return launch<KT, STRIDE, UPS, 4, 1, EPI, ACT, FUSE, 2>(a, s);
            if constexpr (KT == 5 && UPS == 0 && EPI == 0 && ACT >= 2) {   // round 6: the 64-channel ResBlock convs (with or without the fused skip conv)
                if (a.C_out % 64 == 0 && !(a.flags & TQ_CONV_POLY2)) return launch<KT, STRIDE, UPS, 2, 2, EPI, ACT, FUSE, 2, false, 4>(a, s);
            }
        }
        return TQ_ERR_SHAPE;
    }
    if (a.wfmt == TQ_WFMT_F16_MX8) {  // built for stride-1 forward launches with 128 | C_out and 64-channel sources; with the fused
        // skip conv only for the 256-channel tile (the 128-channel one has 4 of its 8 waves' worth of registers to hide latency
        // with and spills > 100 of them)
        if constexpr (STRIDE == 1 && EPI != 1) {

            if (a.C0 % 64 || a.C1 % 64 || a.sC0 % 64 || a.sC1 % 64) return TQ_ERR_SHAPE;
            if constexpr (KT == 1 && UPS == 0 && ACT <= 1) {  // the attention block's 1x1 convs: input-stationary variant
                const int cin = a.C0 + a.C1;
                // (one channel tile, i.e. proj_out: nothing to share, measured equal -> the generic path)
                if (a.C_out % 256 == 0 && a.C_out >= 512 && (cin == 128 || cin == 256))
                    return launch<KT, STRIDE, UPS, 8, 1, EPI, ACT, FUSE, 1, true>(a, s);
            }
            if (a.C_out % 256 == 0) return launch<KT, STRIDE, UPS, 8, 1, EPI, ACT, FUSE, 1>(a, s);
            if (a.C_out % 128 == 0) return launch<KT, STRIDE, UPS, 4, 1, EPI, ACT, FUSE, 1>(a, s);
        }
        return TQ_ERR_SHAPE;
    }
    if (a.wfmt != TQ_WFMT_BF16X3) return TQ_ERR_ARG;
    // 256 output channels: one 8-wave workgroup stages each input tile once instead of two 4-wave workgroups staging it
    // twice (measured -32 % for pointwise convs, which are staging-bound, and -2...-5 % for k = 5)
    if constexpr (STRIDE == 1 && UPS == 0) {
        if (a.C_out % 256 == 0) return launch<KT, STRIDE, UPS, 8, 1, EPI, ACT, FUSE>(a, s);
    }
    if (a.C_out % 128 == 0) return launch<KT, STRIDE, UPS, 4, 1, EPI, ACT, FUSE>(a, s);
    if constexpr (STRIDE == 1) {
#ifdef TQ_BUILD_EXPERIMENTS
        if constexpr (UPS == 0 && EPI == 0 && KT == 5 && ACT != 3) {   // (the dropout prologue spills in this tile: training keeps 64 x 256)
            // 64-channel outputs at T = 4096 are bound by their load / store bursts, not by MFMA cycles (section 5 of DESIGN.md).
            // The slim tile (64 channels x 128 positions, 32 accumulator registers per wave, 152-168 registers) lets three
            // workgroups share a CU instead of two.  Measured (tools/slim_ab.py, B = 64, same box, bit-identical outputs): SLOWER --
            // 64 -> 64: 52-55 vs 50 us, 64+64 -> 64: 90 vs 82, 128+64 -> 64: 118-123 vs 115: 2048 tiles on 768 slots are 2.7 rounds
            // where 1024 tiles on 512 slots are exactly 2, and every tile re-streams the weights and 4 halo rows for half the
            // positions.  Off by default; TQDNE_CONV_SLIM=1 selects it.
            static const int slim = [] { const char* e = getenv("TQDNE_CONV_SLIM"); return (e && e[0] == '1') ? 1 : 0; }();
            if (slim && a.C_out % 64 == 0 && !(a.flags & TQ_CONV_POLY2))
                return launch<KT, STRIDE, UPS, 2, 2, EPI, ACT, FUSE, 0, false, 4>(a, s);
        }
#endif
        if (a.C_out % 64 == 0) return launch<KT, STRIDE, UPS, 2, 2, EPI, ACT, FUSE>(a, s);
        return launch<KT, STRIDE, UPS, 1, 2, EPI, ACT, FUSE>(a, s);
    } else {
        if (a.C_out % 64 == 0) return launch<KT, STRIDE, UPS, 2, 1, EPI, ACT, FUSE>(a, s);
        return launch<KT, STRIDE, UPS, 1, 1, EPI, ACT, FUSE>(a, s);
    }
}

// forward prologue variants; strided / upsampled convs and data gradients only exist un-activated in the networks served
template <int KT>
int dispatch_act(const ConvArgs& a, hipStream_t s) {
    const bool gn = a.flags & TQ_CONV_GN, silu = a.flags & TQ_CONV_SILU, drop = a.flags & TQ_CONV_DROPOUT;
    if ((!gn && silu) || (drop && !silu)) return TQ_ERR_ARG;  // supported prologues: none | GN | GN+SiLU | GN+SiLU+dropout
    if (a.sx0) {  // fused 1x1 skip conv: built for the ResBlock's second conv (k = 5, GN + SiLU [+ dropout])
        if constexpr (KT == 5) {
            if (gn && silu && drop) return dispatch_tile<KT, 1, 0, 0, 3, true>(a, s);
            if (gn && silu) return dispatch_tile<KT, 1, 0, 0, 2, true>(a, s);
        }
        return TQ_ERR_SHAPE;
    }
    if (gn && silu && drop) return dispatch_tile<KT, 1, 0, 0, 3>(a, s);
    if (gn && silu) return dispatch_tile<KT, 1, 0, 0, 2>(a, s);
    if (gn) return dispatch_tile<KT, 1, 0, 0, 1>(a, s);
    return dispatch_tile<KT, 1, 0, 0, 0>(a, s);
}

}  // namespace
