// Backward kernels of the tqdne 1-D UNet training step on gfx950 (the autograd work Lightning runs for
// reference edm.py:136 training_step):
//   * weight gradient of the fused conv (MFMA, bf16x3, transposed LDS reads) + slab reduction
//   * GroupNorm32 backward (finalise group sums -> per-(b,c) coefficients; elementwise apply)
//   * column sums (bias / time-embedding gradients), zero-stuffing and pair-sum (strided / upsampled convs)
//   * stem weight gradient and head (last conv) backward
#include <cstdlib>
#include "common.hpp"
#include "../../include/tqdne_hip.h"

using namespace tq;

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

// -DTQ_WG_ABL_HIHI (timing ablation, wrong numerics, never in the shipped library): the weight gradient's products with the two
// first-order terms dropped (ONE bf16 MFMA per product instead of three) = the bound of any cheaper contraction scheme for it;
// -DTQ_WG_ABL_HALF: two of the three (the MFMA cycles an fp16 + MX-fp6 port would execute: 1.5 units, here as 2)
__device__ __forceinline__ f32x4 wg_mma(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x4 c) {
#if defined(TQ_WG_ABL_HIHI)
    return mfma_bf16(ah, bh, c);
#elif defined(TQ_WG_ABL_HALF)
    c = mfma_bf16(ah, bl, c);
    return mfma_bf16(ah, bh, c);
#else
    return mfma_x3(ah, al, bh, bl, c);
#endif
}

// ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of (row q, cols 4p..4p+3); lane i receives
// column i of rows 0..3.  Two reads (rows +0, +4) give the 8 consecutive k of one MFMA operand fragment.
__device__ __forceinline__ uint2 lds_tr_read(const unsigned char* p) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    union { s16x4 s; uint2 u; } c;
    c.s = v;
    return c.u;
}

// =================================================================================================
// Weight gradient:  dW[co, ci, k] = sum_{b,t} dy[b, t, co] * xhat[b, t*stride + k - pad, ci]
//   xhat = dropout(SiLU(gscale*x + gshift)) is re-computed while staging, exactly as in the forward kernel.
// GEMM view: M = co (128 per workgroup, 32 per wave), N = (tap, ci chunk of 32), reduction over (b, t).
// Both operands need the reduction index contiguous per lane, i.e. a transposed view of the channels-last
// tiles: the LDS images stay channels-last ([t][c], written with 8-byte stores) and are read with
// ds_read_b64_tr_b16.  Taps are row offsets of the xhat image.  Partial results of the (b,t) splits go to
// slabs [split][k][co][ci] with plain 64-byte-segment stores and are summed by wgrad_reduce_kernel
// (scattered fp32 atomics would run an order of magnitude slower).
// =================================================================================================
struct WgArgs {
    const float* dy;
    const float* x0;
    const float* x1;
    const float* gscale;
    const float* gshift;
    float* slab;
    int B, T_in, T_out, C0, C1, C_out;
    int flags, nsplit, units_per_split, n_ttiles, n_cotiles, n_cichunks;
    uint32_t drop_site, drop_thresh;
    float drop_scale;
    uint64_t drop_seed;
    // fused column sums of dy (bias / time-embedding gradients): accumulated by the workgroups of the first input-channel chunk
    float* cs_bc;   // [b * cs_stride + co] += sum_t dy[b, t, co]   (nullable)
    int cs_stride;
    float* cs_c;    // [co] += sum_{b,t} dy                          (nullable)
    float* cs_c2;   // second destination of the same sums           (nullable)
};

constexpr int WG_TT = 64;        // reduction positions per staged tile
constexpr int TQ_WGRAD_PLAIN_ORDER = 1 << 30;  // (internal flag bit of WgArgs.flags: A/B switch TQDNE_WGRAD_XCD=0)
// LDS images of the weight gradient, conflict-free for ds_read_b64_tr_b16 (32-lane groups: rows {r .. r+3} and {r+8 .. r+11},
// 8 bytes per lane): row stride = 8 banks (mod 64) for dy, 16 banks for the 64-byte xhat rows, and the 8-byte column slot is
// XOR-ed with bit 3 of the row (dy: slot ^ 16, xhat: slot ^ 4), so that the two 4-row halves of a group -- 8 rows apart, i.e. a
// multiple of 64 banks -- land in different halves of their rows' bank windows.  (272-byte / plain rows: 40 % of the LDS-active
// cycles were conflicts, rocprofv3 SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.)
constexpr int WG_DY_STRIDE1 = 288;  // bytes per dy image row, 32-channel chunks (128 co * 2 B + 32 pad: 72 banks = 8 mod 64)
constexpr int WG_DY_STRIDE2 = 272;  // 64-channel chunks (k = 1 only): the plain image -- 288-byte rows would cost the third resident workgroup
template <int NCI> __device__ __forceinline__ int wg_dy_swz(int row) { return (NCI == 1 || NCI == 4) ? ((row >> 3) & 1) << 7 : 0; }   // slot ^ 16
__device__ __forceinline__ int wg_x_swz(int row) { return ((row >> 3) & 1) << 5; }    // byte offset XOR: slot ^ 4 (64-byte rows)
// 128-byte xhat rows (W8): a 32-lane read touches rows {r .. r+3} and {r+8 .. r+11}, 4 slots of 8 bytes each; rows of equal parity
// share their 32 banks, so the slot is XOR-ed with 4 * bit 1 and 8 * bit 3 of the row: r, r+2, r+8, r+10 land in four disjoint
// slot groups
__device__ __forceinline__ int wg_x_swz128(int row) { return (((row >> 1) & 1) << 5) | (((row >> 3) & 1) << 6); }
// 256-byte xhat rows (NCI = 4, round 5): every row spans all 64 banks, so the eight rows of a 32-lane read (r .. r+3, r+8 .. r+11; 32
// bytes each) go to eight different 32-byte slots: slot ^= (row & 3) | 4 * bit 3 of the row
__device__ __forceinline__ int wg_x_swz256(int row) { return ((row & 3) << 5) | (((row >> 3) & 1) << 7); }

// NCI = 4 (round 5, k = 1 only): a 128-channel input chunk per workgroup -- the 1x1 convs (ResBlock skip convs, attention qkv / proj) have
// a fifth of a k = 5 conv's MFMA work per staged dy tile, so their launches are bound by re-reading and re-splitting dy once per input
// chunk (C_in / 64 times with NCI = 2): 64 accumulator registers, both LDS images conflict-free (288-byte dy rows, swizzled 256-byte
// xhat rows), 70 KB of LDS = two workgroups per CU as before.
// NCI: input-channel chunk of a workgroup in units of 32.  The dy tile (128 co) is staged, split and read once per workgroup and
// unit whatever the chunk width, so the 64-channel chunk (NCI = 2) halves the number of times dy is re-read from L2 / HBM and
// re-split (measured traffic-bound with 32: every conv's dy was read C_in / 32 times); 160 accumulator registers at k = 5.
// W8 (round 4): EIGHT waves per workgroup, one workgroup per CU: waves 0-3 and 4-7 share the staged dy tile and take the two 32-channel
// halves of a 64-channel xhat tile (accumulators as NCI = 1: 80 registers at k = 5, where the 64-channel chunk in ONE wave -- NCI = 2,
// 160 registers -- spilled).  Every dy element is then fetched from L2, split and stored to LDS C_in / 64 instead of C_in / 32 times:
// the kernel was moving ~6 TB/s from L2 at 38 % MFMA-busy (667 MB per 256 -> 256, k = 5 launch).
// H64 (round 6): the W8 layout for convs with 64 output channels, in FOUR waves: 2 blocks of 32 output channels x the two 32-channel
// halves of the 64-channel xhat tile, a 64-row x 64-channel dy tile, two workgroups per CU.  In the 128-channel tile of the other forms
// such a conv leaves every second wave without matrix work (wave_active) while the whole workgroup still stages, splits and stores a
// dy tile that is half zeros, and it re-stages dy once per 32 input channels.
template <int KT, int STRIDE, int UPS, int NCI, bool W8 = false, bool H64 = false>
__global__ __launch_bounds__((W8 && !H64) ? 512 : 256, (W8 && !H64) ? 1 : 2) void wgrad_kernel(const WgArgs p) {
    static_assert(!W8 || NCI == 1, "W8: per-wave accumulators of a 32-channel chunk");
    static_assert(!H64 || W8, "H64: the W8 layout (two 32-channel halves of one xhat tile) with two output-channel blocks");
    constexpr int NT = (W8 && !H64) ? 512 : 256;
    constexpr int COT = H64 ? 64 : 128;      // output channels of the workgroup's dy tile
    constexpr int DYC4 = COT / 4;            // float4 columns of the dy tile
    constexpr int NCX = W8 ? 2 : NCI;        // width of the staged xhat tile in 32-channel units
    constexpr int PAD = (STRIDE == 1) ? KT / 2 : 1;
    constexpr int XR = (STRIDE == 1) ? (WG_TT + KT - 1) : (2 * WG_TT + 1);
    constexpr int WG_X_STRIDE = 64 * NCX;    // bytes per xhat image row (32 NCX ci * 2 B)
    constexpr int XC4 = 8 * NCX;             // float4 columns of the xhat tile
    constexpr int XIT = (XR * XC4 + NT - 1) / NT;
    constexpr int DYIT = WG_TT * DYC4 / NT;  // float4 of the 64 x COT dy tile per thread
    static_assert(NCI != 4 || (KT == 1 && STRIDE == 1 && !UPS), "128-channel chunks: the 1x1 convs");
    constexpr int WG_DY_STRIDE = (NCI == 1 || NCI == 4) ? WG_DY_STRIDE1 : WG_DY_STRIDE2;
    constexpr int DY_PLANE = WG_TT * WG_DY_STRIDE;
    constexpr int X_PLANE = XR * WG_X_STRIDE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* dy_hi = lds;
    unsigned char* dy_lo = dy_hi + DY_PLANE;
    unsigned char* x_hi = dy_lo + DY_PLANE;
    unsigned char* x_lo = x_hi + X_PLANE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = H64 ? (wave_id & 1) : (W8 ? (wave_id & 3) : wave_id);   // 32-channel block of the COT output channels
    const int wci = H64 ? (wave_id >> 1) : (W8 ? (wave_id >> 2) : 0);        // W8 / H64: which 32-channel half of the xhat tile
    // Workgroups are dealt round-robin over the 8 XCDs (ids i and i + 8 share an L2).  The input-channel-chunk workgroups of one
    // (split, co-tile) pair all stage the same dy tile: keep a pair on ONE XCD, so its L2 serves the re-reads (with the plain
    // order every XCD fetched every dy tile: 404 MB from HBM for 134 MB of operands, 57 % L2 misses).
    int ct, cc, sp;
    {
        const int bid = blockIdx.x;
        const int npairs = p.nsplit * p.n_cotiles;
        if ((npairs & 7) == 0 && !(p.flags & TQ_WGRAD_PLAIN_ORDER)) {
            const int xcd = bid & 7, j = bid >> 3;
            const int pi = (j / p.n_cichunks) * 8 + xcd;
            cc = j % p.n_cichunks;
            sp = pi / p.n_cotiles;
            ct = pi % p.n_cotiles;
        } else {
            ct = bid % p.n_cotiles;
            cc = (bid / p.n_cotiles) % p.n_cichunks;
            sp = bid / (p.n_cotiles * p.n_cichunks);
        }
    }
    const int co0 = ct * COT;
    const int cb = cc * 32 * NCX;
    const int Cin = p.C0 + p.C1;
    const int T_src = UPS ? 2 * p.T_in : p.T_in;
    const int U = p.B * p.n_ttiles;
    const int u_begin = sp * p.units_per_split;
    const int u_end = min(U, u_begin + p.units_per_split);
    const bool wave_active = (co0 + wave * 32) < p.C_out;
    const int cvalid4 = min(COT, p.C_out - co0) >> 2;  // float4 columns of dy actually present

    // xhat source (one concat source per 32-channel chunk)
    const float* xsrc; int xcs, xoff;
    if (cb < p.C0) { xsrc = p.x0; xcs = p.C0; xoff = cb; } else { xsrc = p.x1; xcs = p.C1; xoff = cb - p.C0; }
    const int m = tid & (XC4 - 1);  // float4 column of the xhat tile owned by this thread

    float4 dyr[DYIT];
    float4 xr[XIT];
    float4 g_a = make_float4(1.f, 1.f, 1.f, 1.f), g_s = make_float4(0.f, 0.f, 0.f, 0.f);

    auto xpos = [&](int t0, int i) -> int __attribute__((always_inline)) {
        if (STRIDE == 1) return t0 - PAD + i;
        const int par = (i > WG_TT) ? 1 : 0;
        const int idx = i - par * (WG_TT + 1);
        return 2 * t0 - PAD + 2 * idx + par;
    };

    auto load_unit = [&](int u) __attribute__((always_inline)) {
        const int b = u / p.n_ttiles;
        const int t0 = (u % p.n_ttiles) * WG_TT;
        const float* dyb = p.dy + ((size_t)b * p.T_out) * p.C_out + co0;
#ifdef TQ_ABL_NODY
        if (u == u_begin)   // ablation (wrong numerics): the dy tile is loaded, split and stored for the first unit only
#endif
#pragma unroll
        for (int it = 0; it < DYIT; ++it) {
            const int task = tid + it * NT;
            const int row = task / DYC4, c4 = task & (DYC4 - 1);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t0 + row < p.T_out && c4 < cvalid4) v = *reinterpret_cast<const float4*>(dyb + (size_t)(t0 + row) * p.C_out + 4 * c4);
            dyr[it] = v;
        }
        const float* xb = xsrc + (size_t)b * p.T_in * xcs + xoff + 4 * m;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int i = (tid + it * NT) / XC4;
            const int pos = xpos(t0, i);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < XR && pos >= 0 && pos < T_src) v = *reinterpret_cast<const float4*>(xb + (size_t)(UPS ? (pos >> 1) : pos) * xcs);
            xr[it] = v;
        }
        if (p.flags & TQ_CONV_GN) {
            g_a = *reinterpret_cast<const float4*>(p.gscale + (size_t)b * Cin + cb + 4 * m);
            g_s = *reinterpret_cast<const float4*>(p.gshift + (size_t)b * Cin + cb + 4 * m);
        }
    };

    // column sums of dy, fused (the stand-alone pass re-read every dy once more): accumulated in LDS (registers are what this
    // kernel does not have: a float4 of running sums per thread tipped the k = 5 instantiation into 29 spills), flushed with
    // global atomics when the sample changes and at the end.  All conditions are workgroup-uniform.
    const bool do_cs = (cc == 0) && (p.cs_bc || p.cs_c);
    float* cs_lds = reinterpret_cast<float*>(x_lo + X_PLANE);   // [128]
    int cs_b = -1;
    if (do_cs && tid < 128) cs_lds[tid] = 0.f;   // (W8 launches never carry fused column sums: the host refuses the combination)
    auto cs_flush = [&]() __attribute__((always_inline)) {
        if (tid < 128) {
            const int c = co0 + tid;
            const float v = cs_lds[tid];
            if (cs_b >= 0 && c < p.C_out) {
                if (p.cs_bc) atomicAdd(p.cs_bc + (size_t)cs_b * p.cs_stride + c, v);
                if (p.cs_c) atomicAdd(p.cs_c + c, v);
                if (p.cs_c2) atomicAdd(p.cs_c2 + c, v);
            }
            cs_lds[tid] = 0.f;
        }
    };

    auto write_unit = [&](int u) __attribute__((always_inline)) {
        const int b = u / p.n_ttiles;
        const int t0 = (u % p.n_ttiles) * WG_TT;
        if (do_cs) {
            if (b != cs_b) {   // (the previous unit's LDS adds are complete: a barrier separates the units)
                cs_flush();
                cs_b = b;
                __syncthreads();
            }
            float4 a4 = dyr[0];
#pragma unroll
            for (int it = 1; it < DYIT; ++it) { a4.x += dyr[it].x; a4.y += dyr[it].y; a4.z += dyr[it].z; a4.w += dyr[it].w; }
            float* dst = cs_lds + 4 * (tid & (DYC4 - 1));
            atomicAdd(dst, a4.x); atomicAdd(dst + 1, a4.y); atomicAdd(dst + 2, a4.z); atomicAdd(dst + 3, a4.w);
        }
#ifdef TQ_ABL_NODY
        if (u == u_begin)
#endif
#pragma unroll
        for (int it = 0; it < DYIT; ++it) {
            const int task = tid + it * NT;
            const int row = task / DYC4, c4 = task & (DYC4 - 1);
            const float v[4] = {dyr[it].x, dyr[it].y, dyr[it].z, dyr[it].w};
            bf16x4 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) { __bf16 hh, ll; split_bf16(v[j], hh, ll); h[j] = hh; l[j] = ll; }
            const int off = row * WG_DY_STRIDE + ((c4 * 8) ^ wg_dy_swz<NCI>(row));
            *reinterpret_cast<bf16x4*>(dy_hi + off) = h;
            *reinterpret_cast<bf16x4*>(dy_lo + off) = l;
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int i = (tid + it * NT) / XC4;
            if (i >= XR) continue;
            const int pos = xpos(t0, i);
            float v[4] = {xr[it].x, xr[it].y, xr[it].z, xr[it].w};
            if (pos >= 0 && pos < T_src) {
                if (p.flags & TQ_CONV_GN) {
                    v[0] = g_a.x * v[0] + g_s.x; v[1] = g_a.y * v[1] + g_s.y;
                    v[2] = g_a.z * v[2] + g_s.z; v[3] = g_a.w * v[3] + g_s.w;
                }
                if (p.flags & TQ_CONV_SILU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
                }
                if (p.flags & TQ_CONV_DROPOUT) {
                    const uint32_t e0 = (uint32_t)pos * (uint32_t)Cin + (uint32_t)(cb + 4 * m);
                    const DropKey dkey = drop_key(p.drop_seed, p.drop_site, (uint32_t)b);   // (b is uniform: scalar unit)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[j] = (drop_hash(dkey, e0 + j) >= p.drop_thresh) ? v[j] * p.drop_scale : 0.f;
                }
            }
            bf16x4 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) { __bf16 hh, ll; split_bf16(v[j], hh, ll); h[j] = hh; l[j] = ll; }
            const int off = i * WG_X_STRIDE + (W8 ? ((m * 8) ^ wg_x_swz128(i)) : (NCI == 1 ? ((m * 8) ^ wg_x_swz(i)) : (NCI == 4 ? ((m * 8) ^ wg_x_swz256(i)) : m * 8)));
            *reinterpret_cast<bf16x4*>(x_hi + off) = h;
            *reinterpret_cast<bf16x4*>(x_lo + off) = l;
        }
    };

    f32x4 acc[2][2 * NCI][KT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2 * NCI; ++j)
#pragma unroll
            for (int k = 0; k < KT; ++k) acc[i][j][k] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;

    // xhat image offset of (row, 16-channel block nb, this lane's 8-byte slot)
    auto x_off = [&](int xrow, int nb) -> int __attribute__((always_inline)) {
        const int colx = ((wci * 2 + nb) * 16 + 4 * pp) * 2;
        return xrow * WG_X_STRIDE + (W8 ? (colx ^ wg_x_swz128(xrow)) : (NCI == 1 ? (colx ^ wg_x_swz(xrow)) : (NCI == 4 ? (colx ^ wg_x_swz256(xrow)) : colx)));
    };
    auto a_frags = [&](int r0, Frag (&ah)[2], Frag (&al)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int col = (wave * 32 + mb * 16 + 4 * pp) * 2;
            const int off = r0 * WG_DY_STRIDE + (col ^ wg_dy_swz<NCI>(r0));
            const int off4 = off + 4 * WG_DY_STRIDE;   // (rows r0 and r0 + 4 share bit 3: r0 = 32 ks + 8 g + q, q < 4)
            ah[mb].h[0] = lds_tr_read(dy_hi + off);
            ah[mb].h[1] = lds_tr_read(dy_hi + off4);
            al[mb].h[0] = lds_tr_read(dy_lo + off);
            al[mb].h[1] = lds_tr_read(dy_lo + off4);
        }
    };
#if !defined(TQ_WG_OLD_COMPUTE)
    // Round 4.  The compute phase ran at 2.2x its MFMA time with staging and loads ablated (tools/bwd_micro.py, -DTQ_WG_ABL_*): per
    // (tap, channel block) step hipcc issued the four transposed reads of the xhat fragment right in front of the six MFMAs that
    // consume them -- an LDS round trip exposed twenty times per unit.  For the stride-1 convs the taps are row shifts of ONE image, so
    // a lane's fragments for all KT taps are 8-element windows of the 8 + KT - 1 consecutive rows it can fetch with THREE transposed
    // reads per plane (rows 8g .. 8g + 11): even shifts are register renames, odd ones five v_alignbit per plane.  LDS reads per wave
    // and unit 96 -> 40, issued one (32-row step, channel block) group ahead of the 6 KT MFMAs that use them.
    struct XWin { uint2 h[3], l[3]; };
    auto x_window = [&](int r0, int nb, XWin& w) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int off = x_off(r0 + 4 * j, nb);
            w.h[j] = lds_tr_read(x_hi + off);
            w.l[j] = lds_tr_read(x_lo + off);
        }
    };
    auto tap_frag = [&](const uint2 (&r)[3], const uint32_t (&al5)[5], int k, Frag& f) __attribute__((always_inline)) {
        const uint32_t d[6] = {r[0].x, r[0].y, r[1].x, r[1].y, r[2].x, r[2].y};
        if ((k & 1) == 0) f.u = make_uint4(d[k / 2], d[k / 2 + 1], d[k / 2 + 2], d[k / 2 + 3]);
        else f.u = make_uint4(al5[k / 2], al5[k / 2 + 1], al5[k / 2 + 2], al5[k / 2 + 3]);
    };
    auto compute = [&]() __attribute__((always_inline)) {
        if constexpr (STRIDE == 1 && KT > 1) {
            constexpr int NB = 2 * NCI, NG = (WG_TT / 32) * NB;   // groups = (32-row step, channel block)
            XWin win[2];
            Frag ah[2], al[2];
            x_window(8 * g + q, 0, win[0]);
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const int ks = gi / NB, nb = gi % NB;
                const int r0 = ks * 32 + 8 * g + q;
                if (nb == 0) a_frags(r0, ah, al);
                if (gi + 1 < NG) x_window(((gi + 1) / NB) * 32 + 8 * g + q, (gi + 1) % NB, win[(gi + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const XWin& w = win[gi & 1];
                const uint32_t dh[6] = {w.h[0].x, w.h[0].y, w.h[1].x, w.h[1].y, w.h[2].x, w.h[2].y};
                const uint32_t dl[6] = {w.l[0].x, w.l[0].y, w.l[1].x, w.l[1].y, w.l[2].x, w.l[2].y};
                uint32_t oh[5], ol[5];   // odd shifts: elements (2j + 1, 2j + 2)
#pragma unroll
                for (int j = 0; j < (KT == 5 ? 5 : 4); ++j) {
                    oh[j] = __builtin_amdgcn_alignbit(dh[j + 1], dh[j], 16);
                    ol[j] = __builtin_amdgcn_alignbit(dl[j + 1], dl[j], 16);
                }
                if (KT != 5) { oh[4] = 0; ol[4] = 0; }
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    Frag bh, bl;
                    tap_frag(w.h, oh, k, bh);
                    tap_frag(w.l, ol, k, bl);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
                        acc[mb][nb][k] = wg_mma(ah[mb].v, al[mb].v, bh.v, bl.v, acc[mb][nb][k]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
#else
    auto compute = [&]() __attribute__((always_inline)) {
#endif
#pragma unroll (NCI == 2 && KT == 5 ? 1 : 2)
        for (int ks = 0; ks < WG_TT / 32; ++ks) {
            const int r0 = ks * 32 + 8 * g + q;  // reduction row supplied by this lane (first read; +4 second)
            Frag ah[2], al[2];
            a_frags(r0, ah, al);
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const int xrow = (STRIDE == 1) ? (r0 + k) : ((k & 1) * (WG_TT + 1) + r0 + (k >> 1));
#pragma unroll
                for (int nb = 0; nb < 2 * NCI; ++nb) {
                    const int off = x_off(xrow, nb);
                    const int off4 = x_off(xrow + 4, nb);
                    Frag bh, bl;
                    bh.h[0] = lds_tr_read(x_hi + off);
                    bh.h[1] = lds_tr_read(x_hi + off4);
                    bl.h[0] = lds_tr_read(x_lo + off);
                    bl.h[1] = lds_tr_read(x_lo + off4);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
                        acc[mb][nb][k] = wg_mma(ah[mb].v, al[mb].v, bh.v, bl.v, acc[mb][nb][k]);
                }
            }
        }
    };

    // The next unit's loads are normally in flight under this unit's MFMAs.  With 160 accumulator registers (k = 5, 64-channel
    // chunk) there is no room for the 52 staging registers across the MFMA phase: that variant loads right before it converts and
    // leaves the latency to the co-resident workgroup.
    constexpr bool PREFETCH = !(NCI == 2 && KT == 5);
    if (PREFETCH && u_begin < u_end) load_unit(u_begin);
    // (-DTQ_WG_ABL_*: timing ablations, wrong numerics, never in the shipped library: NOMMA = no MFMA phase, NOSTAGE = units after the
    // first neither converted nor stored, NOLOAD = units after the first not loaded)
    for (int u = u_begin; u < u_end; ++u) {
#ifdef TQ_WG_ABL_NOLOAD
        if (!PREFETCH && u == u_begin) load_unit(u);
#else
        if (!PREFETCH) load_unit(u);
#endif
        __syncthreads();
#ifdef TQ_WG_ABL_NOSTAGE
        if (u == u_begin)
#endif
        write_unit(u);
        __syncthreads();
#ifndef TQ_WG_ABL_NOLOAD
        if (PREFETCH && u + 1 < u_end) load_unit(u + 1);
#endif
#ifndef TQ_WG_ABL_NOMMA
        if (wave_active) compute();
#endif
    }

    if (do_cs) {
        __syncthreads();
        cs_flush();
    }
    // ---- partial result -> slab[sp][k][co][ci]
    if (!wave_active) return;
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2 * NCI; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + wave * 32 + mb * 16 + 4 * (lane >> 4) + r;
                    const int ci = cb + (wci * 2 + nb) * 16 + (lane & 15);
                    p.slab[(((size_t)sp * KT + k) * p.C_out + co) * Cin + ci] = acc[mb][nb][k][r];
                }
}

// dw[i*KT + k] = sum_s slab[s][k][i].  One workgroup (16 x 16 threads) owns 64 consecutive i for ALL taps: thread (x, y) sums the
// splits s = y, y + 16, ... of the four elements 4x .. 4x + 3 of every tap with 16-byte loads -- for the usual <= 32 splits that is ONE
// round of <= 2 KT independent loads per thread, n / 64 workgroups (>= 1024 for the 256-channel layers) -- the sixteen partitions
// meet in LDS and the 64 KT results leave as one contiguous run of dw.  (Rounds 1-3: one 4-byte element per thread and tap, two
// loads in flight, 4-byte stores KT floats apart.  A first round-4 form with 256 i per workgroup had n / 256 workgroups of four
// latency-bound waves and measured slower than that.)
template <int KT>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nsplit, size_t n) {
    __shared__ __attribute__((aligned(16))) float red[16][KT][68];   // (68: 16-byte rows whose taps sit 4 banks apart)
    const int x = threadIdx.x, y = threadIdx.y;
    const size_t i0 = (size_t)blockIdx.x * 64;
    const size_t i = i0 + 4 * x;
    float4 acc[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        const float* base = slab + i;
        int sidx = y;
        for (; sidx + 16 < nsplit; sidx += 32) {
            float4 v0[KT], v1[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                v0[k] = *reinterpret_cast<const float4*>(base + ((size_t)sidx * KT + k) * n);
                v1[k] = *reinterpret_cast<const float4*>(base + ((size_t)(sidx + 16) * KT + k) * n);
            }
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                acc[k].x += v0[k].x + v1[k].x; acc[k].y += v0[k].y + v1[k].y;
                acc[k].z += v0[k].z + v1[k].z; acc[k].w += v0[k].w + v1[k].w;
            }
        }
        if (sidx < nsplit) {
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(base + ((size_t)sidx * KT + k) * n);
                acc[k].x += v.x; acc[k].y += v.y; acc[k].z += v.z; acc[k].w += v.w;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) *reinterpret_cast<float4*>(&red[y][k][4 * x]) = acc[k];
    __syncthreads();
    const int tid = y * 16 + x;
    const size_t nleft = n - i0 < 64 ? n - i0 : 64;   // (n is a multiple of 4, not necessarily of 64)
    for (int o = tid; o < (int)nleft * KT; o += 256) {
        const int il = o / KT, k = o - il * KT;
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][k][il];
        dw[i0 * KT + o] = t;
    }
}

int wgrad_nci(const TqConvDesc* d) {
    // 64-channel chunks where both concat sources are made of whole chunks (every 1-D config: 64 | C); TQDNE_WGRAD_NCI=1 forces 32
    static const int forced = [] { const char* e = getenv("TQDNE_WGRAD_NCI"); return e ? atoi(e) : 0; }();
    if (forced == 1) return 1;
    // Measured (B = 64, paper UNet): the ResBlocks' 1x1 skip convs gain 15-29 % from the 64-channel chunk (their MFMA phase is a
    // fifth of a k = 5 one, so the dy staging dominates); k = 5 loses 35-50 % (160 accumulator registers: 17 spills and no room
    // for the staging prefetch), k = 3 loses 16-34 %, the attention projections are neutral -> 64 only for k = 1
    if (d->ktaps != 1 && forced != 2) return 1;
    // round 5: 128-channel chunks for the 1x1 convs whose sources are made of whole ones (dy is then staged C_in / 128 times)
    if (d->ktaps == 1 && d->stride == 1 && !d->upsample && forced != 2 && d->C_in0 % 128 == 0 && d->C_in1 % 128 == 0) return 4;
    return (d->C_in0 % 64 == 0 && d->C_in1 % 64 == 0) ? 2 : 1;
}

// The 8-wave form (W8, see wgrad_kernel) serves the k = 3 / k = 5 launches whose sources are made of whole 64-channel chunks and whose
// output channels fill the 128-channel tile; TQDNE_WGRAD_W8=0 keeps round 3's 4-wave kernel everywhere.
bool wgrad_w8(const TqConvDesc* d) {
    static const int sw = [] { const char* e = getenv("TQDNE_WGRAD_W8"); return e ? atoi(e) : 1; }();
    if (!sw || d->ktaps == 1) return false;
    return d->C_in0 % 64 == 0 && d->C_in1 % 64 == 0 && d->C_out % 128 == 0;
}

// The four-wave form of the W8 layout for convs with exactly 64 output channels (H64, see wgrad_kernel): k = 3 / k = 5, stride 1, sources
// made of whole 64-channel chunks; TQDNE_WGRAD_H64=0 keeps the 128-channel tile for them (A/B switch).
bool wgrad_h64(const TqConvDesc* d) {
    static const int sw = [] { const char* e = getenv("TQDNE_WGRAD_H64"); return e ? atoi(e) : 1; }();
    if (!sw || d->ktaps == 1 || d->stride != 1 || d->upsample) return false;
    return d->C_in0 % 64 == 0 && d->C_in1 % 64 == 0 && d->C_out == 64;
}

// Workgroups of the weight-gradient kernel the device holds at once (occupancy of the instantiation x compute units), per
// (taps, stride, upsample, chunk width); queried once.  The (b, t) reduction is split over as many workgroups as fill ONE
// residency round: with the former fixed target of 768 the paper UNet's launches were 1.46 rounds of 512 resident workgroups
// (256 -> 256, k = 5: 752), i.e. the second round ran on half the chip (rocprofv3: 1.28 waves per SIMD on average).
template <int KT, int STRIDE, int UPS>
size_t wgrad_lds(int nci, bool w8) {
    constexpr int XR = (STRIDE == 1) ? (WG_TT + KT - 1) : (2 * WG_TT + 1);
    const int ncx = w8 ? 2 : nci;
    return 2 * WG_TT * ((nci == 1 || nci == 4) ? WG_DY_STRIDE1 : WG_DY_STRIDE2) + 2 * XR * 64 * ncx + 128 * sizeof(float);   // (+ the fused column sums)
}

template <int KT, int STRIDE, int UPS, int NCI, bool W8, bool H64 = false>
int wgrad_slots_of() {
    static const int slots = [] {
        const size_t sh = wgrad_lds<KT, STRIDE, UPS>(NCI, W8);
        int per_cu = 0, dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wgrad_kernel<KT, STRIDE, UPS, NCI, W8, H64>, (W8 && !H64) ? 512 : 256, sh) != hipSuccess ||
            per_cu < 1 || cus < 1) {
            (void)hipGetLastError();
            return (W8 && !H64) ? 256 : 512;   // (no device: what __launch_bounds__ asks for on MI355X's 256 compute units)
        }
        return per_cu * cus;
    }();
    return slots;
}

template <int KT, int STRIDE, int UPS>
int wgrad_slots_k(int nci, bool w8, bool h64 = false) {
    if constexpr (KT != 1 && STRIDE == 1 && UPS == 0) {
        if (h64) return wgrad_slots_of<KT, STRIDE, UPS, 1, true, true>();
    }
    if (w8) return wgrad_slots_of<KT, STRIDE, UPS, 1, true>();
    if constexpr (KT == 1 && STRIDE == 1 && UPS == 0) {
        if (nci == 4) return wgrad_slots_of<KT, STRIDE, UPS, 4, false>();
    }
    return nci == 2 ? wgrad_slots_of<KT, STRIDE, UPS, 2, false>() : wgrad_slots_of<KT, STRIDE, UPS, 1, false>();
}

int wgrad_slots(const TqConvDesc* d, int nci, bool w8, bool h64) {
    if (d->stride == 2) return wgrad_slots_k<3, 2, 0>(nci, w8);
    if (d->upsample) return d->ktaps == 5 ? wgrad_slots_k<5, 1, 1>(nci, w8) : wgrad_slots_k<3, 1, 1>(nci, w8);
    if (d->ktaps == 5) return wgrad_slots_k<5, 1, 0>(nci, w8, h64);
    if (d->ktaps == 3) return wgrad_slots_k<3, 1, 0>(nci, w8, h64);
    return wgrad_slots_k<1, 1, 0>(nci, false);
}

// ``wide``: the forms with a 64-channel xhat tile shared by two groups of waves (W8, H64) are allowed; the fused column sums are not
// built into them, so a launch that carries column sums plans -- and runs -- the four-wave kernel of 128 output channels
void wgrad_plan(const TqConvDesc* d, bool wide, int& n_cotiles, int& n_cichunks, int& n_ttiles, int& nsplit, int& ups) {
    const int nci = wgrad_nci(d);
    const bool h64 = wide && wgrad_h64(d);
    const bool w8 = wide && (h64 || wgrad_w8(d));
    n_cotiles = h64 ? 1 : (d->C_out + 127) / 128;
    n_cichunks = (d->C_in0 + d->C_in1) / (32 * (w8 ? 2 : nci));
    n_ttiles = (d->T_out + WG_TT - 1) / WG_TT;
    const int U = d->B * n_ttiles;
    const int ntiles = n_cotiles * n_cichunks;
    static const int forced = [] { const char* e = getenv("TQDNE_WGRAD_SLOTS"); return e ? atoi(e) : 0; }();   // (A/B switch)
    const int slots = forced > 0 ? forced : wgrad_slots(d, nci, w8, h64);
    int want = slots / ntiles;   // splits per output tile: the grid fills one round of resident workgroups, not more
    if (want < 1) want = 1;
    if (want > U) want = U;
    ups = (U + want - 1) / want;
    nsplit = (U + ups - 1) / ups;
}

template <int KT, int STRIDE, int UPS>
int launch_wgrad(const WgArgs& a, int nci, bool w8, bool h64, hipStream_t stream) {
    const size_t sh = wgrad_lds<KT, STRIDE, UPS>(nci, w8 || h64);
    const unsigned grid = (unsigned)(a.n_cotiles * a.n_cichunks * a.nsplit);
    if (h64) {
        if constexpr (KT != 1 && STRIDE == 1 && UPS == 0) {
            hipLaunchKernelGGL((wgrad_kernel<KT, STRIDE, UPS, 1, true, true>), dim3(grid), dim3(256), sh, stream, a);   // (54 KB of LDS)
        } else {
            return TQ_ERR_SHAPE;
        }
    } else if (w8) {
        static const bool raised = [] {   // (the 8-wave form's LDS images exceed the 64 KB default of a dynamic allocation)
            return hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<KT, STRIDE, UPS, 1, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        }();
        (void)raised;
        hipLaunchKernelGGL((wgrad_kernel<KT, STRIDE, UPS, 1, true>), dim3(grid), dim3(512), sh, stream, a);
    } else if (nci == 4) {
        if constexpr (KT == 1 && STRIDE == 1 && UPS == 0) {
            static const bool raised4 = [] {   // (70 KB of LDS images: above the 64 KB default of a dynamic allocation)
                return hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<1, 1, 0, 4, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
            }();
            (void)raised4;
            hipLaunchKernelGGL((wgrad_kernel<1, 1, 0, 4, false>), dim3(grid), dim3(256), sh, stream, a);
        } else {
            return TQ_ERR_SHAPE;
        }
    } else if (nci == 2) {
        hipLaunchKernelGGL((wgrad_kernel<KT, STRIDE, UPS, 2, false>), dim3(grid), dim3(256), sh, stream, a);
    } else {
        hipLaunchKernelGGL((wgrad_kernel<KT, STRIDE, UPS, 1, false>), dim3(grid), dim3(256), sh, stream, a);
    }
    TQ_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" size_t tq_conv1d_bwd_weight_workspace(const TqConvDesc* d) {
    if (!d) return 0;
    int a, b, c, nsplit, nsplit4, ups;   // (the larger of the two plans: with and without fused column sums)
    wgrad_plan(d, true, a, b, c, nsplit, ups);
    wgrad_plan(d, false, a, b, c, nsplit4, ups);
    return (size_t)(nsplit > nsplit4 ? nsplit : nsplit4) * d->ktaps * d->C_out * (d->C_in0 + d->C_in1) * sizeof(float);
}

extern "C" int tq_conv1d_bwd_weight(const TqConvDesc* d, const float* dy, const float* x0, const float* x1,
                                    const float* gscale, const float* gshift, float* dw, void* workspace, size_t ws_bytes,
                                    hipStream_t stream) {
    return tq_conv1d_bwd_weight_colsum(d, dy, x0, x1, gscale, gshift, dw, workspace, ws_bytes, nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int tq_conv1d_bwd_weight_colsum(const TqConvDesc* d, const float* dy, const float* x0, const float* x1,
                                           const float* gscale, const float* gshift, float* dw, void* workspace, size_t ws_bytes,
                                           float* colsum_bc, int bc_stride, float* colsum_c, float* colsum_c2,
                                           hipStream_t stream) {
    if (!d || !dy || !x0 || !dw || !workspace) return TQ_ERR_ARG;
    if (colsum_c2 && !colsum_c) return TQ_ERR_ARG;
    if (d->C_in0 <= 0 || d->C_in0 % 32 || d->C_in1 < 0 || d->C_in1 % 32 || d->C_out <= 0 || d->C_out % 32) return TQ_ERR_SHAPE;
    if (d->C_in1 > 0 && !x1) return TQ_ERR_ARG;
    if ((d->flags & TQ_CONV_GN) && (!gscale || !gshift)) return TQ_ERR_ARG;
    if (ws_bytes < tq_conv1d_bwd_weight_workspace(d)) return TQ_ERR_ARG;
    WgArgs a;
    a.dy = dy; a.x0 = x0; a.x1 = x1; a.gscale = gscale; a.gshift = gshift; a.slab = reinterpret_cast<float*>(workspace);
    a.B = d->B; a.T_in = d->T_in; a.T_out = d->T_out; a.C0 = d->C_in0; a.C1 = d->C_in1; a.C_out = d->C_out;
    a.flags = d->flags;
    a.cs_bc = colsum_bc; a.cs_stride = bc_stride; a.cs_c = colsum_c; a.cs_c2 = colsum_c2;
    const bool wide = !(colsum_bc || colsum_c);   // (the fused column sums live in the four-wave kernel of 128 output channels)
    wgrad_plan(d, wide, a.n_cotiles, a.n_cichunks, a.n_ttiles, a.nsplit, a.units_per_split);
    static const bool plain_order = [] { const char* e = getenv("TQDNE_WGRAD_XCD"); return e && atoi(e) == 0; }();
    if (plain_order) a.flags |= TQ_WGRAD_PLAIN_ORDER;
    a.drop_site = d->dropout_site; a.drop_seed = d->dropout_seed;
    float pdrop = d->dropout_p;
    if (!(d->flags & TQ_CONV_DROPOUT) || pdrop <= 0.f) { a.flags &= ~TQ_CONV_DROPOUT; pdrop = 0.f; }
    a.drop_thresh = (uint32_t)((double)pdrop * 4294967296.0);
    a.drop_scale = 1.0f / (1.0f - pdrop);
    int rc;
    const int nci = wgrad_nci(d);
    const bool h64 = wide && wgrad_h64(d);
    const bool w8 = wide && !h64 && wgrad_w8(d);
    if (d->stride == 2) {
        if (d->ktaps != 3) return TQ_ERR_SHAPE;
        rc = launch_wgrad<3, 2, 0>(a, nci, w8, false, stream);
    } else if (d->upsample) {
        if (d->ktaps == 5) rc = launch_wgrad<5, 1, 1>(a, nci, w8, false, stream);
        else if (d->ktaps == 3) rc = launch_wgrad<3, 1, 1>(a, nci, w8, false, stream);
        else return TQ_ERR_SHAPE;
    } else {
        if (d->ktaps == 5) rc = launch_wgrad<5, 1, 0>(a, nci, w8, h64, stream);
        else if (d->ktaps == 3) rc = launch_wgrad<3, 1, 0>(a, nci, w8, h64, stream);
        else if (d->ktaps == 1) rc = launch_wgrad<1, 1, 0>(a, nci, false, false, stream);
        else return TQ_ERR_SHAPE;
    }
    if (rc) return rc;
    const size_t n = (size_t)d->C_out * (d->C_in0 + d->C_in1);
    const dim3 rgrid((unsigned)((n + 63) / 64)), rblock(16, 16);
    if (d->ktaps == 5) hipLaunchKernelGGL(wgrad_reduce_kernel<5>, rgrid, rblock, 0, stream, a.slab, dw, a.nsplit, n);
    else if (d->ktaps == 3) hipLaunchKernelGGL(wgrad_reduce_kernel<3>, rgrid, rblock, 0, stream, a.slab, dw, a.nsplit, n);
    else hipLaunchKernelGGL(wgrad_reduce_kernel<1>, rgrid, rblock, 0, stream, a.slab, dw, a.nsplit, n);
    TQ_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================
// GroupNorm32 backward.  With u = a_c*x + s_c, g = dL/du (already chained through SiLU/dropout by the dgrad
// epilogue) and per-channel sums S1 = sum_t g, S2 = sum_t g*x:
//   P1 = sum_{c in grp} gamma_c*S1_c,  P2 = rstd * sum_{c in grp} gamma_c*(S2_c - mean*S1_c),  N = (C/32)*T
//   dx = (rstd*gamma_c) * g  -  (rstd^2 * P2 / N) * x  +  (rstd^2 * P2 * mean / N - rstd * P1 / N)
//   dgamma_c += rstd*(S2_c - mean*S1_c),  dbeta_c += S1_c          (atomics over the batch)
// =================================================================================================
namespace {
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ gst, const float* __restrict__ mr,
                                                              const float* __restrict__ gamma, int C, int T, int nslots,
                                                              float* __restrict__ cA, float* __restrict__ cB,
                                                              float* __restrict__ cC, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
    extern __shared__ double sh[];  // [nsub][Cb][2] partial sums (then [Cb][2] totals in place), then [groups of this block][2]
    // grid (B, NP): block (b, part) owns the channels [part * Cb, (part + 1) * Cb) = 32 / NP whole groups of sample b (groups are
    // independent).  Round 4: one block per sample was a latency chain (slot loads -> barrier -> G serial fp64 steps per group ->
    // barrier -> C atomics): 18 us per launch, 51 launches per training step.
    const int b = blockIdx.x;
    const int Cb = C / (int)gridDim.y;
    const int c0 = (int)blockIdx.y * Cb;
    // Slot sums: nsub threads per channel, each walking its slots four at a time with the four loads in flight together.  Fixed
    // summation order: deterministic.
    int nsub = Cb <= 128 ? (int)blockDim.x / Cb : 1;
    if (nsub > nslots) nsub = nslots;
    double* cs = sh;
    double* gp = sh + 2 * Cb * nsub;
    for (int idx = threadIdx.x; idx < Cb * nsub; idx += blockDim.x) {
        const int c = idx % Cb, sub = idx / Cb;
        const float* pp = gst + ((size_t)b * nslots * C + c0 + c) * 2;
        double s1 = 0.0, s2 = 0.0;
        int s = sub;
        for (; s + 3 * nsub < nslots; s += 4 * nsub) {
            const float2 v0 = *reinterpret_cast<const float2*>(pp + (size_t)s * C * 2);
            const float2 v1 = *reinterpret_cast<const float2*>(pp + (size_t)(s + nsub) * C * 2);
            const float2 v2 = *reinterpret_cast<const float2*>(pp + (size_t)(s + 2 * nsub) * C * 2);
            const float2 v3 = *reinterpret_cast<const float2*>(pp + (size_t)(s + 3 * nsub) * C * 2);
            s1 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
            s2 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
        }
        for (; s < nslots; s += nsub) {
            const float2 v = *reinterpret_cast<const float2*>(pp + (size_t)s * C * 2);
            s1 += (double)v.x; s2 += (double)v.y;
        }
        cs[2 * idx] = s1; cs[2 * idx + 1] = s2;
    }
    __syncthreads();
    if (nsub > 1) {
        double t1 = 0.0, t2 = 0.0;
        const int c = threadIdx.x;
        if (c < Cb)
            for (int sub = 0; sub < nsub; ++sub) { t1 += cs[2 * (sub * Cb + c)]; t2 += cs[2 * (sub * Cb + c) + 1]; }
        __syncthreads();
        if (c < Cb) { cs[2 * c] = t1; cs[2 * c + 1] = t2; }
        __syncthreads();
    }
    const int G = C / GN_GROUPS;
    const int ngrp = Cb / G, g0 = c0 / G;
    if ((int)threadIdx.x < ngrp) {
        const int gl = threadIdx.x, g = g0 + gl;
        const double mean = (double)mr[((size_t)b * GN_GROUPS + g) * 2], rstd = (double)mr[((size_t)b * GN_GROUPS + g) * 2 + 1];
        double p1 = 0.0, p2 = 0.0;
        for (int j = 0; j < G; ++j) {
            const int cl = gl * G + j;
            const double gm = (double)gamma[c0 + cl];
            p1 += gm * cs[2 * cl];
            p2 += gm * (cs[2 * cl + 1] - mean * cs[2 * cl]);
        }
        p2 *= rstd;
        const double n = (double)G * (double)T;
        gp[2 * gl] = -rstd * rstd * p2 / n;                             // coefficient of x
        gp[2 * gl + 1] = rstd * rstd * p2 * mean / n - rstd * p1 / n;   // constant
    }
    __syncthreads();
    for (int cl = threadIdx.x; cl < Cb; cl += blockDim.x) {
        const int c = c0 + cl;
        const int g = c / G, gl = g - g0;
        const float mean = mr[((size_t)b * GN_GROUPS + g) * 2], rstd = mr[((size_t)b * GN_GROUPS + g) * 2 + 1];
        cA[(size_t)b * C + c] = rstd * gamma[c];
        cB[(size_t)b * C + c] = (float)gp[2 * gl];
        cC[(size_t)b * C + c] = (float)gp[2 * gl + 1];
        atomicAdd(dgamma + c, (float)((double)rstd * (cs[2 * cl + 1] - (double)mean * cs[2 * cl])));
        atomicAdd(dbeta + c, (float)cs[2 * cl]);
    }
}

// dx (+)= A[b,c]*g + Bc[b,c]*x + Cc[b,c] (+ r)   for one concat source occupying channels [coff, coff+Cs) of the coefficients
__global__ void gn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ r,
                                    const float* __restrict__ cA, const float* __restrict__ cB, const float* __restrict__ cC,
                                    float* __restrict__ dx, int T, int Cs, int Ctot, int coff, int accum, size_t n4) {
    const int c4n = Cs >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const size_t bt = i / c4n;
        const int b = (int)(bt / T);
        const size_t ko = (size_t)b * Ctot + coff + 4 * c4;
        const float4 a = *reinterpret_cast<const float4*>(cA + ko);
        const float4 bb = *reinterpret_cast<const float4*>(cB + ko);
        const float4 cc = *reinterpret_cast<const float4*>(cC + ko);
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        float4 o = make_float4(a.x * gv.x + bb.x * xv.x + cc.x, a.y * gv.y + bb.y * xv.y + cc.y,
                               a.z * gv.z + bb.z * xv.z + cc.z, a.w * gv.w + bb.w * xv.w + cc.w);
        if (r) { const float4 rv = reinterpret_cast<const float4*>(r)[i]; o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w; }
        if (accum) { const float4 ov = reinterpret_cast<const float4*>(dx)[i]; o.x += ov.x; o.y += ov.y; o.z += ov.z; o.w += ov.w; }
        reinterpret_cast<float4*>(dx)[i] = o;
    }
}

// max|.| of a gradient tensor (feeds TqConvBwdDesc.dy_amax) is collected by atomic max into AMAX_WAYS words, one per 128-byte line,
// the workgroup picking its word by index: atomics on ONE address serialise at the memory side at ~11 ns each (measured: with one
// atomic per wave on a single word the 2048 atomics of a launch cost more than streaming the 67 MB tensor did), so one atomic per
// WORKGROUP on one of 16 lines; the consumer takes the maximum of the 16 words.
constexpr int AMAX_WAYS = TQ_AMAX_WAYS;
constexpr int AMAX_STRIDE = TQ_AMAX_STRIDE;   // words between the ways (128 bytes)

// Epilogue shared by colsum_kernel and gn_bwd_apply_cs_kernel: the workgroup's per-thread partial column sums `a` (float4 column c4,
// row lane tr) and partial max|v| `mx` -> out_bc[b*stride + c] / out_c[c] / out_c2[c] (atomic adds into zeroed buffers) and
// *amax_out (atomic max of bit patterns: non-negative floats order like their bits; NaN sorts above everything).
template <int NT>
__device__ __forceinline__ void colsum_finish(float* red, const float4& a, float mx, bool active, int c4n, int nrow, int c4, int tr, int C,
                                              int b, float sc, float* __restrict__ out_bc, int bc_stride, float* __restrict__ out_c,
                                              float* __restrict__ out_c2, unsigned* __restrict__ amax_out) {
    if (active) *reinterpret_cast<float4*>(red + (tr * c4n + c4) * 4) = a;
    unsigned* wmax = reinterpret_cast<unsigned*>(red + (size_t)nrow * c4n * 4);   // [NT / 64] per-wave maxima behind the sums
    if (amax_out) {
        unsigned m = __float_as_uint(mx);   // (mx >= 0 or +inf)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    }
    __syncthreads();
    if (amax_out && threadIdx.x == 0) {
        unsigned m = wmax[0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) m = max(m, wmax[w]);
        if (m != 0u) atomicMax(amax_out + (blockIdx.x % AMAX_WAYS) * AMAX_STRIDE, m);
    }
    for (int c = threadIdx.x; c < C; c += NT) {
        float s = 0.f;
        for (int r2 = 0; r2 < nrow; ++r2) s += red[(r2 * c4n + (c >> 2)) * 4 + (c & 3)];
        s *= sc;
        if (out_bc) atomicAdd(out_bc + (size_t)b * bc_stride + c, s);
        if (out_c) atomicAdd(out_c + c, s);
        if (out_c2) atomicAdd(out_c2 + c, s);
    }
}
__device__ __forceinline__ float amax4(float m, const float4& v) {
    // (fmaxf drops NaN operands; a NaN gradient must stay visible in the scale: carried as +inf, which scales the tensor to ~0 / NaN)
    const float t = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    const bool bad = (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
    return bad ? __builtin_inff() : fmaxf(m, t);
}

// out_bc[b*stride + c] += sum_t dy[b,t,c];  out_c[c] += sum_{b,t} dy   (both optional; atomics into zeroed buffers)
template <int NT>
__global__ __launch_bounds__(NT) void colsum_kernel(const float* __restrict__ dy, int T, int C, float* __restrict__ out_bc,
                                                     int bc_stride, float* __restrict__ out_c, float* __restrict__ out_c2,
                                                     const float* __restrict__ bscale, unsigned* __restrict__ amax_out, int rpw) {
    // rpw: rows (positions) per workgroup
    extern __shared__ float red[];
    const int nsl = (T + rpw - 1) / rpw;
    const int slot = blockIdx.x % nsl, b = blockIdx.x / nsl;
    const int c4n = C >> 2;
    const int nrow = NT / c4n > 0 ? NT / c4n : 1;
    const int c4 = threadIdx.x % c4n, tr = threadIdx.x / c4n;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float mx = 0.f;
    const bool active = tr < nrow && threadIdx.x < nrow * c4n;
    if (active) {
        const int nvalid = min(rpw, T - slot * rpw);
        const float* src = dy + ((size_t)b * T + (size_t)slot * rpw) * C + 4 * c4;
        int tl = tr;
        // 8 rows in flight per thread (a dependent load-add chain over up to 128 rows ran at 2 TB/s)
        for (; tl + 7 * nrow < nvalid; tl += 8 * nrow) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4*>(src + (size_t)(tl + q * nrow) * C);
#pragma unroll
            for (int q = 0; q < 8; ++q) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
            if (amax_out) {
#pragma unroll
                for (int q = 0; q < 8; ++q) mx = amax4(mx, v[q]);
            }
        }
        for (; tl < nvalid; tl += nrow) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)tl * C);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            if (amax_out) mx = amax4(mx, v);
        }
    }
    colsum_finish<NT>(red, a, mx, active, c4n, nrow, c4, tr, C, b, bscale ? bscale[b] : 1.0f, out_bc, bc_stride, out_c, out_c2, amax_out);
}

// gn_bwd_apply with the column sums (and max|.|) of the tensor it writes fused in: the consumers of dx -- the bias / time-embedding
// gradients of the conv whose output gradient dx is, and the power-of-two scale of that conv's fp16-range data gradient -- otherwise
// need a pass of their own over dx (63 tq_colsum launches per training step, 1.3 ms at B = 64).  Same tiling as colsum_kernel: a
// workgroup owns `rpw` rows of one sample, thread = (float4 column, row lane).
template <int NT>
__global__ __launch_bounds__(NT) void gn_bwd_apply_cs_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                              const float* __restrict__ r, const float* __restrict__ cA,
                                                              const float* __restrict__ cB, const float* __restrict__ cC,
                                                              float* __restrict__ dx, int T, int Cs, int Ctot, int coff, int accum,
                                                              float* __restrict__ out_bc, int bc_stride, float* __restrict__ out_c,
                                                              float* __restrict__ out_c2, unsigned* __restrict__ amax_out, int rpw) {
    extern __shared__ float red[];
    const int nsl = (T + rpw - 1) / rpw;
    const int slot = blockIdx.x % nsl, b = blockIdx.x / nsl;
    const int c4n = Cs >> 2;
    const int nrow = NT / c4n > 0 ? NT / c4n : 1;
    const int c4 = threadIdx.x % c4n, tr = threadIdx.x / c4n;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float mx = 0.f;
    const bool active = tr < nrow && threadIdx.x < nrow * c4n;
    if (active) {
        const int nvalid = min(rpw, T - slot * rpw);
        const size_t ko = (size_t)b * Ctot + coff + 4 * c4;
        const float4 ka = *reinterpret_cast<const float4*>(cA + ko);
        const float4 kb = *reinterpret_cast<const float4*>(cB + ko);
        const float4 kc = *reinterpret_cast<const float4*>(cC + ko);
        const size_t base = ((size_t)b * T + (size_t)slot * rpw) * Cs + 4 * c4;
        auto one = [&](const float4& gv, const float4& xv, const float4& rv, const float4& ov) -> float4 __attribute__((always_inline)) {
            float4 o = make_float4(ka.x * gv.x + kb.x * xv.x + kc.x, ka.y * gv.y + kb.y * xv.y + kc.y,
                                   ka.z * gv.z + kb.z * xv.z + kc.z, ka.w * gv.w + kb.w * xv.w + kc.w);
            o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
            o.x += ov.x; o.y += ov.y; o.z += ov.z; o.w += ov.w;
            return o;
        };
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        int tl = tr;
        for (; tl + 3 * nrow < nvalid; tl += 4 * nrow) {   // 4 rows = 8-16 loads in flight per thread
            float4 gv[4], xv[4], rv[4], ov[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t o = base + (size_t)(tl + q * nrow) * Cs;
                gv[q] = *reinterpret_cast<const float4*>(g + o);
                xv[q] = *reinterpret_cast<const float4*>(x + o);
                rv[q] = r ? *reinterpret_cast<const float4*>(r + o) : z4;
                ov[q] = accum ? *reinterpret_cast<const float4*>(dx + o) : z4;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 o4 = one(gv[q], xv[q], rv[q], ov[q]);
                *reinterpret_cast<float4*>(dx + base + (size_t)(tl + q * nrow) * Cs) = o4;
                a.x += o4.x; a.y += o4.y; a.z += o4.z; a.w += o4.w;
                mx = amax4(mx, o4);
            }
        }
        for (; tl < nvalid; tl += nrow) {
            const size_t o = base + (size_t)tl * Cs;
            const float4 o4 = one(*reinterpret_cast<const float4*>(g + o), *reinterpret_cast<const float4*>(x + o),
                                  r ? *reinterpret_cast<const float4*>(r + o) : z4, accum ? *reinterpret_cast<const float4*>(dx + o) : z4);
            *reinterpret_cast<float4*>(dx + o) = o4;
            a.x += o4.x; a.y += o4.y; a.z += o4.z; a.w += o4.w;
            mx = amax4(mx, o4);
        }
    }
    colsum_finish<NT>(red, a, mx, active, c4n, nrow, c4, tr, Cs, b, 1.0f, out_bc, bc_stride, out_c, out_c2, amax_out);
}

__global__ void zero_stuff_kernel(const float* __restrict__ dy, float* __restrict__ out, int T_out, int T_in, int C, size_t n4) {
    const int c4n = C >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const size_t bu = i / c4n;
        const int u = (int)(bu % T_in);
        const size_t b = bu / T_in;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(u & 1) && (u >> 1) < T_out) v = reinterpret_cast<const float4*>(dy)[(b * T_out + (u >> 1)) * c4n + c4];
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

__global__ void pair_sum_kernel(const float* __restrict__ dup, float* __restrict__ dx, int T, int C, int accum, size_t n4) {
    const int c4n = C >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const size_t bt = i / c4n;
        const int t = (int)(bt % T);
        const size_t b = bt / T;
        const float4 a = reinterpret_cast<const float4*>(dup)[(b * 2 * T + 2 * t) * c4n + c4];
        const float4 c = reinterpret_cast<const float4*>(dup)[(b * 2 * T + 2 * t + 1) * c4n + c4];
        float4 o = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
        if (accum) { const float4 ov = reinterpret_cast<const float4*>(dx)[i]; o.x += ov.x; o.y += ov.y; o.z += ov.z; o.w += ov.w; }
        reinterpret_cast<float4*>(dx)[i] = o;
    }
}

inline unsigned ew_grid4(size_t n4) {
    size_t g = (n4 + 255) / 256;
    return (unsigned)(g > 4096 ? 4096 : (g ? g : 1));
}
}  // namespace

extern "C" int tq_gn_bwd_finalize(const float* gstats, const float* mean_rstd, const float* gamma, int B, int C, int T,
                                  float* coef_a, float* coef_b, float* coef_c, float* dgamma, float* dbeta,
                                  hipStream_t stream) {
    if (!gstats || !mean_rstd || !gamma || !coef_a || !coef_b || !coef_c || !dgamma || !dbeta) return TQ_ERR_ARG;
    if (B <= 0 || T <= 0 || C <= 0 || C % GN_GROUPS) return TQ_ERR_SHAPE;
    const int nslots = (T + STAT_SLOT - 1) / STAT_SLOT;
    static const int forced = [] { const char* e = getenv("TQDNE_GNBWD_PARTS"); return e ? atoi(e) : 0; }();   // (A/B switch: 1 = round 3's grid)
    const int NP = (forced == 1 || forced == 2 || forced == 4 || forced == 8) ? forced : 4;   // 32 / NP whole groups per block
    const int Cb = C / NP;
    int nsub = Cb <= 128 ? 256 / Cb : 1;   // (as in the kernel)
    if (nsub > nslots) nsub = nslots;
    const size_t sh = (size_t)(2 * Cb * nsub + 2 * GN_GROUPS) * sizeof(double);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B, NP), dim3(256), sh, stream, gstats, mean_rstd, gamma, C, T, nslots, coef_a,
                       coef_b, coef_c, dgamma, dbeta);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_gn_bwd_apply(const float* g, const float* x, const float* r, const float* coef_a, const float* coef_b,
                               const float* coef_c, float* dx, int B, int T, int C_src, int C_total, int c_offset, int accumulate,
                               hipStream_t stream) {
    if (!g || !x || !coef_a || !coef_b || !coef_c || !dx) return TQ_ERR_ARG;
    if (B <= 0 || T <= 0 || C_src <= 0 || C_src % 4 || c_offset % 4 || c_offset + C_src > C_total) return TQ_ERR_SHAPE;
    const size_t n4 = (size_t)B * T * (C_src / 4);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(ew_grid4(n4)), dim3(256), 0, stream, g, x, r, coef_a, coef_b, coef_c, dx, T,
                       C_src, C_total, c_offset, accumulate, n4);
    TQ_CHECK_LAUNCH();
    return 0;
}

// Tiling of the column-sum kernels (colsum_kernel, gn_bwd_apply_cs_kernel): NT threads per workgroup, `rows` rows of one sample per
// workgroup -- as many rows as still leaves `min_wgs` workgroups.  A/B switches: TQDNE_COLSUM_THREADS (256 | 1024), TQDNE_COLSUM_ROWS,
// TQDNE_COLSUM_WGS.
struct CsTiling { int nt, rows; };
static CsTiling colsum_tiling(int B, int T) {
    static const int f_nt = [] { const char* e = getenv("TQDNE_COLSUM_THREADS"); return e ? atoi(e) : 0; }();
    static const int f_rows = [] { const char* e = getenv("TQDNE_COLSUM_ROWS"); return e ? atoi(e) : 0; }();
    static const int f_wgs = [] { const char* e = getenv("TQDNE_COLSUM_WGS"); return e ? atoi(e) : 0; }();
    CsTiling t;
    t.nt = f_nt == 1024 ? 1024 : 256;
    // (measured, tools/bwd_micro.py with the switches above, B = 64: 256 workgroups of 256 threads beat 512 / 1024 / 2048 and the
    // 1024-thread form on every level -- fused apply 47 us next to 44 us for the plain apply; each workgroup ends in C atomic adds, and
    // the total sums' addresses are shared by ALL workgroups)
    const size_t min_wgs = f_wgs > 0 ? (size_t)f_wgs : 256;
    int rpw = STAT_SLOT;
    while (rpw < 4096 && (size_t)B * ((T + 2 * rpw - 1) / (2 * rpw)) >= min_wgs) rpw <<= 1;
    t.rows = f_rows > 0 ? f_rows : rpw;
    return t;
}

extern "C" int tq_colsum(const float* dy, int B, int T, int C, float* out_bc, int bc_stride, float* out_c, float* out_c2,
                         const float* bscale, uint32_t* amax_out, hipStream_t stream) {
    if (!dy || (!out_bc && !out_c && !out_c2 && !amax_out)) return TQ_ERR_ARG;
    if (B <= 0 || T <= 0 || C < 4 || C % 4 || C > 1024) return TQ_ERR_SHAPE;
    const CsTiling tl = colsum_tiling(B, T);
    const int nsl = (T + tl.rows - 1) / tl.rows;
    const int c4n = C / 4;
    const int nrow = tl.nt / c4n > 0 ? tl.nt / c4n : 1;
    const size_t sh = (size_t)nrow * c4n * 4 * sizeof(float) + (tl.nt / 64) * sizeof(unsigned);
    if (tl.nt == 1024)
        hipLaunchKernelGGL(colsum_kernel<1024>, dim3(B * nsl), dim3(1024), sh, stream, dy, T, C, out_bc, bc_stride, out_c, out_c2, bscale, amax_out, tl.rows);
    else
        hipLaunchKernelGGL(colsum_kernel<256>, dim3(B * nsl), dim3(256), sh, stream, dy, T, C, out_bc, bc_stride, out_c, out_c2, bscale, amax_out, tl.rows);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_gn_bwd_apply_colsum(const float* g, const float* x, const float* r, const float* coef_a, const float* coef_b,
                                      const float* coef_c, float* dx, int B, int T, int C_src, int C_total, int c_offset, int accumulate,
                                      float* colsum_bc, int bc_stride, float* colsum_c, float* colsum_c2, uint32_t* amax_out,
                                      hipStream_t stream) {
    if (!g || !x || !coef_a || !coef_b || !coef_c || !dx) return TQ_ERR_ARG;
    if (colsum_c2 && !colsum_c) return TQ_ERR_ARG;
    if (B <= 0 || T <= 0 || C_src <= 0 || C_src % 4 || C_src > 1024 || c_offset % 4 || c_offset + C_src > C_total) return TQ_ERR_SHAPE;
    if (!colsum_bc && !colsum_c && !amax_out)
        return tq_gn_bwd_apply(g, x, r, coef_a, coef_b, coef_c, dx, B, T, C_src, C_total, c_offset, accumulate, stream);
    const CsTiling tl = colsum_tiling(B, T);
    const int nsl = (T + tl.rows - 1) / tl.rows;
    const int c4n = C_src / 4;
    const int nrow = tl.nt / c4n > 0 ? tl.nt / c4n : 1;
    const size_t sh = (size_t)nrow * c4n * 4 * sizeof(float) + (tl.nt / 64) * sizeof(unsigned);
    if (tl.nt == 1024)
        hipLaunchKernelGGL(gn_bwd_apply_cs_kernel<1024>, dim3(B * nsl), dim3(1024), sh, stream, g, x, r, coef_a, coef_b, coef_c, dx, T, C_src,
                           C_total, c_offset, accumulate, colsum_bc, bc_stride, colsum_c, colsum_c2, amax_out, tl.rows);
    else
        hipLaunchKernelGGL(gn_bwd_apply_cs_kernel<256>, dim3(B * nsl), dim3(256), sh, stream, g, x, r, coef_a, coef_b, coef_c, dx, T, C_src,
                           C_total, c_offset, accumulate, colsum_bc, bc_stride, colsum_c, colsum_c2, amax_out, tl.rows);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_zero_stuff(const float* dy, float* out, int B, int T_out, int T_in, int C, hipStream_t stream) {
    if (!dy || !out || B <= 0 || T_out <= 0 || T_in <= 0 || C % 4) return TQ_ERR_ARG;
    const size_t n4 = (size_t)B * T_in * (C / 4);
    hipLaunchKernelGGL(zero_stuff_kernel, dim3(ew_grid4(n4)), dim3(256), 0, stream, dy, out, T_out, T_in, C, n4);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_pair_sum(const float* d_up, float* dx, int B, int T, int C, int accumulate, hipStream_t stream) {
    if (!d_up || !dx || B <= 0 || T <= 0 || C % 4) return TQ_ERR_ARG;
    const size_t n4 = (size_t)B * T * (C / 4);
    hipLaunchKernelGGL(pair_sum_kernel, dim3(ew_grid4(n4)), dim3(256), 0, stream, d_up, dx, T, C, accumulate, n4);
    TQ_CHECK_LAUNCH();
    return 0;
}

// Round 6: Upsample (blocks.py:56-66: nearest x2, then conv k = 5) trained in its two-phase k = 3 form (TQ_CONV_POLY2: even outputs see the
// taps A = (w0+w1, w2+w3, w4), odd ones B = (w0, w1+w2, w3+w4) of the un-upsampled rows).  Its weight gradient is the plain k = 3 weight
// gradient of that conv -- d W2 (2 C_out, C_in, 3) = [dA | dB], with the output gradient (B, 2T, C_out) read as (B, T, 2 C_out) -- folded
// back onto the five taps by the transpose of the tap sums:  dw0 = dA0 + dB0, dw1 = dA0 + dB1, dw2 = dA1 + dB1, dw3 = dA1 + dB2,
// dw4 = dA2 + dB2.  3/5 of the multiply-adds of the k = 5 gradient over the upsampled gather.
namespace {
__global__ __launch_bounds__(256) void poly_wgrad_fold_kernel(const float* __restrict__ dw2, float* __restrict__ dw, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;   // (co, ci) pair
    if (i >= n) return;
    const float* a = dw2 + (size_t)i * 3;
    const float* b = dw2 + ((size_t)n + i) * 3;
    const float a0 = a[0], a1 = a[1], a2 = a[2], b0 = b[0], b1 = b[1], b2 = b[2];
    float* o = dw + (size_t)i * 5;
    o[0] = a0 + b0; o[1] = a0 + b1; o[2] = a1 + b1; o[3] = a1 + b2; o[4] = a2 + b2;
}
}  // namespace

extern "C" int tq_upsample_poly_wgrad_fold(const float* dw2, float* dw, int C_out, int C_in, hipStream_t stream) {
    if (!dw2 || !dw || C_out <= 0 || C_in <= 0) return TQ_ERR_ARG;
    const int n = C_out * C_in;
    hipLaunchKernelGGL(poly_wgrad_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dw2, dw, n);
    TQ_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================
// Round 4: streaming forms of the stem weight gradient and of the head backward.  The round 1-3 kernels below (thread per output
// element, both operands from LDS: two LDS reads per FMA; 512-2048 workgroups x 960 atomics) took 260 us and 188 us at B = 64 for
// what is one pass over a 67 MB tensor.  Here a thread owns ONE wide-tensor channel and a run of positions: the wide tensor (dy /
// the head's input) is read straight from HBM, a wave fetching 256-byte rows, several rows in flight; the narrow operand (the <= 8
// input channels of the stem, the <= 4 output channels of the head) sits in LDS and is read as wave-uniform broadcasts through a
// sliding register window; the KT x narrow-channel partial sums live in registers across all units of the workgroup, are combined over
// the position segments in LDS and leave as ONE set of atomics per workgroup.
// =================================================================================================
namespace {
// dW[co][ci][k] += sum_{b,t} dy[b,t,co] * in_scale[b] * x[b,ci,t+k-pad].  256 % C_out == 0, C_in <= CI.
template <int KT, int CI>
__global__ __launch_bounds__(256) void stem_wgrad_stream_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ in_scale, float* __restrict__ dw,
                                                                float* __restrict__ part, int C_in, int T, int C_out, int nslots,
                                                                int nunits, int upw) {
    extern __shared__ float shm[];
    constexpr int PAD = KT / 2, TW = STAT_SLOT + KT - 1, BLK = 16;
    float* xs = shm;                  // [C_in][TW]
    float* red = xs + CI * TW;        // [nseg][C_in * KT][C_out]
    const int nseg = 256 / C_out, L = STAT_SLOT / nseg;
    const int co = threadIdx.x % C_out, seg = threadIdx.x / C_out;
    float acc[CI][KT];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[ci][k] = 0.f;
    const int u0 = blockIdx.x * upw, u1 = min(nunits, u0 + upw);
    for (int u = u0; u < u1; ++u) {
        const int slot = u % nslots, b = u / nslots;
        const int t0 = slot * STAT_SLOT;
        const float sc = in_scale ? in_scale[b] : 1.0f;
        __syncthreads();
        for (int i = threadIdx.x; i < C_in * TW; i += 256) {
            const int c = i / TW, j = i % TW;
            const int t = t0 - PAD + j;
            xs[c * TW + j] = (t >= 0 && t < T) ? x[((size_t)b * C_in + c) * T + t] * sc : 0.f;
        }
        __syncthreads();
        const int tb = t0 + seg * L;
        for (int tl = 0; tl < L; tl += BLK) {
            // (branch-free: rows past the signal load a clamped, valid row and are multiplied by 0 -- a predicated load puts control
            // flow between the loads and hipcc then drains them one by one, s_waitcnt vmcnt(0) at every join: measured 209 us)
            float d[BLK];
#pragma unroll
            for (int j = 0; j < BLK; ++j) {
                const int tj = tb + tl + j;
                const int tc = tj < T ? tj : T - 1;
                d[j] = dy[((size_t)b * T + tc) * C_out + co] * (tj < T ? 1.f : 0.f);
            }
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) {
                if (ci < C_in) {
                    float xw[BLK + KT - 1];
#pragma unroll
                    for (int j = 0; j < BLK + KT - 1; ++j) xw[j] = xs[ci * TW + seg * L + tl + j];   // wave-uniform: broadcast
#pragma unroll
                    for (int j = 0; j < BLK; ++j)
#pragma unroll
                        for (int k = 0; k < KT; ++k) acc[ci][k] = fmaf(d[j], xw[j + k], acc[ci][k]);
                }
            }
        }
    }
    __syncthreads();
    const int nck = C_in * KT;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int k = 0; k < KT; ++k)
            if (ci < C_in) red[(seg * nck + ci * KT + k) * C_out + co] = acc[ci][k];
    __syncthreads();
    for (int o = threadIdx.x; o < nck * C_out; o += 256) {
        float v = 0.f;
        for (int sg = 0; sg < nseg; ++sg) v += red[sg * nck * C_out + o];
        const int c = o % C_out, ck = o / C_out;   // ck = ci * KT + k
        // `part` given: this workgroup's sums go to its own row of the scratch (plain stores), partial_rows_sum_kernel adds the rows.
        // The 960 outputs share 30 cache lines, and atomics on one LINE serialise at ~11 ns each: 512 workgroups x 32 atomics per
        // line = 180 us -- that, not the 67 MB pass, was the time of this kernel and of head_bwd (measured 209-236 us with atomics).
        if (part) part[(size_t)blockIdx.x * (nck * C_out) + (size_t)c * nck + ck] = v;
        else atomicAdd(dw + (size_t)c * nck + ck, v);
    }
}

// out[i] += sum_r part[r][i]   (i < n, rows of n floats; a handful of workgroups: n is ~1000)
__global__ __launch_bounds__(256) void partial_rows_sum_kernel(const float* __restrict__ part, int nrows, int stride, int n,
                                                               float* __restrict__ out) {
    __shared__ float red2[4][64];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float a = 0.f;
    if (i < n) {
        int r = q;
        for (; r + 12 < nrows; r += 16)   // 4 loads in flight per thread
            a += (part[(size_t)r * stride + i] + part[(size_t)(r + 4) * stride + i]) +
                 (part[(size_t)(r + 8) * stride + i] + part[(size_t)(r + 12) * stride + i]);
        for (; r < nrows; r += 4) a += part[(size_t)r * stride + i];
    }
    red2[q][threadIdx.x & 63] = a;
    __syncthreads();
    if (q == 0 && i < n) out[i] += (red2[0][threadIdx.x] + red2[1][threadIdx.x]) + (red2[2][threadIdx.x] + red2[3][threadIdx.x]);   // (threadIdx.x < 64 here)
}

// Head backward, C_out <= 4 (the network's 3 output channels): thread = (input channel ci, position segment).
//   dF = c_out[b] * dpred;   dW[co][ci][k] += sum_t dF[co][t] z[ci][t+k-pad];   db[co] += sum_t dF[co][t]
//   G[t][ci] = (sum_{co,k} W[co][ci][k] dF[co][t+pad-k]) * silu'(a h + s),  GN partial sums {sum G, sum G h} per 128-position slot
// Both sums pair the thread's own element z[ci][t] / G[t][ci] with the SAME 3 x KT values dF[co][t + pad - k].
// Round 5: MAXCO = 8 serves the 6-channel head of the reference's real data shape (3 x 4064 waveforms -> 6 x 4064 through the envelope
// representation), which used to take the round-3 kernel (atomics, a fifth of the speed).
template <int KT, int MAXCO = 4>
__global__ __launch_bounds__(256) void head_bwd_stream_kernel(const float* __restrict__ dpred, const float* __restrict__ c_out,
                                                              const float* __restrict__ h, const float* __restrict__ gscale,
                                                              const float* __restrict__ gshift, const float* __restrict__ w,
                                                              float* __restrict__ G, float* __restrict__ gstats,
                                                              float* __restrict__ dw, float* __restrict__ db, float* __restrict__ part,
                                                              int T, int C_in, int C_out, int nslots, int nunits, int upw) {
    extern __shared__ float shm[];
    constexpr int PAD = KT / 2, TW = STAT_SLOT + 2 * PAD, BLK = 8;
    float* dfs = shm;                          // [MAXCO][TW]  dF of positions t0 - pad .. t0 + 127 + pad
    float* red = dfs + MAXCO * TW;             // [nseg][max(2, MAXCO * KT)][C_in]
    const int nseg = 256 / C_in, L = STAT_SLOT / nseg;
    const int ci = threadIdx.x % C_in, seg = threadIdx.x / C_in;
    float wr[MAXCO][KT], aw[MAXCO][KT];
#pragma unroll
    for (int co = 0; co < MAXCO; ++co)
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            wr[co][k] = co < C_out ? w[((size_t)co * C_in + ci) * KT + k] : 0.f;
            aw[co][k] = 0.f;
        }
    float dbs = 0.f;   // thread co < C_out of segment 0 sums dF[co] over the slots' own positions
    const int u0 = blockIdx.x * upw, u1 = min(nunits, u0 + upw);
    for (int u = u0; u < u1; ++u) {
        const int slot = u % nslots, b = u / nslots;
        const int t0 = slot * STAT_SLOT;
        const float cs = c_out ? c_out[b] : 1.0f;
        __syncthreads();
        for (int i = threadIdx.x; i < C_out * TW; i += 256) {
            const int c = i / TW, j = i % TW;
            const int t = t0 - PAD + j;
            dfs[c * TW + j] = (t >= 0 && t < T) ? dpred[((size_t)b * C_out + c) * T + t] * cs : 0.f;
        }
        for (int i = C_out * TW + threadIdx.x; i < MAXCO * TW; i += 256) dfs[i] = 0.f;   // (unused output channels)
        __syncthreads();
        if ((int)threadIdx.x < C_out) {
            const int nv = min(STAT_SLOT, T - t0);
            for (int j = 0; j < nv; ++j) dbs += dfs[threadIdx.x * TW + PAD + j];
        }
        float ga = 1.f, gs = 0.f;
        if (gscale) { ga = gscale[(size_t)b * C_in + ci]; gs = gshift[(size_t)b * C_in + ci]; }
        const int tb = t0 + seg * L;
        const size_t rowb = ((size_t)b * T + tb) * C_in + ci;
        float s1 = 0.f, s2 = 0.f;
        for (int tl = 0; tl < L; tl += BLK) {
            float hv[BLK];
#pragma unroll
            for (int j = 0; j < BLK; ++j) {   // (branch-free clamped loads, see stem_wgrad_stream_kernel)
                const int tj = tb + tl + j;
                const int tc = tj < T ? tj : T - 1;
                hv[j] = h[((size_t)b * T + tc) * C_in + ci];
            }
            float g[BLK], z[BLK], ds[BLK];
#pragma unroll
            for (int j = 0; j < BLK; ++j) {
                g[j] = 0.f;
                if (gscale) { const float uu = ga * hv[j] + gs; z[j] = silu_f(uu); ds[j] = dsilu_f(uu); }
                else { z[j] = hv[j]; ds[j] = 1.f; }
                if (tb + tl + j >= T) z[j] = 0.f;   // (positions past the signal contribute nothing)
            }
#pragma unroll
            for (int co = 0; co < MAXCO; ++co) {
                // window: dF[co] at slot-relative positions (seg L + tl) - pad .. + BLK - 1 + pad  ->  dfs index + PAD
                float q[BLK + 2 * PAD];
#pragma unroll
                for (int j = 0; j < BLK + 2 * PAD; ++j) q[j] = dfs[co * TW + seg * L + tl + j];   // wave-uniform: broadcast
#pragma unroll
                for (int j = 0; j < BLK; ++j)
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        const float f = q[j + 2 * PAD - k];   // dF[co][t + pad - k]
                        aw[co][k] = fmaf(z[j], f, aw[co][k]);
                        g[j] = fmaf(wr[co][k], f, g[j]);
                    }
            }
#pragma unroll
            for (int j = 0; j < BLK; ++j) {
                if (tb + tl + j < T) {
                    const float gv = g[j] * ds[j];
                    G[rowb + (size_t)(tl + j) * C_in] = gv;
                    s1 += gv; s2 += gv * hv[j];
                }
            }
        }
        if (gstats) {   // the slot's partial sums: combine the position segments through LDS
            __syncthreads();
            red[(seg * 2 + 0) * C_in + ci] = s1;
            red[(seg * 2 + 1) * C_in + ci] = s2;
            __syncthreads();
            if (seg == 0) {
                float a1 = 0.f, a2 = 0.f;
                for (int sg = 0; sg < nseg; ++sg) { a1 += red[(sg * 2 + 0) * C_in + ci]; a2 += red[(sg * 2 + 1) * C_in + ci]; }
                float* st = gstats + (((size_t)b * nslots + slot) * C_in + ci) * 2;
                st[0] = a1; st[1] = a2;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int co = 0; co < MAXCO; ++co)
#pragma unroll
        for (int k = 0; k < KT; ++k) red[(seg * MAXCO * KT + co * KT + k) * C_in + ci] = aw[co][k];
    __syncthreads();
    for (int o = threadIdx.x; o < C_out * KT * C_in; o += 256) {
        const int c = o % C_in, ck = o / C_in;   // ck = co * KT + k
        float v = 0.f;
        for (int sg = 0; sg < nseg; ++sg) v += red[(sg * MAXCO * KT + ck) * C_in + c];
        const int co = ck / KT, k = ck % KT;
        if (part) part[(size_t)blockIdx.x * (C_out * C_in * KT + MAXCO) + ((size_t)co * C_in + c) * KT + k] = v;   // (see the stem kernel)
        else atomicAdd(dw + ((size_t)co * C_in + c) * KT + k, v);
    }
    if ((int)threadIdx.x < C_out) {
        if (part) part[(size_t)blockIdx.x * (C_out * C_in * KT + MAXCO) + C_out * C_in * KT + threadIdx.x] = dbs;
        else atomicAdd(db + threadIdx.x, dbs);
    }
}
}  // namespace

// =================================================================================================
// Stem weight gradient: dW[co,ci,k] += sum_{b,t} dy[b,t,co] * in_scale[b]*x[b,ci,t+k-pad];  db via tq_colsum.
// One workgroup per (b, 128-position slot); thread per output element; fp32 atomics into the zeroed gradient.
// =================================================================================================
namespace {
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ in_scale, float* __restrict__ dw,
                                                         int C_in, int T, int C_out, int KT, int nslots, int nunits, int upw) {
    extern __shared__ float shm[];
    const int PAD = KT / 2;
    const int TW = STAT_SLOT + KT - 1;
    float* xs = shm;                 // [C_in][TW]
    float* ds = xs + C_in * TW;      // [128][C_out + 1]
    const int LD = C_out + 1;
    const int nout = C_out * C_in * KT;
    constexpr int MAXO = 8;          // outputs per thread held in registers (nout <= 2048)
    float acc[MAXO];
#pragma unroll
    for (int q = 0; q < MAXO; ++q) acc[q] = 0.f;
    const int u0 = blockIdx.x * upw;
    const int u1 = min(nunits, u0 + upw);
    for (int u = u0; u < u1; ++u) {
        const int slot = u % nslots, b = u / nslots;
        const int t0 = slot * STAT_SLOT;
        const float sc = in_scale ? in_scale[b] : 1.0f;
        __syncthreads();
        for (int i = threadIdx.x; i < C_in * TW; i += 256) {
            const int c = i / TW, j = i % TW;
            const int t = t0 - PAD + j;
            xs[i] = (t >= 0 && t < T) ? x[((size_t)b * C_in + c) * T + t] * sc : 0.f;
        }
        for (int i = threadIdx.x; i < STAT_SLOT * C_out; i += 256) {
            const int tl = i / C_out, c = i % C_out;
            ds[tl * LD + c] = (t0 + tl < T) ? dy[((size_t)b * T + t0 + tl) * C_out + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MAXO; ++q) {
            const int o = threadIdx.x + q * 256;
            if (o < nout) {
                const int co = o % C_out, r = o / C_out;
                const int ci = r % C_in, k = r / C_in;
                float a = 0.f;
                for (int tl = 0; tl < STAT_SLOT; ++tl) a = fmaf(ds[tl * LD + co], xs[ci * TW + tl + k], a);
                acc[q] += a;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < MAXO; ++q) {
        const int o = threadIdx.x + q * 256;
        if (o < nout) {
            const int co = o % C_out, r = o / C_out;
            const int ci = r % C_in, k = r / C_in;
            atomicAdd(dw + ((size_t)co * C_in + ci) * KT + k, acc[q]);
        }
    }
}

// Head backward.  dF = c_out[b]*dpred (B,Co,T).  Produces: G = (W^T * dF) * silu'(a*h+s) (B,T,Ci) with GN partial sums,
// and accumulates dW, db (atomics into zeroed gradients).
template <int KT, int MAXCO>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dpred, const float* __restrict__ c_out,
                                                       const float* __restrict__ h, const float* __restrict__ gscale,
                                                       const float* __restrict__ gshift, const float* __restrict__ w,
                                                       float* __restrict__ G, float* __restrict__ gstats,
                                                       float* __restrict__ dw, float* __restrict__ db, int T, int C_in,
                                                       int C_out, int nslots) {
    extern __shared__ float shm[];
    constexpr int PAD = KT / 2;
    constexpr int TW = STAT_SLOT + KT - 1;
    const int LDZ = TW + 1;
    float* dfs = shm;                    // [C_out][TW]   dF with halo
    float* zs = dfs + MAXCO * TW;        // [C_in][LDZ]   activated input with halo (transposed)
    float* red = zs + C_in * LDZ;        // [nrow][C_in][2]
    const int slot = blockIdx.x % nslots, b = blockIdx.x / nslots;
    const int t0 = slot * STAT_SLOT;
    const float co_s = c_out ? c_out[b] : 1.0f;
    for (int i = threadIdx.x; i < C_out * TW; i += 256) {
        const int c = i / TW, j = i % TW;
        const int t = t0 - PAD + j;
        dfs[c * TW + j] = (t >= 0 && t < T) ? dpred[((size_t)b * C_out + c) * T + t] * co_s : 0.f;
    }
    const int nc4 = C_in >> 2;
    for (int i = threadIdx.x; i < TW * nc4; i += 256) {
        const int c4 = i % nc4, j = i / nc4;
        const int t = t0 - PAD + j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < T) {
            v = *reinterpret_cast<const float4*>(h + ((size_t)b * T + t) * C_in + 4 * c4);
            if (gscale) {
                const float4 a = *reinterpret_cast<const float4*>(gscale + (size_t)b * C_in + 4 * c4);
                const float4 s = *reinterpret_cast<const float4*>(gshift + (size_t)b * C_in + 4 * c4);
                v.x = silu_f(a.x * v.x + s.x); v.y = silu_f(a.y * v.y + s.y);
                v.z = silu_f(a.z * v.z + s.z); v.w = silu_f(a.w * v.w + s.w);
            }
        }
        zs[(4 * c4 + 0) * LDZ + j] = v.x; zs[(4 * c4 + 1) * LDZ + j] = v.y;
        zs[(4 * c4 + 2) * LDZ + j] = v.z; zs[(4 * c4 + 3) * LDZ + j] = v.w;
    }
    __syncthreads();
    // ---- weight / bias gradient: thread per (co, ci, k); the reduction runs over the slot's valid positions
    const int nout = C_out * C_in * KT;
    const int tvalid = min(STAT_SLOT, T - t0);
    for (int o = threadIdx.x; o < nout; o += 256) {
        const int k = o % KT, r = o / KT;
        const int ci = r % C_in, co = r / C_in;
        float a = 0.f;
        for (int tl = 0; tl < tvalid; ++tl) a = fmaf(dfs[co * TW + tl + PAD], zs[ci * LDZ + tl + k], a);
        atomicAdd(dw + o, a);
    }
    if (threadIdx.x < C_out) {
        float a = 0.f;
        for (int tl = 0; tl < tvalid; ++tl) a += dfs[threadIdx.x * TW + tl + PAD];
        atomicAdd(db + threadIdx.x, a);
    }
    // ---- data gradient: thread = (4 input channels, rows tr, tr+nrow, ...)
    const int nrow = 256 / nc4;
    const int c4 = threadIdx.x % nc4, tr = threadIdx.x / nc4;
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (tr < nrow) {
        float4 ga = make_float4(1, 1, 1, 1), gs = make_float4(0, 0, 0, 0);
        if (gscale) {
            ga = *reinterpret_cast<const float4*>(gscale + (size_t)b * C_in + 4 * c4);
            gs = *reinterpret_cast<const float4*>(gshift + (size_t)b * C_in + 4 * c4);
        }
        const float a4[4] = {ga.x, ga.y, ga.z, ga.w}, h4[4] = {gs.x, gs.y, gs.z, gs.w};
        for (int tl = tr; tl < tvalid; tl += nrow) {
            const size_t o = ((size_t)b * T + t0 + tl) * C_in + 4 * c4;
            const float4 hv4 = *reinterpret_cast<const float4*>(h + o);
            const float hv[4] = {hv4.x, hv4.y, hv4.z, hv4.w};
            float gq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ci = 4 * c4 + j;
                float a = 0.f;
                for (int co = 0; co < C_out; ++co)
#pragma unroll
                    for (int k = 0; k < KT; ++k)
                        a = fmaf(w[((size_t)co * C_in + ci) * KT + k], dfs[co * TW + tl + 2 * PAD - k], a);  // dF[t + pad - k]
                if (gscale) a *= dsilu_f(a4[j] * hv[j] + h4[j]);
                gq[j] = a;
                s1[j] += a; s2[j] += a * hv[j];
            }
            *reinterpret_cast<float4*>(G + o) = make_float4(gq[0], gq[1], gq[2], gq[3]);
        }
    }
    if (gstats) {
        __syncthreads();
        if (tr < nrow) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { red[(tr * C_in + 4 * c4 + j) * 2] = s1[j]; red[(tr * C_in + 4 * c4 + j) * 2 + 1] = s2[j]; }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C_in; c += 256) {
            float a1 = 0.f, a2 = 0.f;
            for (int r2 = 0; r2 < nrow; ++r2) { a1 += red[(r2 * C_in + c) * 2]; a2 += red[(r2 * C_in + c) * 2 + 1]; }
            float* st = gstats + (((size_t)b * nslots + slot) * C_in + c) * 2;
            st[0] = a1; st[1] = a2;
        }
    }
}
}  // namespace

// up to STEM_HEAD_WGS_MAX workgroups x <= 1024 partial sums.  Workgroups per launch (A/B switch TQDNE_STEM_HEAD_WGS), measured at B = 64
// (tools/bwd_micro.py, us for 256 / 512 / 1024 / 2048): stem weight gradient 118 / 74 / 62 / 86 (118 registers: four workgroups per CU
// cover each other's staging and load latencies), head backward 102 / 85 / 105 / 145 (157 registers: two per CU).
constexpr int STEM_HEAD_WGS_MAX = 2048;
static int stem_head_wgs(int dflt) {
    static const int forced = [] { const char* e = getenv("TQDNE_STEM_HEAD_WGS"); return e ? atoi(e) : 0; }();
    const int v = forced > 0 ? forced : dflt;
    return v > STEM_HEAD_WGS_MAX ? STEM_HEAD_WGS_MAX : v;
}
extern "C" size_t tq_stem_head_bwd_workspace(void) { return (size_t)STEM_HEAD_WGS_MAX * 1024 * sizeof(float); }

extern "C" int tq_stem_conv_bwd_weight(const float* dy, const float* x_nct, const float* in_scale, float* dw, int B, int C_in,
                                       int T, int C_out, int ktaps, hipStream_t stream) {
    return tq_stem_conv_bwd_weight_ws(dy, x_nct, in_scale, dw, B, C_in, T, C_out, ktaps, nullptr, 0, stream);
}

extern "C" int tq_stem_conv_bwd_weight_ws(const float* dy, const float* x_nct, const float* in_scale, float* dw, int B, int C_in,
                                          int T, int C_out, int ktaps, void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!dy || !x_nct || !dw) return TQ_ERR_ARG;
    if (B <= 0 || T <= 0 || C_in <= 0 || C_in > 16 || C_out <= 0 || (ktaps != 1 && ktaps != 3 && ktaps != 5)) return TQ_ERR_SHAPE;
    const int nslots = (T + STAT_SLOT - 1) / STAT_SLOT;
    const size_t sh = ((size_t)C_in * (STAT_SLOT + ktaps - 1) + (size_t)STAT_SLOT * (C_out + 1)) * sizeof(float);
    if (sh > 160 * 1024) return TQ_ERR_SHAPE;
    if (C_out * C_in * ktaps > 8 * 256) return TQ_ERR_SHAPE;
    static const bool old_form = [] { const char* e = getenv("TQDNE_STEM_HEAD_BWD"); return e && atoi(e) == 3; }();   // (A/B switch: 3 = round 3's kernels)
    if (!old_form && C_in <= 8 && C_out >= 32 && C_out <= 256 && 256 % C_out == 0 && ktaps == 5) {   // the streaming form (round 4)
        const int nunits2 = B * nslots;
        const int nwg2 = nunits2 < stem_head_wgs(1024) ? nunits2 : stem_head_wgs(1024);
        const int upw2 = (nunits2 + nwg2 - 1) / nwg2;
        const unsigned grid2 = (unsigned)((nunits2 + upw2 - 1) / upw2);
        const int nout2 = C_out * C_in * 5;
        float* part = (workspace && ws_bytes >= (size_t)grid2 * nout2 * sizeof(float)) ? reinterpret_cast<float*>(workspace) : nullptr;
        if (C_in <= 4) {
            const size_t sh2 = ((size_t)4 * (STAT_SLOT + 4) + (size_t)256 * C_in * 5) * sizeof(float);
            hipLaunchKernelGGL((stem_wgrad_stream_kernel<5, 4>), dim3(grid2), dim3(256), sh2, stream, dy, x_nct, in_scale, dw, part, C_in, T,
                               C_out, nslots, nunits2, upw2);
        } else {
            const size_t sh2 = ((size_t)8 * (STAT_SLOT + 4) + (size_t)256 * C_in * 5) * sizeof(float);
            hipLaunchKernelGGL((stem_wgrad_stream_kernel<5, 8>), dim3(grid2), dim3(256), sh2, stream, dy, x_nct, in_scale, dw, part, C_in, T,
                               C_out, nslots, nunits2, upw2);
        }
        TQ_CHECK_LAUNCH();
        if (part) {
            hipLaunchKernelGGL(partial_rows_sum_kernel, dim3((nout2 + 63) / 64), dim3(256), 0, stream, part, (int)grid2, nout2, nout2, dw);
            TQ_CHECK_LAUNCH();
        }
        return 0;
    }
    auto kern = stem_wgrad_kernel;
    if (sh > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    const int nunits = B * nslots;
    const int nwg = nunits < 512 ? nunits : 512;
    const int upw = (nunits + nwg - 1) / nwg;
    hipLaunchKernelGGL(kern, dim3((nunits + upw - 1) / upw), dim3(256), sh, stream, dy, x_nct, in_scale, dw, C_in, T, C_out, ktaps,
                       nslots, nunits, upw);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_head_conv_bwd(const float* dpred_nct, const float* c_out, const float* x, const float* gscale,
                                const float* gshift, const float* w, float* g_out, float* gstats, float* dw, float* db, int B,
                                int T, int C_in, int C_out, int ktaps, hipStream_t stream) {
    return tq_head_conv_bwd_ws(dpred_nct, c_out, x, gscale, gshift, w, g_out, gstats, dw, db, B, T, C_in, C_out, ktaps, nullptr, 0, stream);
}

extern "C" int tq_head_conv_bwd_ws(const float* dpred_nct, const float* c_out, const float* x, const float* gscale,
                                   const float* gshift, const float* w, float* g_out, float* gstats, float* dw, float* db, int B,
                                   int T, int C_in, int C_out, int ktaps, void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!dpred_nct || !x || !w || !g_out || !dw || !db) return TQ_ERR_ARG;
    if ((gscale == nullptr) != (gshift == nullptr)) return TQ_ERR_ARG;
    if (B <= 0 || T <= 0 || C_in < 8 || C_in % 8 || 256 % (C_in / 4) || C_out < 1 || C_out > 16) return TQ_ERR_SHAPE;
    const int maxco = C_out <= 4 ? 4 : 16;
    const int nslots = (T + STAT_SLOT - 1) / STAT_SLOT;
    static const bool old_form = [] { const char* e = getenv("TQDNE_STEM_HEAD_BWD"); return e && atoi(e) == 3; }();   // (A/B switch: 3 = round 3's kernels)
    if (!old_form && C_out <= 8 && ktaps == 5 && C_in >= 32 && C_in <= 256 && 256 % C_in == 0) {   // the streaming form (round 4; 5 ... 8 channels: round 5)
        const int mco = C_out <= 4 ? 4 : 8;
        const int nunits = B * nslots;
        const int nwg = nunits < stem_head_wgs(512) ? nunits : stem_head_wgs(512);
        const int upw = (nunits + nwg - 1) / nwg;
        const size_t sh2 = ((size_t)mco * (STAT_SLOT + 4) + (size_t)256 * mco * 5) * sizeof(float);   // (<= 45 KB)
        const unsigned grid2 = (unsigned)((nunits + upw - 1) / upw);
        const int nout2 = C_out * C_in * 5 + mco;   // dw, then db padded to the kernel's MAXCO
        float* part = (workspace && ws_bytes >= (size_t)grid2 * nout2 * sizeof(float)) ? reinterpret_cast<float*>(workspace) : nullptr;
        if (mco == 4)
            hipLaunchKernelGGL((head_bwd_stream_kernel<5, 4>), dim3(grid2), dim3(256), sh2, stream, dpred_nct, c_out, x, gscale, gshift, w, g_out,
                               gstats, dw, db, part, T, C_in, C_out, nslots, nunits, upw);
        else
            hipLaunchKernelGGL((head_bwd_stream_kernel<5, 8>), dim3(grid2), dim3(256), sh2, stream, dpred_nct, c_out, x, gscale, gshift, w, g_out,
                               gstats, dw, db, part, T, C_in, C_out, nslots, nunits, upw);
        TQ_CHECK_LAUNCH();
        if (part) {
            // (dw's rows first: C_out C_in 5 floats; the bias sums sit behind them in every row)
            hipLaunchKernelGGL(partial_rows_sum_kernel, dim3((nout2 - mco + 63) / 64), dim3(256), 0, stream, part, (int)grid2, nout2, nout2 - mco, dw);
            TQ_CHECK_LAUNCH();
            hipLaunchKernelGGL(partial_rows_sum_kernel, dim3(1), dim3(256), 0, stream, part + (nout2 - mco), (int)grid2, nout2, C_out, db);
            TQ_CHECK_LAUNCH();
        }
        return 0;
    }
    const int nrow = 256 / (C_in / 4);
    const size_t sh = ((size_t)maxco * (STAT_SLOT + ktaps - 1) + (size_t)C_in * (STAT_SLOT + ktaps) + (size_t)nrow * C_in * 2) * sizeof(float);
    if (sh > 160 * 1024) return TQ_ERR_SHAPE;
#define TQ_HB(K)                                                                                             \
    {                                                                                                        \
        auto kern = (maxco == 4) ? head_bwd_kernel<K, 4> : head_bwd_kernel<K, 16>;                                                                    \
        if (sh > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
        hipLaunchKernelGGL(kern, dim3(B * nslots), dim3(256), sh, stream, dpred_nct, c_out, x, gscale, gshift, w, g_out, gstats, dw, db, T, C_in, C_out, nslots); \
    }
    if (ktaps == 5) TQ_HB(5)
    else if (ktaps == 3) TQ_HB(3)
    else if (ktaps == 1) TQ_HB(1)
    else return TQ_ERR_SHAPE;
#undef TQ_HB
    TQ_CHECK_LAUNCH();
    return 0;
}


// =================================================================================================
// Small fp32 GEMMs of the embedding-MLP backward (unet.py:91-97, 210-227, 383-388): (B x 4mc)-sized matrices, a few hundred
// MFLOP in all, exact fp32 FMA on the vector units.  One launch runs a LIST of independent GEMMs (the weight gradient, the bias
// gradient as a product with a row of ones, and the data gradient of one MLP level), so the whole backward of the embedding path
// is three dependent launches instead of ~25 library calls.
//   C (M x N) = A (M x K) * f(B) (K x N)  [* dsilu(U)],  A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]
// =================================================================================================
namespace {
constexpr int GJ_T = 32;  // tile edge

__global__ __launch_bounds__(256) void gemm_jobs_kernel(const TqGemmJob* __restrict__ jobs, int njobs) {
    __shared__ float As[GJ_T][GJ_T + 1];
    __shared__ float Bs[GJ_T][GJ_T + 1];
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].tile_begin) ++j;
    const TqGemmJob jb = jobs[j];
    const int tile = blockIdx.x - jb.tile_begin;
    const int ntn = (jb.N + GJ_T - 1) / GJ_T;
    const int m0 = (tile / ntn) * GJ_T, n0 = (tile % ntn) * GJ_T;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // thread -> outputs (m0 + ty + 16 i, n0 + tx + 16 jx)
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int k0 = 0; k0 < jb.K; k0 += GJ_T) {
        for (int i = threadIdx.x; i < GJ_T * GJ_T; i += 256) {
            const int r = i / GJ_T, c = i % GJ_T;
            // A tile: rows m, cols k.  Read along the operand's contiguous direction
            {
                const int mm = jb.sak == 1 ? r : c, kk = jb.sak == 1 ? c : r;
                const int m = m0 + mm, k = k0 + kk;
                As[mm][kk] = (m < jb.M && k < jb.K) ? jb.A[(size_t)m * jb.sam + (size_t)k * jb.sak] : 0.f;
            }
            {
                const int kk = jb.sbn == 1 ? r : c, nn = jb.sbn == 1 ? c : r;
                const int k = k0 + kk, n = n0 + nn;
                float v = (k < jb.K && n < jb.N) ? jb.B[(size_t)k * jb.sbk + (size_t)n * jb.sbn] : 0.f;
                if (jb.pre_b) v = silu_f(v);
                Bs[kk][nn] = v;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < GJ_T; ++k) {
            const float a0 = As[ty][k], a1 = As[ty + 16][k];
            const float b0 = Bs[k][tx], b1 = Bs[k][tx + 16];
            acc[0][0] = fmaf(a0, b0, acc[0][0]); acc[0][1] = fmaf(a0, b1, acc[0][1]);
            acc[1][0] = fmaf(a1, b0, acc[1][0]); acc[1][1] = fmaf(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jx = 0; jx < 2; ++jx) {
            const int m = m0 + ty + 16 * i, n = n0 + tx + 16 * jx;
            if (m < jb.M && n < jb.N) {
                float v = acc[i][jx];
                if (jb.U) v *= dsilu_f(jb.U[(size_t)m * jb.ldu + n]);
                jb.C[(size_t)m * jb.ldc + n] = v;
            }
        }
}

__global__ void fourier_features_kernel(const float* __restrict__ t, const float* __restrict__ W, float* __restrict__ out, int B,
                                        int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i % half;
    float a = t[b] * W[j];      // blocks.py:24: x[:, None] * W[None, :] * 2 * pi, evaluated left to right in fp32
    a = a * 2.0f;
    a = a * 3.14159265358979323846f;
    out[(size_t)b * 2 * half + j] = sinf(a);
    out[(size_t)b * 2 * half + half + j] = cosf(a);
}
}  // namespace

extern "C" int tq_gemm_tiles(int M, int N) { return ((M + GJ_T - 1) / GJ_T) * ((N + GJ_T - 1) / GJ_T); }

extern "C" int tq_gemm_f32_jobs(const TqGemmJob* jobs_device, int njobs, int total_tiles, hipStream_t stream) {
    if (!jobs_device || njobs <= 0 || total_tiles <= 0) return TQ_ERR_ARG;
    hipLaunchKernelGGL(gemm_jobs_kernel, dim3((unsigned)total_tiles), dim3(256), 0, stream, jobs_device, njobs);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" int tq_fourier_features(const float* t, const float* W, float* out, int B, int half, hipStream_t stream) {
    if (!t || !W || !out || B <= 0 || half <= 0) return TQ_ERR_ARG;
    hipLaunchKernelGGL(fourier_features_kernel, dim3((unsigned)((B * half + 255) / 256)), dim3(256), 0, stream, t, W, out, B, half);
    TQ_CHECK_LAUNCH();
    return 0;
}
