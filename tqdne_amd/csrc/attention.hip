// 1-D self-attention core of the tqdne UNet (QKVAttention, tqdne/blocks.py:156-190) for gfx950.
//
//   qkv (B, T, 3*H*D) channels-last, channel order [q heads | k heads | v heads]
//   out[b, t, h*D + c] = sum_s softmax_s( (q*D^-1/4) . (k*D^-1/4) )[t, s] * v[s, c]
//
// Flash-style: the (T x T) score matrix of the reference (4 MB per sample per block at T=512) never
// exists in HBM.  One workgroup = 64 queries of one (b, head), 4 waves x 16 queries; keys/values are
// streamed in tiles of 64 through LDS; QK^T and PV run on v_mfma_f32_16x16x32_bf16 with the same
// bf16 hi/lo 3-product split as the convolutions (scores feed an exponential, so single bf16 is not
// accurate enough for the 1e-3 parity target); the online softmax is fp32, as in the reference.
#include "common.hpp"
#include "../../include/tqdne_hip.h"

using namespace tq;

namespace {

constexpr int QT = 64;   // queries per workgroup
constexpr int KTILE = 64;  // keys per tile

template <int D>
__global__ __launch_bounds__(256, 2) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           float* __restrict__ lse, int T, int H, float scale) {
    constexpr int KS = D / 32;       // k-steps over the head dimension
    constexpr int CB = D / 16;       // output column blocks
    constexpr int KROW = D * 2 + 16;   // bytes per key row of the K image (padded)
    constexpr int VROW = KTILE * 2 + 16;  // bytes per channel row of the V^T image
    constexpr int PROW = KTILE * 2 + 16;  // bytes per query row of the P image
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* k_hi = lds;
    unsigned char* k_lo = k_hi + KTILE * KROW;
    unsigned char* v_hi = k_lo + KTILE * KROW;
    unsigned char* v_lo = v_hi + D * VROW;
    unsigned char* p_base = v_lo + D * VROW;  // [4 waves][2 planes][16][PROW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqt = (T + QT - 1) / QT;
    int bid = xcd_group_id(blockIdx.x, nqt, gridDim.x / nqt);  // the tiles of one (b, h) share an XCD's L2
    const int qt = bid % nqt; bid /= nqt;
    const int h = bid % H;
    const int b = bid / H;
    const int C3 = 3 * H * D;
    const float* base = qkv + (size_t)b * T * C3;
    const int q0 = qt * QT + wave * 16;
    unsigned char* p_hi = p_base + wave * 2 * 16 * PROW;
    unsigned char* p_lo = p_hi + 16 * PROW;

    // ---- Q fragments (A operand: row = query l&15, k = d) kept in registers for the whole kernel
    Frag qh[KS], ql[KS];
    {
        const int q = q0 + (lane & 15);
        const bool ok = q < T;
        const float* qp = base + (size_t)(ok ? q : 0) * C3 + h * D + 8 * (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float4 a = make_float4(0, 0, 0, 0), c = a;
            if (ok) {
                a = *reinterpret_cast<const float4*>(qp + ks * 32);
                c = *reinterpret_cast<const float4*>(qp + ks * 32 + 4);
            }
            const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 hh, ll;
                split_bf16(v[j] * scale, hh, ll);
                qh[ks].v[j] = hh; ql[ks].v[j] = ll;
            }
        }
    }

    f32x4 o[CB];
#pragma unroll
    for (int i = 0; i < CB; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[4], l_run[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }

    const int nkt = (T + KTILE - 1) / KTILE;
    for (int kt = 0; kt < nkt; ++kt) {
        const int s0 = kt * KTILE;
        __syncthreads();  // previous tile fully consumed
        // ---- stage K tile: thread -> (key = i / (D/4), 4 channels)
        for (int i = tid; i < KTILE * (D / 4); i += 256) {
            const int key = i / (D / 4), c4 = i % (D / 4);
            float4 v = make_float4(0, 0, 0, 0);
            if (s0 + key < T) v = *reinterpret_cast<const float4*>(base + (size_t)(s0 + key) * C3 + (H + h) * D + 4 * c4);
            const float u[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
            bf16x4 hv, lv;
#pragma unroll
            for (int j = 0; j < 4; ++j) { __bf16 hh, ll; split_bf16(u[j], hh, ll); hv[j] = hh; lv[j] = ll; }
            *reinterpret_cast<bf16x4*>(k_hi + key * KROW + c4 * 8) = hv;
            *reinterpret_cast<bf16x4*>(k_lo + key * KROW + c4 * 8) = lv;
        }
        // ---- stage V^T tile: thread -> (4 channels c4, 4 keys kg), transposed in registers
        for (int i = tid; i < (KTILE / 4) * (D / 4); i += 256) {
            const int c4 = i % (D / 4), kg = i / (D / 4);
            float4 v[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int key = s0 + 4 * kg + kk;
                v[kk] = make_float4(0, 0, 0, 0);
                if (key < T) v[kk] = *reinterpret_cast<const float4*>(base + (size_t)key * C3 + (2 * H + h) * D + 4 * c4);
            }
            const float cols[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x}, {v[0].y, v[1].y, v[2].y, v[3].y},
                                      {v[0].z, v[1].z, v[2].z, v[3].z}, {v[0].w, v[1].w, v[2].w, v[3].w}};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4 hv, lv;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { __bf16 hh, ll; split_bf16(cols[j][kk], hh, ll); hv[kk] = hh; lv[kk] = ll; }
                *reinterpret_cast<bf16x4*>(v_hi + (4 * c4 + j) * VROW + kg * 8) = hv;
                *reinterpret_cast<bf16x4*>(v_lo + (4 * c4 + j) * VROW + kg * 8) = lv;
            }
        }
        __syncthreads();

        // ---- S = Q K^T  (16 queries x 64 keys per wave)
        f32x4 s[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            s[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int key = cb * 16 + (lane & 15);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag bh, bl;
                const int off = key * KROW + (ks * 4 + (lane >> 4)) * 16;
                bh.u = *reinterpret_cast<const uint4*>(k_hi + off);
                bl.u = *reinterpret_cast<const uint4*>(k_lo + off);
                s[cb] = mfma_x3(qh[ks].v, ql[ks].v, bh.v, bl.v, s[cb]);
            }
        }
        // ---- online softmax; lane holds rows 4*(lane>>4)+r, column cb*16 + (lane&15)
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mx = -INFINITY;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const bool valid = (s0 + cb * 16 + (lane & 15)) < T;
                if (!valid) s[cb][r] = -INFINITY;
                mx = fmaxf(mx, s[cb][r]);
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            const float m_new = fmaxf(m_run[r], mx);
            alpha[r] = (m_run[r] == -INFINITY) ? 0.f : __expf(m_run[r] - m_new);
            float rs = 0.f;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const float pv = (s[cb][r] == -INFINITY) ? 0.f : __expf(s[cb][r] - m_new);
                s[cb][r] = pv;
                rs += pv;
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) rs += __shfl_xor(rs, off);
            l_run[r] = l_run[r] * alpha[r] + rs;
            m_run[r] = m_new;
        }
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[i][r] *= alpha[r];
        // ---- P -> LDS (per-wave image [query][key], bf16 hi/lo)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                __bf16 hh, ll;
                split_bf16(s[cb][r], hh, ll);
                const int off = (4 * (lane >> 4) + r) * PROW + (cb * 16 + (lane & 15)) * 2;
                *reinterpret_cast<__bf16*>(p_hi + off) = hh;
                *reinterpret_cast<__bf16*>(p_lo + off) = ll;
            }
        __syncthreads();
        // ---- O += P V   (A = P: row = query l&15, k = key; B = V^T rows = channel)
#pragma unroll
        for (int ks = 0; ks < KTILE / 32; ++ks) {
            Frag ph, pl;
            const int poff = (lane & 15) * PROW + (ks * 4 + (lane >> 4)) * 16;
            ph.u = *reinterpret_cast<const uint4*>(p_hi + poff);
            pl.u = *reinterpret_cast<const uint4*>(p_lo + poff);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                Frag vh, vl;
                const int voff = (cb * 16 + (lane & 15)) * VROW + (ks * 4 + (lane >> 4)) * 16;
                vh.u = *reinterpret_cast<const uint4*>(v_hi + voff);
                vl.u = *reinterpret_cast<const uint4*>(v_lo + voff);
                o[cb] = mfma_x3(ph.v, pl.v, vh.v, vl.v, o[cb]);
            }
        }
    }
    // ---- normalise and store
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + 4 * (lane >> 4) + r;
        if (q < T) {
            const float inv = 1.0f / l_run[r];
            if (lse && (lane & 15) == 0) lse[((size_t)b * H + h) * T + q] = m_run[r] + __logf(l_run[r]);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                out[((size_t)b * T + q) * (H * D) + h * D + cb * 16 + (lane & 15)] = o[cb][r] * inv;
        }
    }
}

template <int D>
int launch_attn(const float* qkv, float* out, float* lse, int B, int T, int H, hipStream_t stream) {
    constexpr int KROW = D * 2 + 16, VROW = KTILE * 2 + 16, PROW = KTILE * 2 + 16;
    const size_t sh = 2 * KTILE * KROW + 2 * D * VROW + 4 * 2 * 16 * PROW;
    if (sh > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    const int nqt = (T + QT - 1) / QT;
    const float scale = (float)(1.0 / sqrt(sqrt((double)D)));  // blocks.py:173 (python double, then fp32)
    hipLaunchKernelGGL(attention_kernel<D>, dim3(B * H * nqt), dim3(256), sh, stream, qkv, out, lse, T, H, scale);
    TQ_CHECK_LAUNCH();
    return 0;
}
}  // namespace

// -------------------------------------------------------------------------------------------------
// Forward, second generation (used when the caller supplies a workspace):
//   * K (scaled) and V are split into bf16 hi/lo ONCE by a prep pass ([b][h][plane][Tp][D]); the main kernel's staging
//     is then a pure 16-byte copy (the first-generation kernel re-did the split in each of the T/64 query tiles);
//   * 128 queries per workgroup (32 per wave), so each staged key tile feeds twice the MFMA work;
//   * scores are computed transposed, S^T = K Q^T: the accumulator lane then holds 4 consecutive keys of ONE query,
//     which is exactly an operand fragment of the P V product once the k-slots of a 32-key step are permuted as
//     key(g, j) = 16*(j>>2) + 4*g + (j&3).  V is read with ds_read_b64_tr_b16 from its row-major image with the same
//     permutation, so P never goes through LDS and V is never transposed;
//   * round 3: the output is accumulated transposed as well (O^T = V^T P^T, same fragments with the MFMA operands swapped), so a
//     lane's accumulators all belong to ONE query: the softmax bookkeeping is per lane (no shuffles in the loop), the lane stores
//     16 contiguous bytes per channel block, and the row sum's cross-lane reduction happens once, after the loop;
//   * round 3: lazy reference level instead of a running maximum (see the comment at m_ref): the score accumulators start at
//     -m_ref, p = exp2(accumulator) with no subtraction, and the O rescale exists only on a rarely taken wave-uniform path;
//   * round 3: the staging loads are asm statements (att_load): hipcc sank plain loads to the end of the iteration.
//     Same box, B = 64, T = 512, 4 x 64, kernel only (tools/att_ab.sh): 62 -> 57 us.  Ablation builds (-DTQ_ATT_ABL_*) of this
//     kernel: no MFMAs 45 us, MFMAs only 42.5 us (pipe floor 24.6 us at 2 GHz), no staging 51 us, no LDS reads 55 us: the phases
//     of a wave serialise (SQ counters: VALU active 36 %, MFMA busy 34 %, 26 % in s_waitcnt), and 12 us are the two rounds'
//     prologues and epilogues.
// LDS rows are padded by 32 bytes: conflict-free for the b128 K reads and the transposed V reads (all D).
// -------------------------------------------------------------------------------------------------
#ifndef ATT_QB
#define ATT_QB 2  // 16-query blocks per wave of the forward kernel
#endif
namespace {
typedef short s16x4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 tr_read_f(const unsigned char* p) {
    s16x4f v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4f*)(p));
    union { s16x4f s; uint2 u; } c;
    c.s = v;
    return c.u;
}

// kv[b][h][plane][t][d] (bf16), planes: 0 K hi, 1 K lo, 2 V hi, 3 V lo; rows t >= T are zero
__global__ void attn_prep_kernel(const float* __restrict__ qkv, unsigned char* __restrict__ kv, int T, int Tp, int H, int D,
                                 float scale, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d4n = D >> 2;
    const int c4 = (int)(i % d4n);
    size_t r = i / d4n;
    const int t = (int)(r % Tp); r /= Tp;
    const int h = (int)(r % H);
    const size_t b = r / H;
    float4 kx = make_float4(0, 0, 0, 0), vx = kx;
    if (t < T) {
        const float* row = qkv + ((size_t)b * T + t) * (3 * H * D);
        kx = *reinterpret_cast<const float4*>(row + (H + h) * D + 4 * c4);
        vx = *reinterpret_cast<const float4*>(row + (2 * H + h) * D + 4 * c4);
    }
    const float ku[4] = {kx.x * scale, kx.y * scale, kx.z * scale, kx.w * scale};
    const float vu[4] = {vx.x, vx.y, vx.z, vx.w};
    bf16x4 kh, kl, vh, vl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __bf16 a, c;
        split_bf16(ku[j], a, c); kh[j] = a; kl[j] = c;
        split_bf16(vu[j], a, c); vh[j] = a; vl[j] = c;
    }
    const size_t plane = (size_t)Tp * D * 2;  // bytes
    unsigned char* base = kv + (((size_t)b * H + h) * 4) * plane + ((size_t)t * D + 4 * c4) * 2;
    *reinterpret_cast<bf16x4*>(base) = kh;
    *reinterpret_cast<bf16x4*>(base + plane) = kl;
    *reinterpret_cast<bf16x4*>(base + 2 * plane) = vh;
    *reinterpret_cast<bf16x4*>(base + 3 * plane) = vl;
}

// staging vector i = tid + it*256 -> (plane, row, 16-byte column) of one 64-key tile of the pre-split K/V planes.
// The loads are asm statements: hipcc sinks ordinary loads from the top of the key-tile iteration to their only use at its end
// (`tools/pmc_attention.sh`: the waves then spent half of their life in s_waitcnt vmcnt), and neither sched_barrier nor the source
// order stops that IR-level move.  The price: hipcc does not count them, so `att_wait` must stand before the first use.
template <int D, int NIT>
__device__ __forceinline__ void att_load(uint4 (&stg)[NIT], const unsigned char* kvb, size_t gplane, int kt, int tid) {
    constexpr int V16 = D / 8;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * 256;
        const int pl = i / (64 * V16), rem = i % (64 * V16);
        const int row = rem / V16, c16 = rem % V16;
        const unsigned char* src = kvb + pl * gplane + ((size_t)(kt * 64 + row) * D) * 2 + c16 * 16;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stg[it]) : "v"(src));
    }
}
__device__ __forceinline__ void att_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int D, int NIT>
__device__ __forceinline__ void att_write(const uint4 (&stg)[NIT], unsigned char* buf, int, int tid) {
    constexpr int V16 = D / 8;
    constexpr int ROWB = 2 * D + 32;
    constexpr int PLANE = 64 * ROWB;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * 256;
        const int pl = i / (64 * V16), rem = i % (64 * V16);
        const int row = rem / V16, c16 = rem % V16;
        *reinterpret_cast<uint4*>(buf + pl * PLANE + row * ROWB + c16 * 16) = stg[it];
    }
}

#ifdef TQ_STAMP
// diagnostic build only: per-workgroup phase stamps of the forward kernel (s_memrealtime, 10 ns ticks; [6], [7] = s_memtime)
__device__ unsigned long long tq_att_timeline[4096 * 8];
extern "C" int tq_debug_read_att_timeline(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tq_att_timeline), sizeof(unsigned long long) * 8 * (n < 4096 ? n : 4096));
}
#define ATT_T(i) if (tid == 0 && blockIdx.x < 4096) tq_att_timeline[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define ATT_T(i)
#endif

template <int D>
__global__ __launch_bounds__(256, 2) void attention_fwd2_kernel(const float* __restrict__ qkv, const unsigned char* __restrict__ kv,
                                                                float* __restrict__ out, float* __restrict__ lse, int T, int Tp,
                                                                int H, float scale) {
    constexpr int KS = D / 32, CB = D / 16, QB = ATT_QB;
    constexpr int ROWB = 2 * D + 32;
    constexpr int PLANE = 64 * ROWB;
    constexpr int BUFB = 4 * PLANE;          // one LDS buffer: K hi, K lo, V hi, V lo
    constexpr int V16 = D / 8;               // 16-byte vectors per row
    constexpr int NIT = (4 * 64 * V16) / 256;  // staging vectors per thread per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef TQ_STAMP
    if (tid == 0 && blockIdx.x < 4096) tq_att_timeline[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memtime();
#endif
    ATT_T(0)
#ifdef TQ_ATT_SKEW
    if ((blockIdx.x >> 8) & 1) { for (int i = 0; i < TQ_ATT_SKEW; ++i) __builtin_amdgcn_s_sleep(16); }
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int nqt = (T + 64 * ATT_QB - 1) / (64 * ATT_QB);
    int bid = xcd_group_id(blockIdx.x, nqt, gridDim.x / nqt);  // the tiles of one (b, h) share an XCD's L2
    const int qt = bid % nqt; bid /= nqt;
    const int h = bid % H;
    const int b = bid / H;
    const int C3 = 3 * H * D;
    const int q0w = qt * (64 * ATT_QB) + wave * (16 * ATT_QB);

    const float qscale = scale * 1.44269504088896341f;  // scores in log2 units
    // Q as the B operand of S^T = K Q^T: lane (col = query li, k = 8*g + j)
    Frag qh[QB][KS], ql[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q0w + qb * 16 + li;
        const bool ok = q < T;
        const float* qp = qkv + ((size_t)b * T + (ok ? q : 0)) * C3 + h * D + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float4 a = make_float4(0, 0, 0, 0), c = a;
            if (ok) { a = *reinterpret_cast<const float4*>(qp + ks * 32); c = *reinterpret_cast<const float4*>(qp + ks * 32 + 4); }
            const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) { __bf16 hh, ll; split_bf16(v[j] * qscale, hh, ll); qh[qb][ks].v[j] = hh; ql[qb][ks].v[j] = ll; }
        }
    }
#ifdef TQ_STAMP
    if (__builtin_amdgcn_readfirstlane(qh[0][0].u.x ^ ql[1][1].u.y) == 0x12345) return;  // (forces the Q loads to land before the stamp)
#endif
    ATT_T(1)
    // O is accumulated transposed (O^T = V^T P^T: rows = channels 4g + r of a block, column = query li), so everything per query
    // -- reference level, row sum, normalisation -- is per LANE and needs no shuffle.
    f32x4 o[QB][CB];
    // Lazy reference level: scores leave the MFMAs already relative to m_ref (the accumulators START at -m_ref), p = exp2 of that,
    // and m_ref only moves when a tile's maximum exceeds it by more than REF_TH (wave-uniform slow path; always on the first
    // tile).  After tile 0, m_ref is some earlier tile's true row maximum, so the running maximum lies in [m_ref, m_ref + REF_TH]:
    // p <= 2^REF_TH and the row sum >= 1 -- no overflow, no underflow.  The four lanes (li, li + 16 g) of a query always move
    // m_ref together, so their partial row sums share one reference and are added once, after the loop.
    constexpr float REF_TH = 6.0f;
    float m_ref[QB], l_part[QB];
    f32x4 mneg[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_ref[qb] = 0.f; l_part[qb] = 0.f;
        mneg[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) o[qb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const size_t gplane = (size_t)Tp * D * 2;
    const unsigned char* kvb = kv + (((size_t)b * H + h) * 4) * gplane;
    const int nkt = (T + 63) / 64;
    {
        uint4 stg[NIT];
        att_load<D, NIT>(stg, kvb, gplane, 0, tid);
        att_wait();
        att_write<D, NIT>(stg, lds, 0, tid);
    }
    __syncthreads();
    ATT_T(2)
    for (int kt = 0; kt < nkt; ++kt) {
        const int s0 = kt * 64;
        const bool more = (kt + 1) < nkt;
        const unsigned char* k_hi = lds + BUFB * (kt & 1);
        const unsigned char* k_lo = k_hi + PLANE;
        const unsigned char* v_hi = k_hi + 2 * PLANE;
        const unsigned char* v_lo = k_hi + 3 * PLANE;
        uint4 stg[NIT];
#ifndef TQ_ATT_ABL_NOSTAGE
        att_load<D, NIT>(stg, kvb, gplane, more ? kt + 1 : kt, tid);  // in flight under this tile's MFMAs, written to the other buffer afterwards
#endif
        // ---- S^T tiles: st[kb][qb], lane holds (score - m_ref) of keys kb*16 + 4g + r, query qb*16 + li
        f32x4 st[4][QB];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) st[kb][qb] = mneg[qb];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag ah, al;
                const int off = (kb * 16 + li) * ROWB + (ks * 4 + g) * 16;
#ifdef TQ_ATT_ABL_NOLDS
                ah.u = make_uint4(off, kt, off ^ 0x3f803f80, 0x3f803f80); al.u = make_uint4(kt, off, 0x3c003c00, off);
#else
                ah.u = *reinterpret_cast<const uint4*>(k_hi + off);
                al.u = *reinterpret_cast<const uint4*>(k_lo + off);
#endif
#pragma unroll
#ifdef TQ_ATT_ABL_NOS
                for (int qb = 0; qb < QB; ++qb) st[kb][qb] += f32x4{ah.v[0], al.v[1], ah.v[2], al.v[3]} * (float)qh[qb][ks].v[0];
#else
                for (int qb = 0; qb < QB; ++qb) st[kb][qb] = mfma_x3(ah.v, al.v, qh[qb][ks].v, ql[qb][ks].v, st[kb][qb]);
#endif
            }
        }
        if (s0 + 64 > T) {  // ragged last tile: keys >= T get -inf, i.e. p = 0
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (s0 + kb * 16 + 4 * g + r >= T) st[kb][qb][r] = -INFINITY;
        }
        // ---- lane maxima; does any query of the wave need a new reference?
        float mx[QB];
        bool need = kt == 0;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            mx[qb] = st[0][qb][0];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx[qb] = fmaxf(mx[qb], st[kb][qb][r]);
            need = need || (mx[qb] > REF_TH);
        }
        if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float full = fmaxf(mx[qb], __shfl_xor(mx[qb], 16));
                full = fmaxf(full, __shfl_xor(full, 32));  // finite: tile 0 has key 0, later tiles only raise
                const float delta = kt == 0 ? full : fmaxf(full, 0.f);
                const float alpha = kt == 0 ? 0.f : __builtin_amdgcn_exp2f(-delta);  // (tile 0: O and the row sum are still zero)
                m_ref[qb] += delta;
                mneg[qb] = f32x4{-m_ref[qb], -m_ref[qb], -m_ref[qb], -m_ref[qb]};
                l_part[qb] *= alpha;
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) o[qb][cb] *= alpha;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) st[kb][qb] -= delta;
            }
        }
        // ---- p = exp2(score - m_ref), lane-partial row sums
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#ifdef TQ_ATT_ABL_NOEXP
                    const float pv = st[kb][qb][r] * 0.001f;
#else
                    const float pv = __builtin_amdgcn_exp2f(st[kb][qb][r]);
#endif
                    st[kb][qb][r] = pv;
                    rs += pv;
                }
            l_part[qb] += rs;
        }
        // ---- O^T += V^T P^T, k-slot (g, j) of a 32-key step <-> key 16*(j>>2) + 4g + (j&3)
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            Frag ph[QB], pl[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#ifdef TQ_ATT_ABL_NOSPLIT
                    if ((j & 1) == 0) { ph[qb].u = *reinterpret_cast<const uint4*>(&st[2 * ks2][qb]); pl[qb].u = *reinterpret_cast<const uint4*>(&st[2 * ks2 + 1][qb]); }
#else
                    __bf16 hh, ll;
                    split_bf16(st[2 * ks2 + (j >> 2)][qb][j & 3], hh, ll);
                    ph[qb].v[j] = hh; pl[qb].v[j] = ll;
#endif
                }
            const int vrow = ks2 * 32 + 4 * g + ((lane >> 2) & 3);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                Frag bh, bl;
                const int off = vrow * ROWB + (cb * 16 + 4 * (lane & 3)) * 2;
#ifdef TQ_ATT_ABL_NOLDS
                bh.u = make_uint4(off, kt, off ^ 0x3f803f80, 0x3f803f80); bl.u = make_uint4(kt, off, 0x3c003c00, off);
#else
                bh.h[0] = tr_read_f(v_hi + off); bh.h[1] = tr_read_f(v_hi + off + 16 * ROWB);
                bl.h[0] = tr_read_f(v_lo + off); bl.h[1] = tr_read_f(v_lo + off + 16 * ROWB);
#endif
#ifdef TQ_ATT_ABL_NOPV
                for (int qb = 0; qb < QB; ++qb) o[qb][cb] += f32x4{bh.v[0], bl.v[1], bh.v[2], bl.v[3]} * (float)ph[qb].v[0] + f32x4{pl[qb].v[0], pl[qb].v[1], ph[qb].v[2], ph[qb].v[3]};
#else
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) o[qb][cb] = mfma_x3(bh.v, bl.v, ph[qb].v, pl[qb].v, o[qb][cb]);
#endif
            }
        }
#ifndef TQ_ATT_ABL_NOSTAGE
        att_wait();
        att_write<D, NIT>(stg, lds + BUFB * ((kt + 1) & 1), 0, tid);  // (the last iteration re-stages its own tile: harmless)
#endif
#ifndef TQ_ATT_ABL_NOBAR
        __syncthreads();
#endif
    }
    ATT_T(3)
    ATT_T(4)
    // ---- normalise, store: the lane holds channels cb*16 + 4g + r of query qb*16 + li
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float l = l_part[qb] + __shfl_xor(l_part[qb], 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int q = q0w + qb * 16 + li;
        if (q < T) {
            if (lse && g == 0) lse[((size_t)b * H + h) * T + q] = m_ref[qb] * 0.693147180559945309f + __logf(l);
            float* op = out + ((size_t)b * T + q) * (H * D) + h * D + 4 * g;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) *reinterpret_cast<f32x4*>(op + cb * 16) = o[qb][cb] * inv;
        }
    }
    ATT_T(5)
#ifdef TQ_STAMP
    if (tid == 0 && blockIdx.x < 4096) tq_att_timeline[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime();
#endif
}

template <int D>
int launch_attn2(const float* qkv, float* out, float* lse, void* ws, int B, int T, int H, hipStream_t stream, bool presplit = false) {
    const int Tp = (T + 63) / 64 * 64;
    const float scale = (float)(1.0 / sqrt(sqrt((double)D)));
    if (!presplit) {
        const size_t n = (size_t)B * H * Tp * (D / 4);
        hipLaunchKernelGGL(attn_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, qkv,
                           reinterpret_cast<unsigned char*>(ws), T, Tp, H, D, scale, n);
        TQ_CHECK_LAUNCH();
    }
    constexpr int ROWB = 2 * D + 32;
    const size_t sh = 2 * 4 * 64 * ROWB;
    if (sh > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_fwd2_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    const int nqt = (T + 64 * ATT_QB - 1) / (64 * ATT_QB);
    hipLaunchKernelGGL(attention_fwd2_kernel<D>, dim3(B * H * nqt), dim3(256), sh, stream, qkv,
                       reinterpret_cast<const unsigned char*>(ws), out, lse, T, Tp, H, scale);
    TQ_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" size_t tq_attention_workspace_bytes(int B, int T, int H, int D) {
    const size_t Tp = (size_t)(T + 63) / 64 * 64;
    return (size_t)B * H * 4 * Tp * D * 2;
}

extern "C" int tq_attention_fwd(const float* qkv, float* out, float* lse, void* workspace, int B, int T, int H, int D,
                                hipStream_t stream) {
    if (!qkv || !out) return TQ_ERR_ARG;
    if (B <= 0 || T <= 0 || H <= 0) return TQ_ERR_SHAPE;
    if (workspace) {
        if (D == 64) return launch_attn2<64>(qkv, out, lse, workspace, B, T, H, stream);
        if (D == 32) return launch_attn2<32>(qkv, out, lse, workspace, B, T, H, stream);
        if (D != 128) return TQ_ERR_SHAPE;  // D = 128 (tiny config's middle block) stays on the first-generation kernel
    }
    if (D == 64) return launch_attn<64>(qkv, out, lse, B, T, H, stream);
    if (D == 32) return launch_attn<32>(qkv, out, lse, B, T, H, stream);
    if (D == 128) return launch_attn<128>(qkv, out, lse, B, T, H, stream);
    return TQ_ERR_SHAPE;
}

extern "C" int tq_attention_fwd_presplit(const float* qkv, const void* kv_planes, float* out, int B, int T, int H, int D,
                                         hipStream_t stream) {
    if (!qkv || !kv_planes || !out) return TQ_ERR_ARG;
    if (B <= 0 || T <= 0 || H <= 0) return TQ_ERR_SHAPE;
    if (D == 64) return launch_attn2<64>(qkv, out, nullptr, const_cast<void*>(kv_planes), B, T, H, stream, true);
    if (D == 32) return launch_attn2<32>(qkv, out, nullptr, const_cast<void*>(kv_planes), B, T, H, stream, true);
    return TQ_ERR_SHAPE;
}

// =================================================================================================
// Attention backward (flash-style recompute).  Per (b, head), with Qs = scale*Q, Ks = scale*K:
//   S = Qs Ks^T,  P = exp(S - lse),  O = P V,   delta_i = sum_d dO[i,d] O[i,d]
//   dV = P^T dO,  dP = dO V^T,  dS = P o (dP - delta),  dQ = scale * dS Ks,  dK = scale * dS^T Qs
// Pass A keeps 64 queries stationary and streams key tiles (dQ); pass B keeps 64 keys stationary and streams
// query tiles (dK, dV).  No atomics; P is recomputed in each pass.  Operands whose MFMA k index is the LDS row
// (key / query) are fetched with ds_read_b64_tr_b16 from the same row-major images the other products read.
// =================================================================================================
namespace {

typedef short s16x4b __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 tr_read(const unsigned char* p) {
    s16x4b v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4b*)(p));
    union { s16x4b s; uint2 u; } c;
    c.s = v;
    return c.u;
}

__global__ void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ d_o, float* __restrict__ delta, int T,
                                  int H, int D, size_t n) {
    // one wave per (b, t, h) row would be wasteful for D <= 128: one thread per row, 16-byte loads
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int h = (int)(i % H);
    const size_t bt = i / H;
    const int t = (int)(bt % T);
    const size_t b = bt / T;
    const float4* po = reinterpret_cast<const float4*>(o + bt * (size_t)(H * D) + h * D);
    const float4* pd = reinterpret_cast<const float4*>(d_o + bt * (size_t)(H * D) + h * D);
    float a = 0.f;
    for (int j = 0; j < D / 4; ++j) {
        const float4 x = po[j], y = pd[j];
        a += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    delta[(b * H + h) * T + t] = a;
}

// stage a [64 rows][D] fp32 tile (rows of `src` with row stride `rs`, optional scale) as bf16 hi/lo row-major images
template <int D>
__device__ __forceinline__ void stage_rows(const float* src, size_t rs, int row0, int T, float scale, unsigned char* hi,
                                           unsigned char* lo, int ROWB) {
    for (int i = threadIdx.x; i < 64 * (D / 4); i += 256) {
        const int r = i / (D / 4), c4 = i % (D / 4);
        float4 v = make_float4(0, 0, 0, 0);
        if (row0 + r < T) v = *reinterpret_cast<const float4*>(src + (size_t)(row0 + r) * rs + 4 * c4);
        const float u[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
        bf16x4 hv, lv;
#pragma unroll
        for (int j = 0; j < 4; ++j) { __bf16 hh, ll; split_bf16(u[j], hh, ll); hv[j] = hh; lv[j] = ll; }
        *reinterpret_cast<bf16x4*>(hi + r * ROWB + c4 * 8) = hv;
        *reinterpret_cast<bf16x4*>(lo + r * ROWB + c4 * 8) = lv;
    }
}

// A-operand fragments (row = l&15 of a 16-row block starting at row0, k = channel) straight from global memory
template <int D>
__device__ __forceinline__ void load_row_frags(const float* src, size_t rs, int row, bool ok, float scale, Frag (&fh)[D / 32],
                                               Frag (&fl)[D / 32]) {
    const int lane = threadIdx.x & 63;
    const float* p = src + (size_t)(ok ? row : 0) * rs + 8 * (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks) {
        float4 a = make_float4(0, 0, 0, 0), c = a;
        if (ok) { a = *reinterpret_cast<const float4*>(p + ks * 32); c = *reinterpret_cast<const float4*>(p + ks * 32 + 4); }
        const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) { __bf16 hh, ll; split_bf16(v[j] * scale, hh, ll); fh[ks].v[j] = hh; fl[ks].v[j] = ll; }
    }
}

// write a 16 x 64 accumulator tile set (4 column blocks) as bf16 hi/lo [row][col] image for use as an A operand
__device__ __forceinline__ void acc_to_image(const f32x4 (&s)[4], unsigned char* hi, unsigned char* lo, int ROWB) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            __bf16 hh, ll;
            split_bf16(s[cb][r], hh, ll);
            const int off = (4 * (lane >> 4) + r) * ROWB + (cb * 16 + (lane & 15)) * 2;
            *reinterpret_cast<__bf16*>(hi + off) = hh;
            *reinterpret_cast<__bf16*>(lo + off) = ll;
        }
}

// ---- pass A: dQ ------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 2) void attention_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                                  const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  float* __restrict__ dqkv, int T, int H, float scale) {
    constexpr int KS = D / 32, CB = D / 16;
    constexpr int ROWB = D * 2 + 16;
    constexpr int PROW = 64 * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* k_hi = lds;
    unsigned char* k_lo = k_hi + 64 * ROWB;
    unsigned char* v_hi = k_lo + 64 * ROWB;
    unsigned char* v_lo = v_hi + 64 * ROWB;
    unsigned char* p_base = v_lo + 64 * ROWB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqt = (T + 63) / 64;
    int bid = xcd_group_id(blockIdx.x, nqt, gridDim.x / nqt);  // the tiles of one (b, h) share an XCD's L2
    const int qt = bid % nqt; bid /= nqt;
    const int h = bid % H;
    const int b = bid / H;
    const int C3 = 3 * H * D, C1 = H * D;
    const float* base = qkv + (size_t)b * T * C3;
    const int q0 = qt * 64 + wave * 16;
    unsigned char* p_hi = p_base + wave * 2 * 16 * PROW;
    unsigned char* p_lo = p_hi + 16 * PROW;

    Frag qh[KS], ql[KS], gh[KS], gl[KS];
    {
        const int q = q0 + (lane & 15);
        load_row_frags<D>(base + h * D, C3, q, q < T, scale, qh, ql);
        load_row_frags<D>(d_o + (size_t)b * T * C1 + h * D, C1, q, q < T, 1.0f, gh, gl);
    }
    float lrow[4], drow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + 4 * (lane >> 4) + r;
        lrow[r] = (q < T) ? lse[((size_t)b * H + h) * T + q] : 0.f;
        drow[r] = (q < T) ? delta[((size_t)b * H + h) * T + q] : 0.f;
    }
    f32x4 dq[CB];
#pragma unroll
    for (int i = 0; i < CB; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = (T + 63) / 64;
    for (int kt = 0; kt < nkt; ++kt) {
        const int s0 = kt * 64;
        __syncthreads();
        stage_rows<D>(base + (H + h) * D, C3, s0, T, scale, k_hi, k_lo, ROWB);
        stage_rows<D>(base + (2 * H + h) * D, C3, s0, T, 1.0f, v_hi, v_lo, ROWB);
        __syncthreads();
        f32x4 s[4], dp[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            s[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int key = cb * 16 + (lane & 15);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag bh, bl;
                const int off = key * ROWB + (ks * 4 + (lane >> 4)) * 16;
                bh.u = *reinterpret_cast<const uint4*>(k_hi + off);
                bl.u = *reinterpret_cast<const uint4*>(k_lo + off);
                s[cb] = mfma_x3(qh[ks].v, ql[ks].v, bh.v, bl.v, s[cb]);
                bh.u = *reinterpret_cast<const uint4*>(v_hi + off);
                bl.u = *reinterpret_cast<const uint4*>(v_lo + off);
                dp[cb] = mfma_x3(gh[ks].v, gl[ks].v, bh.v, bl.v, dp[cb]);
            }
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const bool valid = (s0 + cb * 16 + (lane & 15)) < T;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = valid ? __expf(s[cb][r] - lrow[r]) : 0.f;
                s[cb][r] = pv * (dp[cb][r] - drow[r]);  // dS
            }
        }
        acc_to_image(s, p_hi, p_lo, PROW);
        __syncthreads();
        // dQ += dS Ks : A = dS image (row = query), B[k = key][col = d] via transposed reads of the K image
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag ah, al;
            const int poff = (lane & 15) * PROW + (ks * 4 + (lane >> 4)) * 16;
            ah.u = *reinterpret_cast<const uint4*>(p_hi + poff);
            al.u = *reinterpret_cast<const uint4*>(p_lo + poff);
            const int krow = ks * 32 + 8 * (lane >> 4) + ((lane >> 2) & 3);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                Frag bh, bl;
                const int off = krow * ROWB + (cb * 16 + 4 * (lane & 3)) * 2;
                bh.h[0] = tr_read(k_hi + off); bh.h[1] = tr_read(k_hi + off + 4 * ROWB);
                bl.h[0] = tr_read(k_lo + off); bl.h[1] = tr_read(k_lo + off + 4 * ROWB);
                dq[cb] = mfma_x3(ah.v, al.v, bh.v, bl.v, dq[cb]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + 4 * (lane >> 4) + r;
        if (q < T) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                dqkv[((size_t)b * T + q) * C3 + h * D + cb * 16 + (lane & 15)] = dq[cb][r] * scale;
        }
    }
}

// ---- pass B: dK, dV --------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 2) void attention_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                                   float* __restrict__ dqkv, int T, int H, float scale) {
    constexpr int KS = D / 32, CB = D / 16;
    constexpr int ROWB = D * 2 + 16;
    constexpr int PROW = 64 * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* q_hi = lds;
    unsigned char* q_lo = q_hi + 64 * ROWB;
    unsigned char* g_hi = q_lo + 64 * ROWB;
    unsigned char* g_lo = g_hi + 64 * ROWB;
    float* lq = reinterpret_cast<float*>(g_lo + 64 * ROWB);  // [64] lse of the query tile
    float* dq_ = lq + 64;                                     // [64] delta of the query tile
    unsigned char* p_base = reinterpret_cast<unsigned char*>(dq_ + 64);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nkt = (T + 63) / 64;
    int bid = xcd_group_id(blockIdx.x, nkt, gridDim.x / nkt);  // the tiles of one (b, h) share an XCD's L2
    const int kt = bid % nkt; bid /= nkt;
    const int h = bid % H;
    const int b = bid / H;
    const int C3 = 3 * H * D, C1 = H * D;
    const float* base = qkv + (size_t)b * T * C3;
    const int k0 = kt * 64 + wave * 16;
    unsigned char* p_hi = p_base + wave * 4 * 16 * PROW;   // P^T image
    unsigned char* p_lo = p_hi + 16 * PROW;
    unsigned char* s_hi = p_lo + 16 * PROW;                // dS^T image
    unsigned char* s_lo = s_hi + 16 * PROW;

    Frag kh[KS], kl[KS], vh[KS], vl[KS];
    {
        const int key = k0 + (lane & 15);
        load_row_frags<D>(base + (H + h) * D, C3, key, key < T, scale, kh, kl);
        load_row_frags<D>(base + (2 * H + h) * D, C3, key, key < T, 1.0f, vh, vl);
    }
    f32x4 dk[CB], dv[CB];
#pragma unroll
    for (int i = 0; i < CB; ++i) { dk[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int nqt = (T + 63) / 64;
    for (int qt = 0; qt < nqt; ++qt) {
        const int q0 = qt * 64;
        __syncthreads();
        stage_rows<D>(base + h * D, C3, q0, T, scale, q_hi, q_lo, ROWB);
        stage_rows<D>(d_o + (size_t)b * T * C1 + h * D, C1, q0, T, 1.0f, g_hi, g_lo, ROWB);
        if (tid < 64) {
            const bool ok = (q0 + tid) < T;
            lq[tid] = ok ? lse[((size_t)b * H + h) * T + q0 + tid] : 0.f;
            dq_[tid] = ok ? delta[((size_t)b * H + h) * T + q0 + tid] : 0.f;
        }
        __syncthreads();
        // S^T = Ks Qs^T,  dP^T = V dO^T   (16 keys x 64 queries)
        f32x4 s[4], dp[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            s[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int qq = cb * 16 + (lane & 15);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag bh, bl;
                const int off = qq * ROWB + (ks * 4 + (lane >> 4)) * 16;
                bh.u = *reinterpret_cast<const uint4*>(q_hi + off);
                bl.u = *reinterpret_cast<const uint4*>(q_lo + off);
                s[cb] = mfma_x3(kh[ks].v, kl[ks].v, bh.v, bl.v, s[cb]);
                bh.u = *reinterpret_cast<const uint4*>(g_hi + off);
                bl.u = *reinterpret_cast<const uint4*>(g_lo + off);
                dp[cb] = mfma_x3(vh[ks].v, vl[ks].v, bh.v, bl.v, dp[cb]);
            }
        }
        f32x4 ds[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int qq = cb * 16 + (lane & 15);
            const bool valid = (q0 + qq) < T;
            const float lv = lq[qq], dl = dq_[qq];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = valid ? __expf(s[cb][r] - lv) : 0.f;
                s[cb][r] = pv;
                ds[cb][r] = pv * (dp[cb][r] - dl);
            }
        }
        acc_to_image(s, p_hi, p_lo, PROW);
        acc_to_image(ds, s_hi, s_lo, PROW);
        __syncthreads();
        // dV += P^T dO,  dK += dS^T Qs : B[k = query][col = d] via transposed reads of the dO / Q images
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag ph, pl, sh_, sl_;
            const int poff = (lane & 15) * PROW + (ks * 4 + (lane >> 4)) * 16;
            ph.u = *reinterpret_cast<const uint4*>(p_hi + poff);
            pl.u = *reinterpret_cast<const uint4*>(p_lo + poff);
            sh_.u = *reinterpret_cast<const uint4*>(s_hi + poff);
            sl_.u = *reinterpret_cast<const uint4*>(s_lo + poff);
            const int qrow = ks * 32 + 8 * (lane >> 4) + ((lane >> 2) & 3);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                Frag bh, bl;
                const int off = qrow * ROWB + (cb * 16 + 4 * (lane & 3)) * 2;
                bh.h[0] = tr_read(g_hi + off); bh.h[1] = tr_read(g_hi + off + 4 * ROWB);
                bl.h[0] = tr_read(g_lo + off); bl.h[1] = tr_read(g_lo + off + 4 * ROWB);
                dv[cb] = mfma_x3(ph.v, pl.v, bh.v, bl.v, dv[cb]);
                bh.h[0] = tr_read(q_hi + off); bh.h[1] = tr_read(q_hi + off + 4 * ROWB);
                bl.h[0] = tr_read(q_lo + off); bl.h[1] = tr_read(q_lo + off + 4 * ROWB);
                dk[cb] = mfma_x3(sh_.v, sl_.v, bh.v, bl.v, dk[cb]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int key = k0 + 4 * (lane >> 4) + r;
        if (key < T) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const size_t o = ((size_t)b * T + key) * C3 + cb * 16 + (lane & 15);
                dqkv[o + (H + h) * D] = dk[cb][r] * scale;
                dqkv[o + (2 * H + h) * D] = dv[cb][r];
            }
        }
    }
}

template <int D>
int launch_attn_bwd(const float* qkv, const float* out, const float* d_o, const float* lse, float* delta, float* dqkv, int B,
                    int T, int H, hipStream_t stream) {
    constexpr int ROWB = D * 2 + 16, PROW = 64 * 2 + 16;
    const size_t n = (size_t)B * T * H;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, d_o, delta, T, H, D, n);
    TQ_CHECK_LAUNCH();
    const float scale = (float)(1.0 / sqrt(sqrt((double)D)));
    const int nt = (T + 63) / 64;
    const size_t shA = 4 * 64 * ROWB + 4 * 2 * 16 * PROW;
    const size_t shB = 4 * 64 * ROWB + 128 * sizeof(float) + 4 * 4 * 16 * PROW;
    if (shA > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_dq_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shA);
    if (shB > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_dkv_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shB);
    hipLaunchKernelGGL(attention_bwd_dq_kernel<D>, dim3(B * H * nt), dim3(256), shA, stream, qkv, d_o, lse, delta, dqkv, T, H, scale);
    TQ_CHECK_LAUNCH();
    hipLaunchKernelGGL(attention_bwd_dkv_kernel<D>, dim3(B * H * nt), dim3(256), shB, stream, qkv, d_o, lse, delta, dqkv, T, H, scale);
    TQ_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" int tq_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* delta,
                                float* dqkv, int B, int T, int H, int D, hipStream_t stream) {
    if (!qkv || !out || !dout || !lse || !delta || !dqkv) return TQ_ERR_ARG;
    if (B <= 0 || T <= 0 || H <= 0) return TQ_ERR_SHAPE;
    if (D == 64) return launch_attn_bwd<64>(qkv, out, dout, lse, delta, dqkv, B, T, H, stream);
    if (D == 32) return launch_attn_bwd<32>(qkv, out, dout, lse, delta, dqkv, B, T, H, stream);
    if (D == 128) return launch_attn_bwd<128>(qkv, out, dout, lse, delta, dqkv, B, T, H, stream);
    return TQ_ERR_SHAPE;
}
