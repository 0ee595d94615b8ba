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
__global__ __launch_bounds__(256, 2) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T,
                                                           int H, float scale) {
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
    int bid = blockIdx.x;
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
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                out[((size_t)b * T + q) * (H * D) + h * D + cb * 16 + (lane & 15)] = o[cb][r] * inv;
        }
    }
}

template <int D>
int launch_attn(const float* qkv, float* out, int B, int T, int H, hipStream_t stream) {
    constexpr int KROW = D * 2 + 16, VROW = KTILE * 2 + 16, PROW = KTILE * 2 + 16;
    const size_t sh = 2 * KTILE * KROW + 2 * D * VROW + 4 * 2 * 16 * PROW;
    if (sh > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    const int nqt = (T + QT - 1) / QT;
    const float scale = (float)(1.0 / sqrt(sqrt((double)D)));  // blocks.py:173 (python double, then fp32)
    hipLaunchKernelGGL(attention_kernel<D>, dim3(B * H * nqt), dim3(256), sh, stream, qkv, out, T, H, scale);
    TQ_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" int tq_attention_fwd(const float* qkv, float* out, int B, int T, int H, int D, hipStream_t stream) {
    if (!qkv || !out) return TQ_ERR_ARG;
    if (B <= 0 || T <= 0 || H <= 0) return TQ_ERR_SHAPE;
    if (D == 64) return launch_attn<64>(qkv, out, B, T, H, stream);
    if (D == 32) return launch_attn<32>(qkv, out, B, T, H, stream);
    if (D == 128) return launch_attn<128>(qkv, out, B, T, H, stream);
    return TQ_ERR_SHAPE;
}
