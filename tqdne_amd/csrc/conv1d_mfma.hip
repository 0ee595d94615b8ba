// C ABI of the fused 1-D convolution (tq_conv1d_fwd*, tq_conv1d_bwd_data) and the weight packers.  The kernel itself
// (conv1d_kernel.hpp) is instantiated in conv1d_fwd_k5a/k5b/k13.hip, conv1d_resample.hip and conv1d_dgrad.hip; this file validates
// the descriptors, fills ConvArgs and hands over to those translation units' entry points (conv_args.hpp).
#include <cmath>
#include <cstdlib>

#include "common.hpp"
#include "conv_args.hpp"
#include "../../include/tqdne_hip.h"

using namespace tq;

extern "C" int tq_conv_tile_co(int C_out) {
    // output-channel tile the dispatcher will use; packed weights are padded to it
    if (C_out % 128 == 0) return 128;
    if (C_out % 64 == 0) return 64;
    return 32;
}

static int conv_exp_stagger() {   // (TQDNE_CONV_STAGGER: see -DTQ_EXP_STAGGER in conv1d_kernel.hpp; 0 in default builds' kernels: unread)
    static const int v = [] { const char* e = getenv("TQDNE_CONV_STAGGER"); return e ? atoi(e) : 0; }();
    return v;
}

static int conv1d_fwd_impl(const TqConvDesc* d, const float* x0, const float* x1, const float* gscale, const float* gshift,
                           const void* wpk, const float* bias, const float* emb, const float* res, const float* skip_x0,
                           const float* skip_x1, const float* skip_bias, float* y, float* stats, hipStream_t stream,
                           unsigned char* kv_planes = nullptr, int kvH = 0, int kvD = 0, int kvTp = 0, float kvscale = 1.f, int kv_vf16 = 0);

extern "C" int tq_conv1d_fwd_qkv(const TqConvDesc* d, const float* x, const float* gscale, const float* gshift, const void* wpk,
                                 const float* bias, float* qkv, void* kv_planes, int H, int D, int v_format, hipStream_t stream) {
    if (!d || !kv_planes || H <= 0 || (D != 32 && D != 64)) return TQ_ERR_ARG;
    if (v_format != TQ_KV_V_BF16 && v_format != TQ_KV_V_F16) return TQ_ERR_ARG;
    if (d->ktaps != 1 || d->stride != 1 || d->upsample || d->C_in1 || d->C_out != 3 * H * D || (d->C_skip0 | d->C_skip1)) return TQ_ERR_SHAPE;
    if (d->flags & (TQ_CONV_EMB | TQ_CONV_RES | TQ_CONV_STATS | TQ_CONV_DROPOUT | TQ_CONV_SILU)) return TQ_ERR_ARG;
    const int Tp = (d->T_out + 63) / 64 * 64;
    const float scale = (float)(1.0 / sqrt(sqrt((double)D)));
    return conv1d_fwd_impl(d, x, nullptr, gscale, gshift, wpk, bias, nullptr, nullptr, nullptr, nullptr, nullptr, qkv, nullptr, stream,
                           reinterpret_cast<unsigned char*>(kv_planes), H, D, Tp, scale, v_format == TQ_KV_V_F16 ? 1 : 0);
}

extern "C" int tq_conv1d_fwd(const TqConvDesc* d, const float* x0, const float* x1, const float* gscale,
                             const float* gshift, const void* wpk, const float* bias, const float* emb,
                             const float* res, float* y, float* stats, hipStream_t stream) {
    if (d && (d->C_skip0 || d->C_skip1)) return TQ_ERR_ARG;  // fused-skip descriptors go through tq_conv1d_fwd_skip
    return conv1d_fwd_impl(d, x0, x1, gscale, gshift, wpk, bias, emb, res, nullptr, nullptr, nullptr, y, stats, stream);
}

extern "C" int tq_conv1d_fwd_skip(const TqConvDesc* d, const float* x0, const float* x1, const float* gscale,
                                  const float* gshift, const void* wpk_main_then_skip, const float* bias, const float* emb,
                                  const float* skip_x0, const float* skip_x1, const float* skip_bias, float* y, float* stats,
                                  hipStream_t stream) {
    if (!d || !skip_x0 || d->C_skip0 <= 0 || d->C_skip0 % 32 || d->C_skip1 < 0 || d->C_skip1 % 32) return TQ_ERR_ARG;
    if (d->C_skip1 > 0 && !skip_x1) return TQ_ERR_ARG;
    if (d->stride != 1 || d->upsample || (d->flags & TQ_CONV_RES)) return TQ_ERR_SHAPE;
    return conv1d_fwd_impl(d, x0, x1, gscale, gshift, wpk_main_then_skip, bias, emb, nullptr, skip_x0, skip_x1, skip_bias, y,
                           stats, stream);
}

static int conv1d_fwd_impl(const TqConvDesc* d, const float* x0, const float* x1, const float* gscale, const float* gshift,
                           const void* wpk, const float* bias, const float* emb, const float* res, const float* skip_x0,
                           const float* skip_x1, const float* skip_bias, float* y, float* stats, hipStream_t stream,
                           unsigned char* kv_planes, int kvH, int kvD, int kvTp, float kvscale, int kv_vf16) {
    if (!d || !x0 || !wpk || !y) return TQ_ERR_ARG;
    if (d->C_in0 <= 0 || d->C_in0 % 32 || d->C_in1 < 0 || d->C_in1 % 32 || d->C_out <= 0 || d->C_out % 32) return TQ_ERR_SHAPE;
    if (d->C_in1 > 0 && !x1) return TQ_ERR_ARG;
    if ((d->flags & TQ_CONV_GN) && (!gscale || !gshift)) return TQ_ERR_ARG;
    if ((d->flags & TQ_CONV_EMB) && !emb) return TQ_ERR_ARG;
    if ((d->flags & TQ_CONV_RES) && !res) return TQ_ERR_ARG;
    if ((d->flags & TQ_CONV_STATS) && !stats) return TQ_ERR_ARG;
    if (d->B <= 0 || d->T_in <= 0 || d->T_out <= 0) return TQ_ERR_SHAPE;
    if (d->stride == 1) {
        const int Tsrc = d->upsample ? 2 * d->T_in : d->T_in;
        if (d->T_out != Tsrc || d->pad != d->ktaps / 2) return TQ_ERR_SHAPE;
    } else if (d->stride == 2) {
        if (d->ktaps != 3 || d->pad != 1 || d->upsample || d->T_out != (d->T_in + 2 - 3) / 2 + 1) return TQ_ERR_SHAPE;
    } else {
        return TQ_ERR_SHAPE;
    }
    ConvArgs a;
    a.x0 = x0; a.x1 = x1; a.gscale = gscale; a.gshift = gshift;
    a.wpk = reinterpret_cast<const uint4*>(wpk);
    a.bias = bias; a.emb = emb; a.res = res; a.y = y; a.stats = stats;
    a.B = d->B; a.T_in = d->T_in; a.T_out = d->T_out; a.C0 = d->C_in0; a.C1 = d->C_in1; a.C_out = d->C_out;
    a.emb_stride = d->emb_stride; a.flags = d->flags;
    const int tile = tq_conv_tile_co(d->C_out);
    a.ncob_pad = ((d->C_out + tile - 1) / tile) * tile / 16;
    a.nslots = (d->T_out + STAT_SLOT - 1) / STAT_SLOT;
    a.t_tile = 0;
    if (d->t_tile) {   // the small tile: see TqConvDesc.t_tile for what it is built for (the dispatcher refuses the rest)
        const int need = TQ_CONV_GN | TQ_CONV_SILU;
        if (d->t_tile != 32) return TQ_ERR_ARG;
        if (d->ktaps != 5 || d->stride != 1 || d->upsample || kv_planes || (d->flags & need) != need || (d->flags & TQ_CONV_POLY2) || d->gn_fuse)
            return TQ_ERR_SHAPE;
        a.t_tile = 32;
        a.nslots = (d->T_out + 31) / 32;
    }
    if (d->flags & TQ_CONV_POLY2) {
        if (d->ktaps != 3 || d->stride != 1 || d->upsample || (d->C_out & 63) || kv_planes || d->C_skip0 || d->C_skip1) return TQ_ERR_SHAPE;
        // the statistics tensor has ceil(2 T / 128) slots (what the consuming tq_gn_finalize assumes); this launch fills 2 ceil(T / 128):
        // the same number when T is a multiple of 128 or leaves more than half a slot (T = 508, 1016, 2032 of the 4064-sample signals)
        if ((d->flags & TQ_CONV_STATS) && d->T_out % STAT_SLOT && d->T_out % STAT_SLOT <= STAT_SLOT / 2) return TQ_ERR_SHAPE;
        a.nslots *= 2;
    }
    a.drop_site = d->dropout_site;
    a.drop_seed = d->dropout_seed;
    float pdrop = d->dropout_p;
    if (!(d->flags & TQ_CONV_DROPOUT) || pdrop <= 0.f) { a.flags &= ~TQ_CONV_DROPOUT; pdrop = 0.f; }
    if (pdrop >= 1.f) return TQ_ERR_ARG;
    a.drop_thresh = (uint32_t)((double)pdrop * 4294967296.0);
    a.drop_scale = 1.0f / (1.0f - pdrop);
    a.fx0 = a.fx1 = a.fgs = a.fgh = nullptr; a.y1 = nullptr; a.OC0 = d->C_out; a.bflags = 0;
    a.sx0 = skip_x0; a.sx1 = skip_x1; a.sbias = skip_bias; a.sC0 = d->C_skip0; a.sC1 = d->C_skip1;
    a.wfmt = d->wfmt;
    a.kv = kv_planes; a.kvH = kvH; a.kvD = kvD; a.kvTp = kvTp; a.kvscale = kvscale; a.kv_vf16 = (kv_planes && kv_vf16) ? 1 : 0;
    a.range_flag = d->range_flag;
    a.in_amax = nullptr;
    a.gf_counters = nullptr; a.exp_stagger = conv_exp_stagger(); a.gf_partner = nullptr; a.gf_Cp = 0; a.gf_partner_first = 0; a.gf_narrive = 0;
    a.gf_gamma = a.gf_beta = nullptr; a.gf_gscale = a.gf_gshift = a.gf_mean_rstd = nullptr;
    a.cf_st0 = a.cf_st1 = a.cf_gamma = a.cf_beta = nullptr; a.cf_mean_rstd = nullptr; a.cf_ns0 = a.cf_ns1 = 0;
    if (d->gn_fold) {   // consumer-side GroupNorm fold (ABI 7): this launch forms its own folded coefficients (see TqGnFold)
        const TqGnFold* f = d->gn_fold;
        // built for the small tile, and (experiment) for the default tiles of the fp16 + MX-fp6 scheme -- the dispatcher refuses the rest
        if (!(d->flags & TQ_CONV_GN) || kv_planes || d->gn_fuse || (a.t_tile != 32 && d->wfmt != TQ_WFMT_F16_MX6)) return TQ_ERR_SHAPE;
        if (!f->stats0 || !f->gamma || !f->beta || (d->C_in1 > 0 && !f->stats1)) return TQ_ERR_ARG;
        const int s0 = f->slot0 ? f->slot0 : STAT_SLOT, s1 = f->slot1 ? f->slot1 : STAT_SLOT;
        if ((s0 != STAT_SLOT && s0 != 32) || (s1 != STAT_SLOT && s1 != 32)) return TQ_ERR_ARG;
        if ((d->C_in0 + d->C_in1) % GN_GROUPS) return TQ_ERR_SHAPE;
        a.cf_st0 = f->stats0; a.cf_st1 = d->C_in1 > 0 ? f->stats1 : nullptr;
        a.cf_ns0 = (d->T_in + s0 - 1) / s0; a.cf_ns1 = (d->T_in + s1 - 1) / s1;
        a.cf_gamma = f->gamma; a.cf_beta = f->beta; a.cf_mean_rstd = f->mean_rstd;
    }
#ifndef TQ_BUILD_EXPERIMENTS
    if (d->gn_fuse) return TQ_ERR_ARG;   // reserved: the in-launch GroupNorm fold is an experiment (TQDNE_BUILD_EXPERIMENTS=1 builds it)
#else
    if (d->gn_fuse) {
        const TqGnFuse* g = d->gn_fuse;
        if (!(d->flags & TQ_CONV_STATS) || kv_planes || !g->counters || !g->gamma || !g->beta || !g->gscale || !g->gshift) return TQ_ERR_ARG;
        if (g->C_partner < 0 || (g->C_partner > 0 && !g->partner_stats)) return TQ_ERR_ARG;
        const int Cown = (d->flags & TQ_CONV_POLY2) ? d->C_out / 2 : d->C_out;
        if ((Cown + g->C_partner) % GN_GROUPS) return TQ_ERR_SHAPE;
        a.gf_counters = g->counters; a.gf_partner = g->partner_stats; a.gf_Cp = g->C_partner; a.gf_partner_first = g->partner_first;
        a.gf_gamma = g->gamma; a.gf_beta = g->beta; a.gf_gscale = g->gscale; a.gf_gshift = g->gshift; a.gf_mean_rstd = g->mean_rstd;
    }
#endif

    if (d->stride == 2 || d->upsample) {
        if (a.flags & (TQ_CONV_GN | TQ_CONV_SILU | TQ_CONV_DROPOUT)) return TQ_ERR_SHAPE;  // resampling convs take raw inputs
        return conv_launch_resample(a, d->ktaps, d->stride, stream);
    }
    if (a.kv) {  // qkv projection with the pre-split K / V epilogue: k = 1, plain or folded-GN prologue
        return conv_launch_qkv(a, stream);
    }
#ifdef TQ_BUILD_EXPERIMENTS
    {   // the one-wave-per-SIMD kernel where it is built (conv1d_w4.hip); TQ_ERR_SHAPE = "not mine", nothing launched
        const int rc = conv1d_w4_launch(a, d->ktaps, stream);
        if (rc != TQ_ERR_SHAPE) return rc;
    }
#endif
    switch (d->ktaps) {
        case 1: return conv_launch_fwd_k1(a, stream);
        case 3: return conv_launch_fwd_k3(a, stream);
        case 5: return conv_launch_fwd_k5(a, stream);
        default: return TQ_ERR_SHAPE;
    }
}

// Data gradient of a stride-1 "same" convolution (the transposed, tap-flipped weights are packed with mode 1):
//   acc[b,t,ci] = sum_{k,co} W[co,ci,K-1-k] * dy[b, t + k - pad, co]
// followed by the chain rule through the forward conv's prologue (see the EPI == 1 epilogue above).
extern "C" int tq_conv1d_bwd_data(const TqConvBwdDesc* d, const float* dy, const void* wpk_t, const float* x0,
                                  const float* x1, const float* gscale, const float* gshift, float* dx0, float* dx1,
                                  float* gstats, hipStream_t stream) {
    if (!d || !dy || !wpk_t || !dx0) return TQ_ERR_ARG;
    if (d->C_dy <= 0 || d->C_dy % 32 || d->C_dx0 <= 0 || d->C_dx0 % 32 || d->C_dx1 < 0 || d->C_dx1 % 32) return TQ_ERR_SHAPE;
    if (d->C_dx1 > 0 && !dx1) return TQ_ERR_ARG;
    const int need_x = d->flags & (TQ_BWD_GN | TQ_BWD_SILU | TQ_BWD_STATS);
    if (need_x && (!x0 || (d->C_dx1 > 0 && !x1))) return TQ_ERR_ARG;
    if ((d->flags & TQ_BWD_GN) && (!gscale || !gshift)) return TQ_ERR_ARG;
    if ((d->flags & TQ_BWD_STATS) && !gstats) return TQ_ERR_ARG;
    if (d->B <= 0 || d->T <= 0 || (d->ktaps != 1 && d->ktaps != 3 && d->ktaps != 5)) return TQ_ERR_SHAPE;
    ConvArgs a;
    a.x0 = dy; a.x1 = nullptr; a.gscale = nullptr; a.gshift = nullptr;
    a.wpk = reinterpret_cast<const uint4*>(wpk_t);
    a.bias = nullptr; a.emb = nullptr; a.res = nullptr; a.y = dx0; a.stats = gstats;
    a.B = d->B; a.T_in = d->T; a.T_out = d->T; a.C0 = d->C_dy; a.C1 = 0; a.C_out = d->C_dx0 + d->C_dx1;
    a.emb_stride = 0; a.flags = 0;
    const int tile = tq_conv_tile_co(a.C_out);
    a.ncob_pad = ((a.C_out + tile - 1) / tile) * tile / 16;
    a.nslots = (d->T + STAT_SLOT - 1) / STAT_SLOT;
    a.drop_site = d->dropout_site; a.drop_seed = d->dropout_seed;
    float pdrop = d->dropout_p;
    a.bflags = d->flags;
    if (!(d->flags & TQ_BWD_DROPOUT) || pdrop <= 0.f) { a.bflags &= ~TQ_BWD_DROPOUT; pdrop = 0.f; }
    if (pdrop >= 1.f) return TQ_ERR_ARG;
    a.drop_thresh = (uint32_t)((double)pdrop * 4294967296.0);
    a.drop_scale = 1.0f / (1.0f - pdrop);
    a.fx0 = x0; a.fx1 = x1; a.fgs = gscale; a.fgh = gshift; a.y1 = dx1; a.OC0 = d->C_dx0;
    a.sx0 = a.sx1 = a.sbias = nullptr; a.sC0 = a.sC1 = 0;
    // contraction scheme: bf16x3 (fp32 range), or fp16 + MX-fp6 on dy scaled by the power of two that *dy_amax selects (1.5 instead of
    // 3 MFMA products per multiply-add); the packed weights must be in the matching format (pack mode 1 / 5)
    a.wfmt = TQ_WFMT_BF16X3;
    a.in_amax = nullptr;
    a.t_tile = 0;
    if (d->wfmt == TQ_WFMT_F16_MX6) {
        if (!d->dy_amax) return TQ_ERR_ARG;
        if (d->C_dy % 64 || a.C_out % 64) return TQ_ERR_SHAPE;
        a.wfmt = TQ_WFMT_F16_MX6;
        a.in_amax = d->dy_amax;
    } else if (d->wfmt != TQ_WFMT_BF16X3) {
        return TQ_ERR_ARG;
    }
    a.kv = nullptr; a.kvH = a.kvD = a.kvTp = 0; a.kvscale = 1.f; a.kv_vf16 = 0;
    a.range_flag = nullptr;
    a.gf_counters = nullptr; a.exp_stagger = conv_exp_stagger(); a.gf_partner = nullptr; a.gf_Cp = 0; a.gf_partner_first = 0; a.gf_narrive = 0;
    a.gf_gamma = a.gf_beta = nullptr; a.gf_gscale = a.gf_gshift = a.gf_mean_rstd = nullptr;
    a.cf_st0 = a.cf_st1 = a.cf_gamma = a.cf_beta = nullptr; a.cf_mean_rstd = nullptr; a.cf_ns0 = a.cf_ns1 = 0;
    return conv_launch_dgrad(a, d->ktaps, stream);
}

// ------------------------------------------------------------------------------------------------
// Weight packing: torch Conv1d weight (C_out, C_in, K) fp32 -> per-lane MFMA A fragments, bf16 hi/lo.
//   packed[((chunk*K + tap) * ncob_pad + cob) * 2 + hl][lane][8]   (8 bf16 = 16 bytes)
//   lane l holds A[row = cob*16 + (l&15)][k = chunk*32 + 8*(l>>4) + j], j = 0..7
// mode 0: A[row=co][k=ci] = W[co][ci][tap]                           (forward)
// mode 1: A[row=ci][k=co] = W[co][ci][K-1-tap]                       (data gradient: transposed, flipped)
// mode 2: forward, TQ_WFMT_F16_MX8 (same total size as mode 0 when 64 | C_in): per 64-channel chunk, tap and 16-row block four
//   16-byte fragments per lane (row = cob*16 + (l&15), kq = l>>4):  [0] fp16 W of channels 8 kq + j,  [1] of 32 + 8 kq + j (j < 8);
//   [2] fp8(W) of channels 16 kq + j,  [3] fp8((W - fp16(W)) * 2^12) of the same channels (j < 16)
// ------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ void pack_mx_body(const float* __restrict__ w, int C_out, int C_in, int K, int ncob_pad,
                                             uint4* __restrict__ out, size_t gid) {
    const int nchunks = (C_in + 63) / 64;
    const size_t total = (size_t)nchunks * K * ncob_pad * 64;
    if (gid >= total) return;
    const int lane = gid & 63;
    size_t r = gid >> 6;
    const int cob = r % ncob_pad; r /= ncob_pad;
    const int tap = r % K;
    const int chunk = r / K;
    const int row = cob * 16 + (lane & 15), kq = lane >> 4;
    auto wv = [&](int ch) -> float {
        return (row < C_out && ch < C_in) ? w[((size_t)row * C_in + ch) * K + tap] : 0.f;
    };
    f16x8 m0, m1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        m0[j] = (_Float16)wv(chunk * 64 + 8 * kq + j);
        m1[j] = (_Float16)wv(chunk * 64 + 32 + 8 * kq + j);
    }
    int c0[4], c1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float a[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = wv(chunk * 64 + 16 * kq + 4 * q + j);
            a[j] = __builtin_amdgcn_fmed3f(v, -448.f, 448.f);
            l[j] = __builtin_amdgcn_fmed3f((v - (float)(_Float16)v) * 4096.f, -448.f, 448.f);
        }
        c0[q] = __builtin_amdgcn_cvt_pk_fp8_f32(a[0], a[1], 0, false);
        c0[q] = __builtin_amdgcn_cvt_pk_fp8_f32(a[2], a[3], c0[q], true);
        c1[q] = __builtin_amdgcn_cvt_pk_fp8_f32(l[0], l[1], 0, false);
        c1[q] = __builtin_amdgcn_cvt_pk_fp8_f32(l[2], l[3], c1[q], true);
    }
    const size_t o = ((((size_t)chunk * K + tap) * ncob_pad + cob) * 4) * 64 + lane;
    out[o] = __builtin_bit_cast(uint4, m0);
    out[o + 64] = __builtin_bit_cast(uint4, m1);
    out[o + 128] = make_uint4((unsigned)c0[0], (unsigned)c0[1], (unsigned)c0[2], (unsigned)c0[3]);
    out[o + 192] = make_uint4((unsigned)c1[0], (unsigned)c1[1], (unsigned)c1[2], (unsigned)c1[3]);
}

// mode 3 (TQ_WFMT_F16_MX6): fragments [0], [1] as mode 2; [2] = dwords 0..3 and [3] = {dwords 4, 5, E8M0 byte, 0} of the lane's
// e2m3 block: channels 16 kq + i, element 2i = W, element 2i + 1 = (W - fp16(W)) * 2^12, both divided by the block's 2^e
__global__ void pack_conv_weight_mx_kernel(const float* __restrict__ w, int C_out, int C_in, int K, int ncob_pad,
                                           uint4* __restrict__ out) {
    pack_mx_body(w, C_out, C_in, K, ncob_pad, out, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// TRANSPOSED (mode 5, data gradient): A[row = ci][k = co] = W[co][ci][K - 1 - tap], i.e. rows run over the forward conv's input
// channels and the contraction over its output channels
template <bool TRANSPOSED = false>
__device__ __forceinline__ void pack_mx6_body(const float* __restrict__ w, int C_out, int C_in, int K, int ncob_pad,
                                              uint4* __restrict__ out, size_t gid) {
    const int rows = TRANSPOSED ? C_in : C_out, kdim = TRANSPOSED ? C_out : C_in;
    const int nchunks = (kdim + 63) / 64;
    const size_t total = (size_t)nchunks * K * ncob_pad * 64;
    if (gid >= total) return;
    const int lane = gid & 63;
    size_t r = gid >> 6;
    const int cob = r % ncob_pad; r /= ncob_pad;
    const int tap = r % K;
    const int chunk = r / K;
    const int row = cob * 16 + (lane & 15), kq = lane >> 4;
    auto wv = [&](int ch) -> float {
        if (!(row < rows && ch < kdim)) return 0.f;
        return TRANSPOSED ? w[((size_t)ch * C_in + row) * K + (K - 1 - tap)] : w[((size_t)row * C_in + ch) * K + tap];
    };
    f16x8 m0, m1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        m0[j] = (_Float16)wv(chunk * 64 + 8 * kq + j);
        m1[j] = (_Float16)wv(chunk * 64 + 32 + 8 * kq + j);
    }
    f32x16 wa, wl;
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float v = wv(chunk * 64 + 16 * kq + i);
        const float l = (v - (float)(_Float16)v) * 4096.f;
        wa[i] = v; wl[i] = l;
        mx = fmaxf(mx, fmaxf(fabsf(v), fabsf(l)));
    }
    const unsigned ba = e8m0_block_scale(mx);
    const u32x6 pk = cvt_2xpk16_fp6(wa, wl, __uint_as_float(ba << 23));
    const size_t o = ((((size_t)chunk * K + tap) * ncob_pad + cob) * 4) * 64 + lane;
    out[o] = __builtin_bit_cast(uint4, m0);
    out[o + 64] = __builtin_bit_cast(uint4, m1);
    out[o + 128] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    out[o + 192] = make_uint4(pk[4], pk[5], ba, 0u);
}

__global__ void pack_conv_weight_mx6_kernel(const float* __restrict__ w, int C_out, int C_in, int K, int ncob_pad,
                                            uint4* __restrict__ out, int transposed) {
    if (transposed) pack_mx6_body<true>(w, C_out, C_in, K, ncob_pad, out, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
    else pack_mx6_body<false>(w, C_out, C_in, K, ncob_pad, out, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

__device__ __forceinline__ void pack_bf16_body(const float* __restrict__ w, int C_out, int C_in, int K, int mode, int rows, int kdim,
                                               int ncob_pad, uint4* __restrict__ out, size_t gid) {
    const int nchunks = (kdim + 31) / 32;
    const size_t total = (size_t)nchunks * K * ncob_pad * 64;
    if (gid >= total) return;
    const int lane = gid & 63;
    size_t r = gid >> 6;
    const int cob = r % ncob_pad; r /= ncob_pad;
    const int tap = r % K;
    const int chunk = r / K;
    const int row = cob * 16 + (lane & 15);
    Frag hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kk = chunk * 32 + 8 * (lane >> 4) + j;
        float v = 0.f;
        if (row < rows && kk < kdim) {
            if (mode == 0) v = w[((size_t)row * C_in + kk) * K + tap];
            else           v = w[((size_t)kk * C_in + row) * K + (K - 1 - tap)];
        }
        __bf16 h, l;
        split_bf16(v, h, l);
        hi.v[j] = h; lo.v[j] = l;
    }
    const size_t o = ((((size_t)chunk * K + tap) * ncob_pad + cob) * 2) * 64 + lane;
    out[o] = hi.u;
    out[o + 64] = lo.u;
}

__global__ void pack_conv_weight_kernel(const float* __restrict__ w, int C_out, int C_in, int K, int mode,
                                        int rows, int kdim, int ncob_pad, uint4* __restrict__ out) {
    pack_bf16_body(w, C_out, C_in, K, mode, rows, kdim, ncob_pad, out, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// Every weight (re)pack of a plan in ONE launch: after an optimizer step the paper UNet re-packs ~160 tensors (forward fragments,
// transposed fragments for the data gradients) and gathers 44 embedding-projection tensors into their concatenated buffers -- as
// separate 4-8 us launches that was ~1.2 ms of a 28 ms training step at < 0.1 waves per SIMD.  Workgroup -> job by binary search
// over the jobs' first block.
__global__ __launch_bounds__(256) void pack_jobs_kernel(const TqPackJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const TqPackJob jb = jobs[lo];
    const size_t gid = (size_t)(blockIdx.x - jb.block_begin) * 256 + threadIdx.x;
    const float* w = reinterpret_cast<const float*>(jb.src);
    if (jb.mode == 4) {   // plain copy of C_out floats
        if (gid < (size_t)jb.C_out) reinterpret_cast<float*>(jb.dst)[gid] = w[gid];
        return;
    }
    const bool tr = jb.mode == 1 || jb.mode == 5;
    const int rows = tr ? jb.C_in : jb.C_out;
    const int kdim = tr ? jb.C_out : jb.C_in;
    const int tile = (rows % 128 == 0) ? 128 : ((rows % 64 == 0) ? 64 : 32);   // = tq_conv_tile_co
    const int ncob_pad = ((rows + tile - 1) / tile) * tile / 16;
    uint4* out = reinterpret_cast<uint4*>(jb.dst);
    if (jb.mode == 2) pack_mx_body(w, jb.C_out, jb.C_in, jb.K, ncob_pad, out, gid);
    else if (jb.mode == 3) pack_mx6_body<false>(w, jb.C_out, jb.C_in, jb.K, ncob_pad, out, gid);
    else if (jb.mode == 5) pack_mx6_body<true>(w, jb.C_out, jb.C_in, jb.K, ncob_pad, out, gid);
    else pack_bf16_body(w, jb.C_out, jb.C_in, jb.K, jb.mode, rows, kdim, ncob_pad, out, gid);
}
}  // namespace

extern "C" int tq_pack_job_blocks(int C_out, int C_in, int K, int mode) {
    if (C_out <= 0 || mode < 0 || mode > 5) return 0;
    if (mode == 4) return (C_out + 255) / 256;
    if (C_in <= 0 || K <= 0) return 0;
    const bool tr = mode == 1 || mode == 5;
    const int rows = tr ? C_in : C_out;
    const int kdim = tr ? C_out : C_in;
    const int tile = tq_conv_tile_co(rows);
    const int ncob_pad = ((rows + tile - 1) / tile) * tile / 16;
    const size_t total = (mode == 2 || mode == 3 || mode == 5) ? (size_t)((kdim + 63) / 64) * K * ncob_pad * 64
                                                               : (size_t)((kdim + 31) / 32) * K * ncob_pad * 64;
    return (int)((total + 255) / 256);
}

extern "C" int tq_pack_jobs(const TqPackJob* jobs_device, int njobs, int total_blocks, hipStream_t stream) {
    if (!jobs_device || njobs <= 0 || total_blocks <= 0) return TQ_ERR_ARG;
    hipLaunchKernelGGL(pack_jobs_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, jobs_device, njobs);
    TQ_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t tq_conv_weight_pack_bytes(int C_out, int C_in, int K, int mode) {
    const bool tr = mode == 1 || mode == 5;
    const int rows = tr ? C_in : C_out;
    const int kdim = tr ? C_out : C_in;
    const int tile = tq_conv_tile_co(rows);
    const int ncob_pad = ((rows + tile - 1) / tile) * tile / 16;
    if (mode == 2 || mode == 3 || mode == 5) return (size_t)((kdim + 63) / 64) * K * ncob_pad * 4 * 64 * 16;
    return (size_t)((kdim + 31) / 32) * K * ncob_pad * 2 * 64 * 16;
}

extern "C" int tq_pack_conv_weight(const float* w, int C_out, int C_in, int K, int mode, void* out, hipStream_t stream) {
    if (!w || !out || C_out <= 0 || C_in <= 0 || K <= 0 || mode < 0 || mode > 5 || mode == 4) return TQ_ERR_ARG;
    const bool tr = mode == 1 || mode == 5;
    const int rows = tr ? C_in : C_out;
    const int kdim = tr ? C_out : C_in;
    const int tile = tq_conv_tile_co(rows);
    const int ncob_pad = ((rows + tile - 1) / tile) * tile / 16;
    if (mode == 2 || mode == 3 || mode == 5) {
        const size_t total2 = (size_t)((kdim + 63) / 64) * K * ncob_pad * 64;
        if (mode == 2) {
            hipLaunchKernelGGL(pack_conv_weight_mx_kernel, dim3((unsigned)((total2 + 255) / 256)), dim3(256), 0, stream, w, C_out,
                               C_in, K, ncob_pad, reinterpret_cast<uint4*>(out));
        } else {
            hipLaunchKernelGGL(pack_conv_weight_mx6_kernel, dim3((unsigned)((total2 + 255) / 256)), dim3(256), 0, stream, w, C_out,
                               C_in, K, ncob_pad, reinterpret_cast<uint4*>(out), mode == 5 ? 1 : 0);
        }
        TQ_CHECK_LAUNCH();
        return 0;
    }
    const int nchunks = (kdim + 31) / 32;
    const size_t total = (size_t)nchunks * K * ncob_pad * 64;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid), dim3(256), 0, stream, w, C_out, C_in, K, mode, rows, kdim,
                       ncob_pad, reinterpret_cast<uint4*>(out));
    TQ_CHECK_LAUNCH();
    return 0;
}
