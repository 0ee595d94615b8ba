// Launch arguments of the fused 1-D convolution kernels (conv1d_mfma.hip: two waves per SIMD; conv1d_w4.hip: one wave per SIMD)
// and the fp6 block-scale helpers both use.
#pragma once
#include "common.hpp"

namespace tq {

struct ConvArgs {
    const float* x0;
    const float* x1;
    const float* gscale;
    const float* gshift;
    const uint4* wpk;
    const float* bias;
    const float* emb;
    const float* res;
    float* y;
    float* stats;
    int B, T_in, T_out, C0, C1, C_out;
    int emb_stride, flags, ncob_pad, nslots;
    uint32_t drop_site;
    uint32_t drop_thresh;  // keep if hash >= thresh
    float drop_scale;      // 1/(1-p)
    uint64_t drop_seed;
    // data-gradient epilogue (EPI == 1): forward inputs / folded GN of the forward conv, split destination
    const float* fx0;
    const float* fx1;
    const float* fgs;
    const float* fgh;
    float* y1;
    int OC0;     // output channels [0, OC0) -> y (row stride OC0), [OC0, C_out) -> y1 (row stride C_out - OC0)
    int bflags;  // TQ_BWD_*
    // fused 1x1 skip convolution (FUSE): extra K chunks read un-activated from the block input, centre tap only
    const float* sx0;
    const float* sx1;
    const float* sbias;
    int sC0, sC1;
    int wfmt;  // TQ_WFMT_*: packed weight format = contraction scheme
    // EPI == 2 (qkv projection at inference): K / V channels go straight to the attention kernel's pre-split planes
    unsigned char* kv;
    int kvH, kvD, kvTp;
    float kvscale;
    int kv_vf16;       // V planes as fp16 hi / lo (the attention kernel's one-fp16-p form) instead of bf16 hi / lo
    int* range_flag;  // see TqConvDesc.range_flag
    int t_tile;       // 0: the default tiles (128 / 256 positions per workgroup); 32: the small tile (TqConvDesc.t_tile)
    const uint32_t* in_amax;  // data gradient, TQ_WFMT_F16_MX6: bit pattern of max|dy| over the whole tensor (see TqConvBwdDesc.dy_amax)
    // fused GroupNorm finalisation (TqConvDesc.gn_fuse): the workgroup that completes a sample's statistics folds them
    unsigned long long* gf_counters;   // nullptr: off
    const float* gf_partner;
    int gf_Cp, gf_partner_first, gf_narrive;
    const float* gf_gamma;
    const float* gf_beta;
    float* gf_gscale;
    float* gf_gshift;
    float* gf_mean_rstd;
    // consumer-side GroupNorm fold (TqConvDesc.gn_fold): statistics of the two sources, slots per source, affine parameters; cf_st0 == nullptr: off
    const float* cf_st0;
    const float* cf_st1;
    int cf_ns0, cf_ns1;
    const float* cf_gamma;
    const float* cf_beta;
    float* cf_mean_rstd;
    int exp_stagger;   // experiment builds (-DTQ_EXP_STAGGER): s_sleep units by which the workgroups of a launch start apart; else 0, unread
};


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));

// hipcc (ROCm 7.2) lets the destination of v_cvt_scalef32_2xpk16_fp6_f32 overlap its scale / source registers and the instruction
// does not read everything before it writes (tools/micro/fp6_scheme_probe.hip): the early-clobber output keeps them apart.
// out[2i] = fp6(a[i] / scale), out[2i + 1] = fp6(b[i] / scale), e2m3, round to nearest, saturating.
__device__ __forceinline__ u32x6 cvt_2xpk16_fp6(const f32x16& a, const f32x16& b, float scale) {
    u32x6 out;
    asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(out) : "v"(a), "v"(b), "v"(scale));
    return out;
}
// biased E8M0 exponent of the smallest power of two 2^e with mx / 2^e <= 7.5 (the e2m3 maximum); >= 13 so that "- 12" stays valid
__device__ __forceinline__ unsigned e8m0_block_scale(float mx) {
    const unsigned b = (__float_as_uint(mx * (1.0f / 7.5f)) + 0x7FFFFFu) >> 23;
    return b < 13u ? 13u : (b > 254u ? 254u : b);
}


// Entry points of the translation units that instantiate conv1d_kernel.hpp (split so that the library builds in parallel); each
// returns 0, a TQ_ERR_* code or a hipError_t like the C ABI.
int conv_launch_fwd_k1(const ConvArgs& a, hipStream_t stream);
int conv_launch_fwd_k3(const ConvArgs& a, hipStream_t stream);
int conv_launch_fwd_k5(const ConvArgs& a, hipStream_t stream);    // conv1d_fwd_k5a.hip (hands dropout / fused-skip launches to k5b)
int conv_launch_fwd_k5b(const ConvArgs& a, hipStream_t stream);
int conv_launch_qkv(const ConvArgs& a, hipStream_t stream);
int conv_launch_resample(const ConvArgs& a, int ktaps, int stride, hipStream_t stream);
int conv_launch_dgrad(const ConvArgs& a, int ktaps, hipStream_t stream);

// conv1d_w4.hip (TQDNE_BUILD_EXPERIMENTS builds only): the one-wave-per-SIMD variant of the stride-1 forward launches in the fp16 +
// MX-fp6 scheme.  Returns TQ_ERR_SHAPE (nothing launched) for a launch it is not built for; the caller then takes the
// two-waves-per-SIMD kernel.
int conv1d_w4_launch(const ConvArgs& a, int ktaps, hipStream_t stream);

}  // namespace tq
