/* tqdne_hip.h -- C ABI of libtqdne_hip.so: the MI355X (gfx950) kernels behind the tqdne 1-D EDM hot path.
 *
 * The reference (highfem/tqdne) has no FFI / plugin interface: its hot path is ordinary PyTorch modules
 * (SURVEY.md section 8b).  These entry points are what a maintainer of the reference would bind (ctypes, see
 * INTEGRATION.md) to replace the ATen op chains cited on each function.  Conventions:
 *   - device pointers only, caller-owned memory, nothing allocated or synchronised inside;
 *   - all work is enqueued on the given HIP stream (graph-capture safe);
 *   - return 0 on success, a TQ_ERR_* code (negative) for bad arguments, or a hipError_t (>0) from the launch;
 *   - re-entrant, no mutable global state: safe for one process per GPU and for several streams.
 * Layouts: activations (B, T, C) fp32, C contiguous; network input/output (B, C, T) fp32 as in the reference;
 * per-channel partial statistics (B, nslots, C, 2) fp32 with nslots = ceil(T / 128) and {sum, sum of squares};
 * GroupNorm folded to per-(b, c) scale/shift (B, C) fp32:  norm(x) = scale * x + shift.
 */
#ifndef TQDNE_HIP_H
#define TQDNE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

#define TQ_ABI_VERSION 7

#define TQ_ERR_ARG (-1)   /* null / inconsistent pointer arguments */
#define TQ_ERR_SHAPE (-2) /* unsupported shape */

/* flags of TqConvDesc.flags */
#define TQ_CONV_GN 1       /* apply gscale/gshift (folded GroupNorm32) to the input */
#define TQ_CONV_SILU 2     /* apply SiLU to the (normalised) input */
#define TQ_CONV_EMB 4      /* add emb[b, co] to the output (ResBlock time embedding, unet.py:141) */
#define TQ_CONV_RES 8      /* add res[b, t, co] to the output (residual, unet.py:143 / blocks.py:145) */
#define TQ_CONV_STATS 16   /* emit per-channel partial statistics of the output */
#define TQ_CONV_DROPOUT 32 /* training-mode dropout on the activated input (unet.py:101) */
/* TQ_CONV_POLY2 (tq_conv1d_fwd, stride 1, ktaps 3, no upsample): the launch is BOTH phases of "nearest x2 upsampling, then conv k = 5"
 * (Upsample.forward, blocks.py:56-66) written as one k = 3 conv over the un-upsampled input -- even outputs see the taps
 * (w0+w1, w2+w3, w4), odd outputs (w0, w1+w2, w3+w4), 3/5 of the multiply-adds.  C_out = 2*C: output channel block [p*C, (p+1)*C)
 * is phase p and is stored to row 2t+p of a (B, 2*T_out, C) tensor; bias / emb / res are indexed by the real channel / row;
 * statistics: 2*ceil(T_out/128) slots of C channels (slot = 2*tile + p; with TQ_CONV_STATS T_out must be a multiple of 128 or leave
 * more than 64 rows in its last tile, so that this equals the ceil(2*T_out/128) slots of the (B, 2*T_out, C) tensor). */
#define TQ_CONV_POLY2 64
/* TQ_CONV_CH_TILES (round 6): a hint of launch-bound plans (a grid far below the chip): where a conv has an input-stationary form with one
 * workgroup per position tile (the qkv projection: C_out >= 512 over 128 / 256 input channels) take the channel-tiled form instead --
 * C_out / 256 times the workgroups (a 16-sample plan alone on the device at T = 512: 64 -> 192; 37 -> 21 us).  Same numbers.  With the
 * device full (B = 64, or four 16-sample lanes) the input-stationary form is the faster one (+0.7 % of a sample with this flag). */
#define TQ_CONV_CH_TILES 128

/* TqConvDesc.wfmt: how the fp32 product x * w is contracted on the matrix cores (= format of the packed weights).
 * BF16X3: both operands split into bf16 hi + lo, three bf16 MFMA products; fp32 range, ~2^-16 relative (pack modes 0 / 1).
 * F16_MX8: per 64 channels two fp16 MFMAs plus one block-scaled fp8 MFMA carrying both first-order corrections; ~2^-15 relative at
 * 2/3 of the MFMA cycles; the activation operand has fp16 RANGE (|x| > 65504 overflows to inf / NaN in the output, relative precision lost below 6e-5).  Built
 * for stride-1 forward launches (tq_conv1d_fwd, tq_conv1d_fwd_skip, tq_conv1d_fwd_qkv) with 128 | C_out and 64 | every source's
 * channels (pack mode 2). */
#define TQ_WFMT_BF16X3 0
#define TQ_WFMT_F16_MX8 1
/* F16_MX6: as F16_MX8 with e2m3 (fp6) correction operands and per-lane E8M0 block scales (one per 16 channels): 12 instead of 16
 * MFMA passes per 64 channels and corrections that do not clamp; same shapes, same fp16 RANGE of the activation operand (pack
 * mode 3).  Round 6: additionally 64 | C_out for the k = 5 launches with a TQ_CONV_GN | TQ_CONV_SILU prologue (tq_conv1d_fwd,
 * tq_conv1d_fwd_skip: the 64-channel ResBlock convs; tile of 64 channels x 128 positions). */
#define TQ_WFMT_F16_MX6 2

/* EXPERIMENT (only in builds with TQ_BUILD_EXPERIMENTS_BIT; otherwise TqConvDesc.gn_fuse is reserved and must be NULL: a non-NULL
 * value returns TQ_ERR_ARG).  GroupNorm finalisation fused into the launch that completes a tensor's statistics (ABI 3).
 * With TQ_CONV_STATS the launch's workgroups publish their partial sums, take an arrival ticket per sample, and the workgroup whose
 * ticket completes sample b folds the statistics of that sample into the scale / shift of the GroupNorm that CONSUMES the produced
 * tensor (what tq_gn_finalize would do in a launch of its own: bit-identical coefficients).  The consumer may normalise the
 * concatenation of the produced tensor with a second, older one (`partner_stats`, complete before this launch).
 * `counters`: device, one uint64 per sample, zero-initialised ONCE by the caller and owned by this (launch site, consumer) pair:
 * tickets count up monotonically, nothing is reset between launches.  Launches using the same TqGnFuse must not overlap. */
typedef struct TqGnFuse {
    unsigned long long* counters; /* (B) arrival tickets */
    const float* partner_stats;   /* (B, nslots, C_partner, 2) or NULL */
    int32_t C_partner;
    int32_t partner_first;        /* 1: the consumer's channels are [partner | produced], 0: [produced | partner] */
    const float* gamma;           /* (C_total) affine parameters of the consuming GroupNorm */
    const float* beta;
    float* gscale;                /* (B, C_total) out */
    float* gshift;                /* (B, C_total) out */
    float* mean_rstd;             /* (B, 32, 2) out, nullable */
} TqGnFuse;

/* ABI 7.  Consumer-side GroupNorm fold (TqConvDesc.gn_fold; optional, NULL = off): the launch forms the folded coefficients of ITS OWN
 * prologue (TQ_CONV_GN) from the partial statistics of its (up to two, concatenated) source tensors -- every workgroup folds its sample,
 * the arithmetic of tq_gn_finalize, bit-identical coefficients -- and WRITES them to the gscale / gshift arguments of the call (and
 * mean_rstd), where the weight / data gradients of the same conv read them later.  Replaces the tq_gn_finalize launch in front of the
 * conv in launch-bound plans.  Built for the small tile (t_tile = 32: tq_conv1d_fwd, tq_conv1d_fwd_skip); other launches return TQ_ERR_SHAPE.
 * Host pointer, read at launch. */
typedef struct TqGnFold {
    const float* stats0;   /* (B, ceil(T_in / slot0), C_in0, 2) partial statistics of source 0 */
    const float* stats1;   /* (B, ceil(T_in / slot1), C_in1, 2) or NULL */
    int32_t slot0, slot1;  /* positions per statistics slot of each source: 128 (0 = 128) or 32 */
    const float* gamma;    /* (C_in0 + C_in1) affine parameters of the GroupNorm */
    const float* beta;
    float* mean_rstd;      /* (B, 32, 2) out, nullable */
} TqGnFold;

typedef struct TqConvDesc {
    int32_t B, T_in, T_out;
    int32_t C_in0, C_in1; /* channels of the two concatenated sources (C_in1 = 0: single source) */
    int32_t C_out;
    int32_t ktaps, stride, pad;
    int32_t upsample; /* 1: input is read through nearest x2 upsampling (T_out = 2*T_in) */
    int32_t flags;
    int32_t emb_stride; /* floats between consecutive samples in emb */
    uint32_t dropout_site;
    float dropout_p;
    uint64_t dropout_seed;
    int32_t C_skip0, C_skip1; /* tq_conv1d_fwd_skip only: channels of the fused 1x1 skip conv's (concatenated) input; else 0 */
    int32_t wfmt;             /* TQ_WFMT_*: format of packed_w = contraction scheme of this launch */
    /* Range guard of the fp16-range scheme (optional, NULL = off): with TQ_CONV_STATS the epilogue sets *range_flag = 1 when a
     * channel's sum of squares over one 128-position slot reaches (65504 / 2)^2, i.e. when max|y| of the written tensor MAY exceed
     * half of the fp16 range (max|y| <= sqrt(sum y^2)), or is not finite.  The caller polls the flag and moves the launches that
     * read such a tensor un-normalised (fused 1x1 skip convs, up-sampling convs) to TQ_WFMT_BF16X3.  Device pointer, never reset
     * by the library. */
    int32_t* range_flag;
    const TqGnFuse* gn_fuse; /* reserved, NULL (experiment builds: host pointer, read at launch; see TqGnFuse) */
    /* ABI 5.  Positions per workgroup along T: 0 = the default tiles (128 / 256); 32 = the small tile for launch-bound batches (a
     * launch of the default tiling with far fewer workgroups than compute units: four times the workgroups, a quarter of the work
     * each).  Built for tq_conv1d_fwd / tq_conv1d_fwd_skip with ktaps 5, stride 1, no upsampling, TQ_CONV_GN | TQ_CONV_SILU, in
     * TQ_WFMT_BF16X3 and TQ_WFMT_F16_MX6 (128 | C_out); other launches return TQ_ERR_SHAPE.  With TQ_CONV_STATS the partial
     * statistics then have one slot per 32 positions: stats_partial is (B, ceil(T_out / 32), C, 2), and tq_gn_finalize must be told
     * (its slot arguments).  Convolution results are bit-identical to the default tiles'; the statistics are the same sums in another
     * association order. */
    int32_t t_tile;
    int32_t reserved2;
    const TqGnFold* gn_fold; /* ABI 7: consumer-side GroupNorm fold (see TqGnFold), NULL = off */
} TqConvDesc;

/* flags of TqConvBwdDesc.flags: which stages the FORWARD conv applied to its input */
#define TQ_BWD_GN 1       /* forward input was GroupNorm-folded (gscale/gshift given) */
#define TQ_BWD_SILU 2     /* forward applied SiLU: multiply by silu'(gscale*x+gshift) */
#define TQ_BWD_DROPOUT 4  /* forward applied dropout: re-generate the same mask */
#define TQ_BWD_ACCUM 8    /* add to dx instead of overwriting (tensor consumed by several ops) */
#define TQ_BWD_STATS 16   /* emit per-channel partial sums {sum g, sum g*x} for the GroupNorm backward */

typedef struct TqConvBwdDesc {
    int32_t B, T;
    int32_t C_dy;         /* channels of the incoming gradient (= forward C_out) */
    int32_t C_dx0, C_dx1; /* channel split of the produced gradient (= forward C_in0, C_in1) */
    int32_t ktaps;        /* stride-1 "same" convs only; strided / upsampled ones are composed with tq_zero_stuff / tq_pair_sum */
    int32_t flags;
    uint32_t dropout_site;
    float dropout_p;
    uint64_t dropout_seed;
    /* ABI 5.  Contraction scheme of the data gradient: TQ_WFMT_BF16X3 (0: fp32 range, packed_w_t from pack mode 1) or
     * TQ_WFMT_F16_MX6 (packed_w_t from pack mode 5; 64 | C_dy and 64 | C_dx0 + C_dx1): dy is staged times the exact power of two
     * that brings max|dy| into [2^13, 2^14) -- gradients are far below fp16's normal range otherwise -- and the accumulators are
     * multiplied by its inverse, so the result has the accuracy of the forward scheme (~2^-15 relative) at half the MFMA cycles of
     * bf16x3.  dy_amax: DEVICE pointer to a TQ_AMAX_WORDS block as tq_colsum(..., amax_out) / tq_gn_bwd_apply_colsum fill it for the same
     * dy (the maximum over its TQ_AMAX_WAYS words = IEEE bit pattern of max|dy|; any upper bound within a factor 8 will do); required for
     * TQ_WFMT_F16_MX6. */
    int32_t wfmt;
    int32_t reserved;
    const uint32_t* dy_amax;
} TqConvBwdDesc;

int tq_abi_version(void);
/* Bit mask of optional parts compiled into this library.  TQ_BUILD_EXPERIMENTS_BIT: the opt-in kernels that lost their A/B
 * (one-wave-per-SIMD conv, slim 64-channel tile, in-launch GroupNorm fold = TqConvDesc.gn_fuse) are present; the default build
 * (`python -m tqdne_amd._build`) leaves them out, `TQDNE_BUILD_EXPERIMENTS=1` builds them into libtqdne_hip_exp.so. */
#define TQ_BUILD_EXPERIMENTS_BIT 1
int tq_build_flags(void);

/* ---- weights -------------------------------------------------------------------------------------------- */
/* Pack a torch Conv1d weight (C_out, C_in, K) fp32 into per-lane MFMA fragments.
 * mode 0: forward operand, TQ_WFMT_BF16X3; mode 1: data-gradient operand (transposed + tap-flipped), TQ_WFMT_BF16X3;
 * mode 2: forward operand, TQ_WFMT_F16_MX8; mode 3: forward operand, TQ_WFMT_F16_MX6; mode 5 (ABI 5): data-gradient operand,
 * TQ_WFMT_F16_MX6.  (mode 4 is a plain copy job of tq_pack_jobs.) */
size_t tq_conv_weight_pack_bytes(int C_out, int C_in, int K, int mode);
int tq_pack_conv_weight(const float* w, int C_out, int C_in, int K, int mode, void* packed, hipStream_t stream);
int tq_conv_tile_co(int C_out);
/* All (re)packs of a plan in one launch (what a training step re-does after every optimizer update: torch parameters ->
 * forward / transposed fragments, plus the gather of the ResBlocks' embedding projections into their concatenated buffers,
 * unet.py:91-97).  jobs: a DEVICE array, block_begin = running sum of tq_pack_job_blocks over the preceding jobs.
 * mode 0..3 as tq_pack_conv_weight; mode 4: copy C_out floats src -> dst. */
typedef struct TqPackJob {
    const void* src;
    void* dst;
    int32_t C_out, C_in, K, mode;
    int32_t block_begin;
    int32_t reserved;
} TqPackJob;
int tq_pack_job_blocks(int C_out, int C_in, int K, int mode);
int tq_pack_jobs(const TqPackJob* jobs_device, int njobs, int total_blocks, hipStream_t stream);

/* ---- fused convolution ---------------------------------------------------------------------------------- */
/* Replaces GroupNorm32 -> SiLU -> Dropout -> Conv1d(k in {1,3,5}) -> +emb -> +residual, the channel concat and the
 * nearest-x2 upsample of tqdne/unet.py:86-102,131-143,396 and tqdne/blocks.py:56-66,92-101,127-145.
 * Constraints: C_in0, C_in1, C_out multiples of 32; stride 1 ("same" padding, pad = k/2) or stride 2 (k=3, pad=1). */
int tq_conv1d_fwd(const TqConvDesc* desc, const float* x0, const float* x1, const float* gscale, const float* gshift,
                  const void* packed_w, const float* bias, const float* emb, const float* residual, float* y,
                  float* stats_partial, hipStream_t stream);

/* Same, with the ResBlock's 1x1 skip convolution fused in (unet.py:112,143): y = conv_k(f(x)) + W_skip * x_skip + biases.
 * packed_w holds tq_pack_conv_weight(main, mode 0) immediately followed by tq_pack_conv_weight(skip 1x1, mode 0).
 * Built for k = 5 with GN + SiLU (+ dropout) prologues. */
int tq_conv1d_fwd_skip(const TqConvDesc* desc, const float* x0, const float* x1, const float* gscale, const float* gshift,
                       const void* packed_w_main_then_skip, const float* bias, const float* emb, const float* skip_x0,
                       const float* skip_x1, const float* skip_bias, float* y, float* stats_partial, hipStream_t stream);

/* Inference form of the AttentionBlock's qkv projection (blocks.py:127-145): 1x1 conv of GN(x) whose K and V output channels are
 * written directly as the pre-split planes tq_attention_fwd_presplit streams (K pre-scaled by D^-1/4: bf16 hi / lo; V in
 * `v_format`, which both calls of the pair must be given alike), q as fp32 into qkv (B, T, 3 H D) (its K / V part is left
 * untouched).  kv_planes: tq_attention_workspace_bytes(B, T, H, D) bytes whose rows t >= T (padding to a multiple of 64) must be
 * zero.  desc: ktaps 1, single source, C_out = 3 H D, flags none or TQ_CONV_GN.
 * v_format (ABI 6; was the environment variable TQDNE_ATTN_VF16 read inside the library):
 *   TQ_KV_V_F16: V as fp16 hi / lo planes, which the attention kernel multiplies by ONE fp16 softmax weight -- two products instead
 *     of three, 1.4e-4 of the output scale.  fp16 RANGE: with desc->range_flag set the epilogue raises it when |v| reaches half of
 *     the fp16 range (or is not finite), exactly like the fp16-range conv schemes; the caller repeats with TQ_KV_V_BF16;
 *   TQ_KV_V_BF16: bf16 hi / lo V, three products, fp32 range. */
#define TQ_KV_V_BF16 0
#define TQ_KV_V_F16 1
int tq_conv1d_fwd_qkv(const TqConvDesc* desc, const float* x, const float* gscale, const float* gshift, const void* packed_w,
                      const float* bias, float* qkv, void* kv_planes, int H, int D, int v_format, hipStream_t stream);

/* Data gradient of tq_conv1d_fwd (stride 1): g = (W^T * dy) chained through the forward prologue (dropout, SiLU,
 * folded GN scale); packed_w_t from tq_pack_conv_weight(mode 1).  x0/x1/gscale/gshift are the FORWARD conv's inputs.
 * Autograd counterpart of the ATen conv/SiLU/dropout backward chain Lightning runs for edm.py:136 training_step. */
int tq_conv1d_bwd_data(const TqConvBwdDesc* desc, const float* dy, const void* packed_w_t, const float* x0, const float* x1,
                       const float* gscale, const float* gshift, float* dx0, float* dx1, float* gstats_partial,
                       hipStream_t stream);

/* Weight gradient of tq_conv1d_fwd: dw (C_out, C_in, K) fp32 (overwritten) = sum_{b,t} dy * xhat, xhat recomputed from the
 * FORWARD conv's inputs exactly as the forward prologue does (same desc, incl. dropout seed).  workspace: >=
 * tq_conv1d_bwd_weight_workspace(desc) bytes of scratch (partial slabs of the (b,t) splits). */
size_t tq_conv1d_bwd_weight_workspace(const TqConvDesc* desc);
int tq_conv1d_bwd_weight(const TqConvDesc* desc, const float* dy, const float* x0, const float* x1, const float* gscale,
                         const float* gshift, float* dw, void* workspace, size_t ws_bytes, hipStream_t stream);
/* Same, with the column sums of dy fused in (the bias gradient and, per sample, the gradient of the broadcast time embedding,
 * unet.py:141): colsum_bc[b * bc_stride + co] += sum_t dy[b, t, co], colsum_c[co] (and colsum_c2[co]) += sum_{b,t} dy -- each
 * nullable, accumulated with atomics into buffers the caller has zeroed.  Saves the separate pass of tq_colsum over dy. */
int tq_conv1d_bwd_weight_colsum(const TqConvDesc* desc, const float* dy, const float* x0, const float* x1, const float* gscale,
                                const float* gshift, float* dw, void* workspace, size_t ws_bytes, float* colsum_bc, int bc_stride,
                                float* colsum_c, float* colsum_c2, hipStream_t stream);

/* First conv of the network: (B, C_in<=16, T) fp32 input, scaled per sample by in_scale[b] (EDM c_in, edm.py:107;
 * NULL = 1), k taps "same" -> (B, T, C_out) channels-last + bias (+ partial statistics).  unet.py:233. */
int tq_stem_conv_fwd(const float* x_nct, const float* in_scale, const float* w, const float* bias, float* y,
                     float* stats_partial, int B, int C_in, int T, int C_out, int ktaps, hipStream_t stream);

/* Last conv: GroupNorm32+SiLU (folded scale/shift) -> conv k "same" to C_out<=16 -> (B, C_out, T) output,
 * then out = c_out[b] * conv + c_skip[b] * skip_src[b, co, t]  (EDM / consistency preconditioning, edm.py:111-113,
 * consistency_model.py:78); c_out/c_skip/skip_src NULL = plain conv output.  unet.py:355-357,398. */
/* LDS bytes tq_head_conv_fwd needs for a shape; 0 = unsupported shape (16 | C_in <= 128, C_out <= 16, k in {1, 3, 5}, <= 64 KB). */
size_t tq_head_conv_lds_bytes(int C_in, int C_out, int ktaps);
int tq_head_conv_fwd(const float* x, const float* gscale, const float* gshift, const float* w, const float* bias,
                     const float* c_out, const float* c_skip, const float* skip_src, float* y_nct, int B, int T, int C_in,
                     int C_out, int ktaps, hipStream_t stream);

/* ---- GroupNorm32 statistics -> folded scale/shift ---------------------------------------------------------- */
/* The normalised tensor is the channel concat of up to two sources whose per-channel partial statistics were
 * emitted by their producers.  Writes scale/shift (B, C0+C1) and mean/rstd (B, 32, 2).  nn.py:11-13,90-105. */
/* slot0 / slot1 (ABI 5): positions per statistics slot of each source -- 128 (0 = default) or 32 (a tensor written by a
 * TqConvDesc.t_tile = 32 launch); stats_i is (B, ceil(T / slot_i), C_i, 2). */
int tq_gn_finalize(const float* stats0, int C0, const float* stats1, int C1, int B, int T, const float* gamma,
                   const float* beta, float* gscale, float* gshift, float* mean_rstd, int slot0, int slot1, hipStream_t stream);

/* GroupNorm32 backward, step 1: per-channel partial sums {sum g, sum g*x} (from tq_conv1d_bwd_data / tq_head_conv_bwd)
 * + saved mean/rstd + gamma -> coefficients with dx = A*g + Bc*x + Cc, and dgamma/dbeta (atomically added: zero them first). */
int tq_gn_bwd_finalize(const float* gstats_partial, const float* mean_rstd, const float* gamma, int B, int C, int T,
                       float* coef_a, float* coef_b, float* coef_c, float* dgamma, float* dbeta, hipStream_t stream);
/* step 2, per concat source (channels [c_offset, c_offset+C_src) of the coefficients):
 * dx (+)= A*g + Bc*x + Cc (+ r);  r = gradient arriving over the residual path (NULL: none). */
int tq_gn_bwd_apply(const float* g, const float* x, const float* r, const float* coef_a, const float* coef_b,
                    const float* coef_c, float* dx, int B, int T, int C_src, int C_total, int c_offset, int accumulate,
                    hipStream_t stream);
/* step 2 with the column sums of the tensor it writes fused in (ABI 5): what tq_colsum(dx, ...) would add to colsum_bc / colsum_c /
 * colsum_c2 / amax_out afterwards, from the values while they are in registers (`dx` must then be COMPLETE after this launch: the
 * last writer of an accumulated gradient).  All four outputs NULL = tq_gn_bwd_apply. */
int tq_gn_bwd_apply_colsum(const float* g, const float* x, const float* r, const float* coef_a, const float* coef_b,
                           const float* coef_c, float* dx, int B, int T, int C_src, int C_total, int c_offset, int accumulate,
                           float* colsum_bc, int bc_stride, float* colsum_c, float* colsum_c2, uint32_t* amax_out,
                           hipStream_t stream);
/* out_bc[b*bc_stride + c] += bscale[b] * sum_t dy[b,t,c];  out_c[c], out_c2[c] += sum_{b,t} (...)  (each optional; bias and
 * embedding gradients: two biases fed by the same tensor are served by one pass) */
/* Small fp32 GEMMs of the embedding-MLP backward (unet.py:91-97,210-227,383-388; the autograd of blocks.py:15-26): one launch runs
 * a list of independent products  C (M x N) = A (M x K) * f(B) (K x N) [* silu'(U)]  with strided operands
 * A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn] (a stride of 1 marks the contiguous direction), f = SiLU when pre_b,
 * U (M x N, row stride ldu) optional.  The job table lives in device memory; tile_begin = running sum of tq_gemm_tiles(M, N). */
typedef struct TqGemmJob {
    const float* A;
    const float* B;
    float* C;
    const float* U;
    int32_t M, N, K;
    int32_t sam, sak, sbk, sbn, ldc, ldu;
    int32_t pre_b;
    int32_t tile_begin;
} TqGemmJob;
int tq_gemm_tiles(int M, int N);
int tq_gemm_f32_jobs(const TqGemmJob* jobs_device, int njobs, int total_tiles, hipStream_t stream);
/* GaussianFourierProjection features (blocks.py:15-26): out (B, 2 half) = [sin(2 pi t W) | cos(2 pi t W)] */
int tq_fourier_features(const float* t, const float* W, float* out, int B, int half, hipStream_t stream);

/* amax_out (ABI 5, nullable): a block of TQ_AMAX_WORDS uint32 that receives the bit pattern of max|dy| by atomic max -- zero it first.
 * The maximum is SPREAD over TQ_AMAX_WAYS words TQ_AMAX_STRIDE apart (one per 128-byte line; a workgroup updates word (its index mod
 * TQ_AMAX_WAYS): atomics on one address serialise at ~11 ns each); the maximum of those words is max|dy|.  Non-negative floats order
 * like their bit patterns; a NaN in dy is recorded as +inf, so a poisoned tensor stays visible.  Feeds TqConvBwdDesc.dy_amax. */
#define TQ_AMAX_WAYS 16
#define TQ_AMAX_STRIDE 32
#define TQ_AMAX_WORDS (TQ_AMAX_WAYS * TQ_AMAX_STRIDE)
int tq_colsum(const float* dy, int B, int T, int C, float* out_bc, int bc_stride, float* out_c, float* out_c2,
              const float* bscale, uint32_t* amax_out, hipStream_t stream);
/* gradient plumbing of the strided / upsampled convs: out[b,u,:] = (u even) ? dy[b,u/2,:] : 0 for u < T_in;
 * dx[b,t,:] (+)= d_up[b,2t,:] + d_up[b,2t+1,:] */
int tq_zero_stuff(const float* dy, float* out, int B, int T_out, int T_in, int C, hipStream_t stream);
int tq_pair_sum(const float* d_up, float* dx, int B, int T, int C, int accumulate, hipStream_t stream);
/* ABI 7.  Upsample (blocks.py:56-66) trained in its two-phase k = 3 form (TQ_CONV_POLY2 forward; data gradient = tq_conv1d_bwd_data with
 * ktaps 3 and C_dy = 2 C_out on the output gradient (B, 2T, C_out) read as (B, T, 2 C_out); weight gradient = tq_conv1d_bwd_weight of that
 * k = 3 conv): folds dw2 (2 C_out, C_in, 3) = [d even-phase taps | d odd-phase taps] back onto the conv's five taps, dw (C_out, C_in, 5)
 * (overwritten): dw0 = dA0 + dB0, dw1 = dA0 + dB1, dw2 = dA1 + dB1, dw3 = dA1 + dB2, dw4 = dA2 + dB2. */
int tq_upsample_poly_wgrad_fold(const float* dw2, float* dw, int C_out, int C_in, hipStream_t stream);
/* stem conv weight gradient (atomically added into zeroed dw (C_out, C_in, K)) */
int tq_stem_conv_bwd_weight(const float* dy, const float* x_nct, const float* in_scale, float* dw, int B, int C_in, int T,
                            int C_out, int ktaps, hipStream_t stream);
/* ABI 5: the same two with a scratch buffer of tq_stem_head_bwd_workspace() bytes (own buffer per call site and stream): every
 * workgroup's 960 partial sums go to its row of the scratch and a second small launch adds the rows into dw (db) -- the outputs share
 * 30 cache lines, and atomics on one LINE serialise at ~11 ns each (512 workgroups: 180 us, the whole time of these kernels).
 * workspace NULL or too small: the atomic form.  dw / db are still ADDED to (zero them first). */
size_t tq_stem_head_bwd_workspace(void);
int tq_stem_conv_bwd_weight_ws(const float* dy, const float* x_nct, const float* in_scale, float* dw, int B, int C_in, int T,
                               int C_out, int ktaps, void* workspace, size_t ws_bytes, hipStream_t stream);
int tq_head_conv_bwd_ws(const float* dpred_nct, const float* c_out, const float* x, const float* gscale, const float* gshift,
                        const float* w, float* g_out, float* gstats_partial, float* dw, float* db, int B, int T, int C_in,
                        int C_out, int ktaps, void* workspace, size_t ws_bytes, hipStream_t stream);
/* head conv backward: dF = c_out[b]*dpred; g_out (B,T,C_in) = (W^T*dF)*silu'(gscale*x+gshift) with GN partial sums;
 * dw, db atomically added (zero them first) */
int tq_head_conv_bwd(const float* dpred_nct, const float* c_out, const float* x, const float* gscale, const float* gshift,
                     const float* w, float* g_out, float* gstats_partial, float* dw, float* db, int B, int T, int C_in,
                     int C_out, int ktaps, hipStream_t stream);

/* ---- embeddings ------------------------------------------------------------------------------------------- */
/* emb = time_mlp(fourier(t)) (+ cond_mlp(cond)); writes emb (B, E) and silu(emb) (B, E).  E = 4*mc.
 * blocks.py:22-26, unet.py:210-227,383-388.  cond pointers NULL when the model is unconditioned. */
int tq_embed_fwd(const float* t, const float* cond, const float* fourier_w, const float* w0, const float* b0,
                 const float* w2, const float* b2, const float* cw0, const float* cb0, const float* cw2, const float* cb2,
                 float* emb, float* silu_emb, float* hidden /* (B, 2, E) scratch: pre-activations, kept for backward */,
                 int B, int mc, int ncond, hipStream_t stream);

/* All per-ResBlock projections Linear(SiLU(emb)) (unet.py:91-97) as one GEMM: out (B, N) = silu_emb (B, E) W^T + bias. */
int tq_linear_fwd(const float* x, const float* w, const float* bias, float* out, int B, int E, int N, hipStream_t stream);

/* ---- attention -------------------------------------------------------------------------------------------- */
/* QKVAttention (blocks.py:156-190), qkv (B, T, 3*H*D) channels-last with channel order [q heads | k heads | v heads],
 * q and k each scaled by D^-1/4, softmax over keys in fp32, out (B, T, H*D).  D in {32, 64, 128}.
 * The workspace holds the pre-split K / V planes of the second-generation kernels (D = 32 / 64); for D = 128 (first-generation kernel)
 * it holds the partial rows of the key split that kernel uses where its grid is far below the chip (round 6; without a workspace it
 * never splits). */
size_t tq_attention_workspace_bytes(int B, int T, int H, int D);
int tq_attention_fwd(const float* qkv, float* out, float* lse /* (B,H,T) log-sum-exp per query, NULL at inference */,
                     void* workspace /* tq_attention_workspace_bytes(); NULL selects the workspace-free kernel */, int B, int T,
                     int H, int D, hipStream_t stream);
/* backward of the above (recomputes P from qkv and lse): dqkv (B, T, 3*H*D) from dout (B, T, H*D);
 * delta (B,H,T) is scratch.  Two passes: queries-stationary for dq, keys-stationary for dk/dv (no atomics). */
/* attention core on q from qkv and K / V planes already written by tq_conv1d_fwd_qkv (no log-sum-exp output: inference) */
int tq_attention_fwd_presplit(const float* qkv, const void* kv_planes, float* out, int B, int T, int H, int D, int v_format,
                              hipStream_t stream);
int tq_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* delta, float* dqkv, int B,
                     int T, int H, int D, hipStream_t stream);
/* the same with a scratch buffer of 2 * tq_attention_workspace_bytes(): second-generation kernels for D = 32 / 64 (one prep pass
 * writes bf16 hi / lo planes of Q, K, V, dO and delta; both passes then stream 16-byte copies, P / dS stay in registers); D = 128 or
 * workspace == NULL fall through to tq_attention_bwd.  Replaces the autograd backward of blocks.py:156-190. */
int tq_attention_bwd_ws(const float* qkv, const float* out, const float* dout, const float* lse, float* delta, float* dqkv,
                        void* workspace, int B, int T, int H, int D, hipStream_t stream);
/* ABI 6.  The same with the K / V planes the training forward already wrote: `kv_planes` is the `workspace` argument of the
 * tq_attention_fwd call whose backward this is (D = 32 / 64; it must still hold that call's planes: give every attention block of a
 * training plan a workspace of its own).  The prep pass then forms Q, dO and delta only.  `workspace`: as tq_attention_bwd_ws
 * (its first half stays unused).  Gradients bit-identical to tq_attention_bwd_ws. */
int tq_attention_bwd_ws_kv(const float* qkv, const float* out, const float* dout, const float* lse, float* delta, float* dqkv,
                           void* workspace, const void* kv_planes, int B, int T, int H, int D, hipStream_t stream);

/* ---- EDM / sampler elementwise ------------------------------------------------------------------------------ */
/* per-sample scalars from sigma: c_in, c_out, c_skip, c_noise, loss weight  (edm.py:24-37); sigma_stride 0 = shared */
int tq_edm_scalars(const float* sigma, int sigma_stride, float sigma_data, float* c_in, float* c_out, float* c_skip,
                   float* c_noise, float* lweight, int B, hipStream_t stream);
/* consistency preconditioning scalars (consistency_model.py:69-74) */
int tq_cm_scalars(const float* sigma, int sigma_stride, float sigma_data, float sigma_min, float* c_out, float* c_skip,
                  int B, hipStream_t stream);
/* training noise injection: sigma = exp(eps*P_std + P_mean); x = y + sigma * n   (edm.py:126-129) */
int tq_edm_noise_inject(const float* y, const float* unit_noise, const float* eps, float P_mean, float P_std,
                        float* sigma, float* x_noisy, int B, int n_per_sample, hipStream_t stream);
/* loss = mean(lweight[b] * (pred - y)^2); also d loss / d pred (edm.py:131-134).  loss_out: 1 float (zeroed inside). */
int tq_edm_loss(const float* pred, const float* y, const float* lweight, float* loss_out, float* dpred, int B,
                int n_per_sample, hipStream_t stream);
/* Heun sampler state updates, fp64 state / fp32 network output (edm.py:182-194).
 * euler:   d = (x - D)/s;  x_next = x + d * (float)(s_next - s);  x32 = (float)x_next
 * correct: d' = (x_next - D')/s_next;  x_new = x + (float)(s_next - s) * (0.5 d + 0.5 d');  x32 = (float)x_new */
int tq_heun_euler(const double* x, const float* denoised, const float* sigma, const float* sigma_next, double* d_cur,
                  double* x_next, float* x32, size_t n, hipStream_t stream);
int tq_heun_correct(const double* x, const double* x_next, const float* denoised_next, const double* d_cur,
                    const float* sigma, const float* sigma_next, double* x_out, float* x32, size_t n, hipStream_t stream);
/* x64 = unit_noise64 * sigma0; x32 = (float)x64   (edm.py:160) */
int tq_sampler_init(const double* unit_noise, const float* sigma0, double* x, float* x32, size_t n, hipStream_t stream);

/* Stochastic (churned) sampler, edm.py:198-230: the temporary noise increase x_hat = x + (n * S_noise) * c (lines 205-208), c =
 * sqrt(sigma_hat^2 - sigma^2) formed by the caller in fp32 as the reference does (device scalar).  The Euler / correction steps are
 * tq_heun_euler / tq_heun_correct with sigma_hat in place of sigma (lines 217-228). */
int tq_heun_churn(const double* x, const double* unit_noise, const float* coef, double s_noise, double* x_hat, float* x32,
                  size_t n, hipStream_t stream);

/* out[b] = x[b] + noise[b] * sigma[b]: the two noised copies of the iCT training step (consistency_model.py:150-160). */
int tq_axpy_sigma(const float* x, const float* noise, const float* sigma, float* out, int B, int n_per_sample,
                  hipStream_t stream);

/* ---- DDPM (tqdne/diffusion.py:55-109; ABI 3).  The reference delegates this arithmetic to diffusers' DDPMScheduler, which is not in
 * its lockfile: restated from the published algorithm (Ho et al. 2020), see tqdne_amd/diffusion.py. ------------------------------- */
/* out[b] = a[b] x[b] + c[b] y[b]: the forward process of the training step (noise_scheduler.add_noise, diffusion.py:98). */
int tq_scale_add2(const float* x, const float* y, const float* a, const float* c, float* out, int B, int n_per_sample,
                  hipStream_t stream);
/* One ancestral sampling step (noise_scheduler.step, diffusion.py:77): x0 = (x - sqrt(1 - abar_t) model_out) / sqrt(abar_t) for an
 * epsilon-predicting network, else model_out; clipped to +-clip when clip > 0; out = coef_x0 x0 + coef_xt x + sigma noise
 * (noise nullable: the last step adds none). */
int tq_ddpm_step(const float* x, const float* model_out, const float* noise, float* out, size_t n, int epsilon_prediction,
                 double sqrt_one_minus_abar, double inv_sqrt_abar, double clip, double coef_x0, double coef_xt, double sigma,
                 hipStream_t stream);

/* Weighted pseudo-Huber distance of improved consistency training (consistency_model.py:163-173):
 * loss = mean(w_b (sqrt((pred - target)^2 + c^2) - c)); dpred (nullable) = d loss / d pred.  loss_out is overwritten. */
int tq_pseudo_huber_loss(const float* pred, const float* target, const float* weight, float c, float* loss_out, float* dpred,
                         int B, int n_per_sample, hipStream_t stream);

/* Mean squared error (autoencoder.py:61-63): loss_out = mean((a - b)^2) (overwritten), d (nullable) = 2 (a - b) / n. */
int tq_mse_loss(const float* a, const float* b, float* loss_out, float* d, size_t n, hipStream_t stream);

/* VAE bottleneck (autoencoder.py:37-43,64-66).  enc (B, 2L, T) = [mean | log_std]; eps, z, dz (B, L, T).
 * fwd: z = mean + eps exp(log_std); kl_out (nullable, overwritten) = mean over (b, t) of 0.5 sum_c (mean^2 + std^2 - 2 log_std - 1).
 * bwd: denc (B, 2L, T) = [dz + kw mean | dz eps std + kw (std^2 - 1)], kw = kl_weight / (B T). */
int tq_vae_reparam_fwd(const float* enc, const float* eps, float* z, float* kl_out, int B, int L, int T, hipStream_t stream);
int tq_vae_reparam_bwd(const float* enc, const float* eps, const float* dz, float* denc, float kl_weight, int B, int L, int T,
                       hipStream_t stream);

/* Stem input of a signal-conditioned model (edm.py:108-109): out (B, C0 + C1, T) = [x * scale[b] | cond_signal]; scale nullable. */
int tq_concat_scale(const float* x, const float* scale, const float* cond_signal, float* out, int B, int C0, int C1, int T,
                    hipStream_t stream);

/* ---- optimizer (edm.py:240-251, ema.py:24-28) ------------------------------------------------------------- */
/* One launch for the whole model: torch.optim.Adam's update (no weight decay, no amsgrad) on every chunk of the table,
 *   g' = g * grad_scale;  m += (1 - beta1) (g' - m);  v = beta2 v + (1 - beta2) g'^2;
 *   p -= step_size * m / (sqrt(v) * inv_bias2_sqrt + eps)       step_size = lr / (1 - beta1^t), inv_bias2_sqrt = 1/sqrt(1 - beta2^t)
 * preceded by p *= decay_factor (torch.optim.AdamW's decoupled weight decay 1 - lr * wd, autoencoder.py:93-95; 1.0 = Adam) and
 * followed, where ema != NULL, by the EMA callback's lerp  ema += ema_weight * (p - ema)  (ema_weight = 1 - decay).
 * The scalars are doubles (host-side values as torch computes them: 1 - beta in double) and are rounded to fp32 once.
 * `chunks` is a DEVICE array; a chunk is <= TQ_ADAM_CHUNK consecutive elements of one tensor, its pointers 16-byte aligned. */
#define TQ_ADAM_CHUNK 4096
typedef struct TqAdamChunk {
    float* p;
    const float* g;
    float* m;
    float* v;
    float* ema; /* NULL: no EMA for this chunk */
    int32_t n;
    int32_t reserved;
} TqAdamChunk;
int tq_adam_ema_step(const TqAdamChunk* chunks, int n_chunks, double step_size, double beta1, double beta2, double eps,
                     double inv_bias2_sqrt, double ema_weight, double grad_scale, double decay_factor, hipStream_t stream);
/* ABI 3: the same update behind a DEVICE-side predicate: when `*skip_flag != 0` at execution time the launch leaves parameters,
 * moments and EMA untouched.  The data-parallel trainer passes the range-guard flag of the fp16-range forward scheme
 * (TqConvDesc.range_flag, max-reduced over the ranks): a step whose forward came close to the fp16 range is dropped on the device,
 * without a host synchronisation, before the host has seen the flag and moved the plan to bf16x3.  NULL: unconditional. */
int tq_adam_ema_step_guarded(const TqAdamChunk* chunks, int n_chunks, double step_size, double beta1, double beta2, double eps,
                             double inv_bias2_sqrt, double ema_weight, double grad_scale, double decay_factor,
                             const int32_t* skip_flag, hipStream_t stream);

/* ---- signal representation either side of the path (representation.py:41-60, MovingAverageEnvelope) ------- */
/* x (N, C, T) fp32 NCW -> out (N, 2C, T) fp32: channels [0, C) = x / (env + eps), [C, 2C) = log(env + log_eps) - log(log_eps)/2,
 * env = mean of |x| over [t - W/2, t + (W-1)/2] with zeros outside the signal (np.convolve(..., mode="same")); float64 inside,
 * like numpy.  Requires window <= T (the reference's np.convolve changes length otherwise) and window <= 4096. */
int tq_envelope_fwd(const float* x, float* out, int N, int C, int T, int window, double log_eps, double eps, hipStream_t stream);
/* repr (N, 2C, T) -> waveform (N, C, T): scaled * (exp(log_env + log(log_eps)/2) + eps)   (representation.py:56-59) */
int tq_envelope_inv(const float* repr, float* out, int N, int C, int T, double log_eps, double eps, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TQDNE_HIP_H */
