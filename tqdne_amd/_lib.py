"""ctypes binding of libtqdne_hip.so (include/tqdne_hip.h).  Fails loudly: there is no CPU or
PyTorch fallback for the hot path -- if the library is missing the product raises."""

from __future__ import annotations

import ctypes as C
import os

from . import _build

_LIB = None

c_f32p = C.c_void_p  # device pointers travel as integers
VP = C.c_void_p
I = C.c_int
F = C.c_float
SZ = C.c_size_t

TQ_CONV_GN, TQ_CONV_SILU, TQ_CONV_EMB, TQ_CONV_RES, TQ_CONV_STATS, TQ_CONV_DROPOUT = 1, 2, 4, 8, 16, 32
TQ_CONV_POLY2 = 64
TQ_CONV_CH_TILES = 128
TQ_WFMT_BF16X3, TQ_WFMT_F16_MX8, TQ_WFMT_F16_MX6 = 0, 1, 2
TQ_KV_V_BF16, TQ_KV_V_F16 = 0, 1     # v_format of tq_conv1d_fwd_qkv / tq_attention_fwd_presplit
PACK_MODE = {TQ_WFMT_BF16X3: 0, TQ_WFMT_F16_MX8: 2, TQ_WFMT_F16_MX6: 3}   # tq_pack_conv_weight mode of a forward weight format
PACK_MODE_T = {TQ_WFMT_BF16X3: 1, TQ_WFMT_F16_MX6: 5}                      # ... of a data-gradient (transposed) weight format


DEFAULT_SCHEME = "f16mx6"
ABI_VERSION = 7   # include/tqdne_hip.h TQ_ABI_VERSION


def requested_scheme() -> str:
    """TQDNE_CONV_SCHEME resolved and validated: the scheme the forward convs AND (engine_bwd._dgrad) the data gradients may use."""
    v = os.environ.get("TQDNE_CONV_SCHEME", DEFAULT_SCHEME).lower()
    if v not in ("bf16x3", "f16mx8", "f16mx6"):
        raise ValueError(f"TQDNE_CONV_SCHEME={v!r}: expected bf16x3, f16mx8 or f16mx6")
    return v


def attn_v_format() -> int:
    """V-plane format of the inference attention pair: fp16 hi / lo + ONE fp16 softmax weight (two products, fp16 range, guarded by
    the plan's range flag) unless TQDNE_ATTN_VF16=0 or the fp32-range scheme is requested (TQDNE_CONV_SCHEME=bf16x3)."""
    if os.environ.get("TQDNE_ATTN_VF16", "1") == "0" or requested_scheme() == "bf16x3":
        return TQ_KV_V_BF16
    return TQ_KV_V_F16


# Round 6's 64-channel fp16 + MX-fp6 tile (64 channels x 128 positions): built, parity-green, and measured NEUTRAL TO SLOWER against the
# bf16x3 tile of 64 x 256 on every 64-channel layer of the paper UNet (profiles/r06_b_c64_tile_ab.txt: sample 159.6 vs 159.4 ms, train
# 23.4 vs 23.4 ms; fused-skip launches +5 ... +12 %): those layers are bound by their load / convert / store phases, not by matrix work.
# Off by default; TQDNE_CONV_MX6_C64=1 selects it.
MX6_C64 = os.environ.get("TQDNE_CONV_MX6_C64", "0") == "1"


def forward_wfmt(C_out: int, sources, stride: int = 1, upsample: bool = False, fused_skip: bool = False, k5_act: bool = False) -> int:
    """Contraction scheme of a forward conv launch (include/tqdne_hip.h, TQ_WFMT_*): fp16 + block-scaled corrections where the kernel
    is built for the shape (stride 1 incl. the nearest-upsampling convs, 128 | C_out, 64 | every source's channels incl. a fused skip
    conv's; round 6: also 64 | C_out for the ResBlock convs -- ``k5_act``: k = 5, GN + SiLU prologue, f16mx6 only), bf16x3 elsewhere.
    TQDNE_CONV_SCHEME: f16mx6 (default: e2m3 corrections with per-lane block scales), f16mx8 (round 1's e4m3 corrections with uniform
    scales), bf16x3 (the fp32-range three-product scheme everywhere)."""
    v = requested_scheme()
    co_ok = C_out % 128 == 0 or (C_out % 64 == 0 and k5_act and not upsample and v == "f16mx6" and MX6_C64)
    ok = stride == 1 and co_ok and all(c % 64 == 0 for c in sources if c)
    if fused_skip:
        ok = ok and os.environ.get("TQDNE_FUSED_SKIP_MX8", "1") != "0"
    if not ok or v == "bf16x3":
        return TQ_WFMT_BF16X3
    return TQ_WFMT_F16_MX6 if v == "f16mx6" else TQ_WFMT_F16_MX8
TQ_BWD_GN, TQ_BWD_SILU, TQ_BWD_DROPOUT, TQ_BWD_ACCUM, TQ_BWD_STATS = 1, 2, 4, 8, 16
STAT_SLOT = 128
TQ_AMAX_WAYS, TQ_AMAX_STRIDE = 16, 32           # include/tqdne_hip.h: max|dy| blocks of the column-sum kernels
TQ_AMAX_WORDS = TQ_AMAX_WAYS * TQ_AMAX_STRIDE


class TqGnFuse(C.Structure):
    _fields_ = [("counters", C.c_void_p), ("partner_stats", C.c_void_p), ("C_partner", C.c_int32), ("partner_first", C.c_int32),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("gscale", C.c_void_p), ("gshift", C.c_void_p),
                ("mean_rstd", C.c_void_p)]


class TqGnFold(C.Structure):
    _fields_ = [("stats0", C.c_void_p), ("stats1", C.c_void_p), ("slot0", C.c_int32), ("slot1", C.c_int32),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean_rstd", C.c_void_p)]


class TqConvDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T_in", C.c_int32), ("T_out", C.c_int32),
        ("C_in0", C.c_int32), ("C_in1", C.c_int32), ("C_out", C.c_int32),
        ("ktaps", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("upsample", C.c_int32), ("flags", C.c_int32), ("emb_stride", C.c_int32),
        ("dropout_site", C.c_uint32), ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
        ("C_skip0", C.c_int32), ("C_skip1", C.c_int32), ("wfmt", C.c_int32),
        ("range_flag", C.c_void_p), ("gn_fuse", C.POINTER(TqGnFuse)), ("t_tile", C.c_int32), ("reserved2", C.c_int32),
        ("gn_fold", C.POINTER(TqGnFold)),
    ]


class TqAdamChunk(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("ema", C.c_void_p),
                ("n", C.c_int32), ("reserved", C.c_int32)]


TQ_ADAM_CHUNK = 4096


class TqGemmJob(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("U", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("sam", C.c_int32), ("sak", C.c_int32), ("sbk", C.c_int32), ("sbn", C.c_int32), ("ldc", C.c_int32), ("ldu", C.c_int32),
                ("pre_b", C.c_int32), ("tile_begin", C.c_int32)]


class TqPackJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("C_out", C.c_int32), ("C_in", C.c_int32), ("K", C.c_int32),
                ("mode", C.c_int32), ("block_begin", C.c_int32), ("reserved", C.c_int32)]


class TqConvBwdDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T", C.c_int32), ("C_dy", C.c_int32), ("C_dx0", C.c_int32), ("C_dx1", C.c_int32),
        ("ktaps", C.c_int32), ("flags", C.c_int32), ("dropout_site", C.c_uint32), ("dropout_p", C.c_float),
        ("dropout_seed", C.c_uint64), ("wfmt", C.c_int32), ("reserved", C.c_int32), ("dy_amax", C.c_void_p),
    ]


_PROTOS = {
    "tq_abi_version": (I, []),
    "tq_build_flags": (I, []),
    "tq_conv_weight_pack_bytes": (SZ, [I, I, I, I]),
    "tq_pack_conv_weight": (I, [VP, I, I, I, I, VP, VP]),
    "tq_conv_tile_co": (I, [I]),
    "tq_conv1d_fwd": (I, [C.POINTER(TqConvDesc)] + [VP] * 11),
    "tq_conv1d_fwd_skip": (I, [C.POINTER(TqConvDesc)] + [VP] * 13),
    "tq_stem_conv_fwd": (I, [VP] * 6 + [I] * 5 + [VP]),
    "tq_head_conv_fwd": (I, [VP] * 9 + [I] * 5 + [VP]),
    "tq_head_conv_lds_bytes": (SZ, [I, I, I]),
    "tq_gn_finalize": (I, [VP, I, VP, I, I, I, VP, VP, VP, VP, VP, I, I, VP]),
    "tq_embed_fwd": (I, [VP] * 14 + [I, I, I, VP]),
    "tq_linear_fwd": (I, [VP] * 4 + [I, I, I, VP]),
    "tq_attention_fwd": (I, [VP, VP, VP, VP, I, I, I, I, VP]),
    "tq_attention_fwd_presplit": (I, [VP, VP, VP, I, I, I, I, I, VP]),
    "tq_conv1d_fwd_qkv": (I, [C.POINTER(TqConvDesc)] + [VP] * 7 + [I, I, I, VP]),
    "tq_attention_workspace_bytes": (SZ, [I, I, I, I]),
    "tq_attention_bwd": (I, [VP] * 6 + [I, I, I, I, VP]),
    "tq_attention_bwd_ws": (I, [VP] * 7 + [I, I, I, I, VP]),
    "tq_attention_bwd_ws_kv": (I, [VP] * 8 + [I, I, I, I, VP]),
    "tq_edm_scalars": (I, [VP, I, F, VP, VP, VP, VP, VP, I, VP]),
    "tq_cm_scalars": (I, [VP, I, F, F, VP, VP, I, VP]),
    "tq_edm_noise_inject": (I, [VP, VP, VP, F, F, VP, VP, I, I, VP]),
    "tq_edm_loss": (I, [VP, VP, VP, VP, VP, I, I, VP]),
    "tq_heun_euler": (I, [VP] * 7 + [SZ, VP]),
    "tq_heun_correct": (I, [VP] * 8 + [SZ, VP]),
    "tq_sampler_init": (I, [VP] * 4 + [SZ, VP]),
    "tq_heun_churn": (I, [VP, VP, VP, C.c_double, VP, VP, SZ, VP]),
    "tq_axpy_sigma": (I, [VP, VP, VP, VP, I, I, VP]),
    "tq_pseudo_huber_loss": (I, [VP, VP, VP, F, VP, VP, I, I, VP]),
    "tq_mse_loss": (I, [VP, VP, VP, VP, SZ, VP]),
    "tq_scale_add2": (I, [VP, VP, VP, VP, VP, I, I, VP]),
    "tq_ddpm_step": (I, [VP, VP, VP, VP, SZ, I] + [C.c_double] * 6 + [VP]),
    "tq_vae_reparam_fwd": (I, [VP, VP, VP, VP, I, I, I, VP]),
    "tq_vae_reparam_bwd": (I, [VP, VP, VP, VP, F, I, I, I, VP]),
    "tq_concat_scale": (I, [VP, VP, VP, VP, I, I, I, I, VP]),
    "tq_envelope_fwd": (I, [VP, VP, I, I, I, I, C.c_double, C.c_double, VP]),
    "tq_envelope_inv": (I, [VP, VP, I, I, I, C.c_double, C.c_double, VP]),
    "tq_adam_ema_step": (I, [VP, I] + [C.c_double] * 8 + [VP]),
    "tq_adam_ema_step_guarded": (I, [VP, I] + [C.c_double] * 8 + [VP, VP]),
    "tq_conv1d_bwd_data": (I, [VP] * 11),
    "tq_conv1d_bwd_weight_workspace": (SZ, [VP]),
    "tq_conv1d_bwd_weight": (I, [VP] * 8 + [SZ, VP]),
    "tq_conv1d_bwd_weight_colsum": (I, [VP] * 8 + [SZ, VP, I, VP, VP, VP]),
    "tq_gn_bwd_finalize": (I, [VP, VP, VP, I, I, I, VP, VP, VP, VP, VP, VP]),
    "tq_gn_bwd_apply": (I, [VP] * 7 + [I] * 6 + [VP]),
    "tq_gn_bwd_apply_colsum": (I, [VP] * 7 + [I] * 6 + [VP, I, VP, VP, VP, VP]),
    "tq_pack_job_blocks": (I, [I, I, I, I]),
    "tq_pack_jobs": (I, [VP, I, I, VP]),
    "tq_gemm_tiles": (I, [I, I]),
    "tq_gemm_f32_jobs": (I, [VP, I, I, VP]),
    "tq_fourier_features": (I, [VP, VP, VP, I, I, VP]),
    "tq_colsum": (I, [VP, I, I, I, VP, I, VP, VP, VP, VP, VP]),
    "tq_zero_stuff": (I, [VP, VP, I, I, I, I, VP]),
    "tq_pair_sum": (I, [VP, VP, I, I, I, I, VP]),
    "tq_upsample_poly_wgrad_fold": (I, [VP, VP, I, I, VP]),
    "tq_stem_conv_bwd_weight": (I, [VP] * 4 + [I] * 5 + [VP]),
    "tq_head_conv_bwd": (I, [VP] * 10 + [I] * 5 + [VP]),
    "tq_stem_head_bwd_workspace": (SZ, []),
    "tq_stem_conv_bwd_weight_ws": (I, [VP] * 4 + [I] * 5 + [VP, SZ, VP]),
    "tq_head_conv_bwd_ws": (I, [VP] * 10 + [I] * 5 + [VP, SZ, VP]),
}

# entry points of later rounds are optional at load time but listed in the header check
_OPTIONAL = {}


def lib_path() -> str:
    # TQDNE_HIP_LIB: developer override used by tools/ to A/B kernel variants; the default is the in-tree build
    # (TQDNE_BUILD_EXPERIMENTS=1: the build that also holds the opt-in kernels, libtqdne_hip_exp.so)
    return os.environ.get("TQDNE_HIP_LIB", _build.EXP_LIBPATH if _build.EXPERIMENTS else _build.LIBPATH)


def has_experiments() -> bool:
    """True when the loaded library was built with the opt-in kernels (conv1d_w4, slim tile, in-launch GroupNorm fold)."""
    return bool(load().tq_build_flags() & 1)


def load():
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the tqdne_amd hot path is HIP-only (no CPU / PyTorch fallback). "
            "Build it with `python -m tqdne_amd._build` (needs hipcc, cross-compiles gfx950 without a GPU)."
        )
    lib = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError = header/library mismatch: loud
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in _OPTIONAL.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    if lib.tq_abi_version() != ABI_VERSION:
        raise RuntimeError("libtqdne_hip.so ABI version mismatch")
    _LIB = lib
    return lib


def exported_symbols():
    return sorted(list(_PROTOS) + list(_OPTIONAL))


class TqError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        kind = {-1: "TQ_ERR_ARG", -2: "TQ_ERR_SHAPE"}.get(rc, f"hipError_t {rc}")
        raise TqError(f"libtqdne_hip: {what} failed with {kind}")
