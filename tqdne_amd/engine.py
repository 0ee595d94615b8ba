"""Execution plans that drive libtqdne_hip.so for one (model, batch, length, device).

A plan is built once: every intermediate tensor gets a static buffer in HBM ((B, T, C) fp32 channels-last plus
per-channel partial statistics), every kernel launch becomes a pre-bound ``(function, argument tuple)`` and a
forward pass is a flat loop of C-ABI calls on torch's current HIP stream (so it can be captured in a HIP graph).
PyTorch is used for device memory and streams only.

Fusion map (reference op chain -> launches), per block:
  ResBlock (unet.py:131-143):  gn_finalize | conv1[GN+SiLU -> k5 -> +bias +emb, stats] | gn_finalize |
                               (1x1 skip conv) | conv2[GN+SiLU(+dropout) -> k5 -> +bias +skip, stats]
  AttentionBlock (blocks.py:139-145): gn_finalize | qkv conv[GN -> k1] | flash attention | proj conv[k1 + x, stats]
  Downsample / Upsample (blocks.py:29-108): one conv (stride 2 / nearest-x2 folded into the gather)
  th.cat([h, hs.pop()]) (unet.py:396): never materialised, consumers read two sources.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from .engine_bwd import BackwardPlan
from ._lib import (TQ_CONV_DROPOUT, TQ_CONV_EMB, TQ_CONV_GN, TQ_CONV_RES, TQ_CONV_SILU, TQ_CONV_STATS, STAT_SLOT,
                   TqConvDesc, check)


def require_device(x: torch.Tensor):
    if not x.is_cuda:
        raise RuntimeError(
            "tqdne_amd runs its hot path on MI355X HIP kernels only; got a CPU tensor. "
            "(The CPU restatement lives in oracle/ and is test infrastructure, not a fallback.)"
        )
    if x.dtype != torch.float32:
        raise TypeError("tqdne_amd expects fp32 tensors at the boundary, like the reference (precision 32)")


def _noop_launch(*_args):
    """stands where a launch was planned that another launch has absorbed (returns the C ABI's success code)"""
    return 0


def nslots(T: int) -> int:
    return (T + STAT_SLOT - 1) // STAT_SLOT


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class Act:
    """A channels-last activation (B, T, C) and, optionally, its per-channel partial statistics."""

    __slots__ = ("buf", "stats", "C", "T", "grad", "gw", "prod", "slot")

    def __init__(self, buf, stats, C, T, slot=STAT_SLOT):
        self.buf, self.stats, self.C, self.T = buf, stats, C, T
        self.slot = slot   # positions per statistics slot (32 for the output of a small-tile conv, TqConvDesc.t_tile)
        self.prod = None   # descriptors (TqConvDesc) of the launches that write this tensor and its statistics, if they can fold them
        self.grad = None   # gradient buffer (training plans only)
        self.gw = False    # backward-plan construction: has some op already written the gradient?


FUSE_SKIP = os.environ.get("TQDNE_FUSE_SKIP", "1") != "0"  # A/B switch for the fused skip-conv launch
# GroupNorm finalisation inside the launch that completes the statistics (TqConvDesc.gn_fuse, "last arriver") instead of a
# tq_gn_finalize launch per GroupNorm: TQDNE_GN_FUSE=1.  Built, parity- and concurrency-tested (tests/test_concurrency.py), and
# measured NEUTRAL (18-step sample at B = 64, 4 lanes: 165.7 vs 166.0 ms; 1 lane 174.5 vs 172.8; tiny UNet B = 4: 41.9 vs 41.6 ms):
# what the 49 launches cost comes back as the fold's dependent-load chain on the tail of the producing launch -> off by default
# (an experiment since round 4: only libraries built with TQDNE_BUILD_EXPERIMENTS=1 carry it)
GN_FUSE = os.environ.get("TQDNE_GN_FUSE", "0") == "1"
# Round 6: the TRAINING forward of Upsample runs in the two-phase k = 3 form too (it was the inference form only), and its gradients are
# those of that k = 3 conv (engine_bwd._bwd_up_poly): 3/5 of the multiply-adds in all three passes, no (B, 2T, C) scratch gradient, no
# tq_pair_sum.  TQDNE_POLYPHASE_TRAIN=0: the k = 5 launches over the upsampled gather, as before.
POLY_TRAIN = os.environ.get("TQDNE_POLYPHASE_TRAIN", "1") != "0"
FUSE_SKIP_CO = 32  # smallest output-channel multiple fused (measured: 128 -> +3.9 %, 64 -> +1.3 % more on the bench step)
# Small position tile (TqConvDesc.t_tile = 32) for the ResBlock convs of launch-bound plans: a plan whose batch is at most SMALL_TILE_B
# samples launches 16-64 workgroups of the default tiles per conv on 256 compute units (tiny UNet, B = 4: 12-30 us per conv launch).
# The choice is a property of the PLAN (its batch), never of how many lanes run: lanes of a larger batch have >= 8 samples each, so
# the lane-versus-one-lane bit-identity holds.  A sample computed in a small batch differs from the same sample in a large one at
# rounding level only through the association order of the GroupNorm statistics.  TQDNE_SMALL_TILE=0 turns it off.
SMALL_TILE_B = int(os.environ.get("TQDNE_SMALL_TILE_B", "4")) if os.environ.get("TQDNE_SMALL_TILE", "1") != "0" else 0
# ... and, per LAYER of a plan that has the device to itself (UNetEngine.solo: not a lane of a multi-lane sampler), where the default
# tile's grid B * ceil(T_out / 128) is at most SMALL_TILE_WGS workgroups -- the T = 512 level of a 16-sample plan: 64 workgroups on 256
# compute units, 36 -> 26 us per 256 -> 256 conv, cfg3's sample 92.5 -> 87.0 ms.  Lanes keep the default tile: four lanes fill the
# device between them, and there the small tile's 4x weight traffic costs 161 -> 180 ms per B = 64 sample; the mid level (128
# workgroups) loses 7-18 % alone (profiles/r04_v_small_tile_per_layer.txt).
SMALL_TILE_WGS = int(os.environ.get("TQDNE_SMALL_TILE_WGS", "64")) if os.environ.get("TQDNE_SMALL_TILE", "1") != "0" else 0
# Round 6: a small-tile conv folds its own GroupNorm (TqConvDesc.gn_fold, consumer side) instead of a tq_gn_finalize launch in front of it:
# a plan of <= 4 samples is ~100 dependent launches of 5-30 us, and 45 % of them were these.  TQDNE_GN_FOLD_SMALL=0: the launches.
GN_FOLD_SMALL = os.environ.get("TQDNE_GN_FOLD_SMALL", "1") != "0"
# ... and (experiment) the default tiles of the fp16 + MX-fp6 scheme, with the fold behind the first chunk's loads: TQDNE_GN_FOLD=1
GN_FOLD_DEFAULT = os.environ.get("TQDNE_GN_FOLD", "0") == "1"
CONCURRENT_LANE0 = 8   # plan-cache lane ids from here on: sub-batch plans that run concurrently (see UNetModel._engine)


class ConvRec:
    """Everything the backward of one fused conv launch needs (the forward descriptor is reused for the weight gradient)."""

    __slots__ = ("site", "desc", "srcs", "gn", "out", "stride", "upsample", "silu", "dropout", "poly")

    def __init__(self, site, desc, srcs, gn, out, stride, upsample, silu, dropout):
        self.site, self.desc, self.srcs, self.gn, self.out = site, desc, srcs, gn, out
        self.stride, self.upsample, self.silu, self.dropout = stride, upsample, silu, dropout
        self.poly = None   # (derived two-phase k = 3 site, its descriptor): the TRAINING forward of this Upsample ran in that form


_SIDE_STREAMS: dict = {}


def side_stream(dev, i: int = 1) -> "torch.cuda.Stream":
    """The process-wide side stream number ``i`` >= 1 of a device.  The sampler lanes, the consistency sampler's lanes and the
    backward plans' weight-gradient stream all draw from this ONE pool: ROCm multiplexes HIP streams onto 4 hardware queues by
    default, and a fifth live stream (a backward plan with a stream of its own) put two sampler lanes on one queue -- the
    18-step sample went from 163 to 222 ms."""
    key = (str(dev), i)
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return s


def hiprio_stream(dev) -> "torch.cuda.Stream":
    """a high-priority stream of the device (experiment TQDNE_BWD_HIPRIO, engine_bwd.BackwardPlan.run)"""
    key = (str(dev), "hiprio")
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev, priority=-1)
    return s


def reserve_side_streams(dev, n: int = 3):
    """Create the pool's first ``n`` streams NOW (called when the first plan of a device is built).  Stream creation order decides
    how ROCm spreads streams over hardware queues: with the three lane streams created AFTER anything had captured a HIP graph
    (torch's capture stream, even for a one-kernel graph), the 4-lane sampler ran at 250 instead of 160 ms per 18-step sample at
    B = 64 -- also with 16 hardware queues -- while the same graph captured after the lanes existed cost nothing
    (tools/graph_side_effect.py, round 3).  Reserving them up front makes the good order the only one."""
    if torch.device(dev).type != "cuda" or (str(dev), "reserved") in _SIDE_STREAMS or torch.cuda.is_current_stream_capturing():
        return
    _SIDE_STREAMS[(str(dev), "reserved")] = True
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        import warnings
        warnings.warn("tqdne_amd: the RCCL process group exists before this device's side streams made their first submission; if its "
                      "communicator is already created the backward's two streams may share a hardware queue (training step ~25 % "
                      "slower).  Call tqdne_amd.trainer.init_process_group(...) (or tqdne_amd.engine.reserve_side_streams(device)) "
                      "before torch.distributed.init_process_group.", RuntimeWarning, stacklevel=2)
    # (a stream is bound to its hardware queue by its first submission, not by its creation: one tiny launch on each)
    main = torch.cuda.current_stream(dev)
    z = torch.zeros(1, device=dev)
    for i in range(1, n + 1):
        s = side_stream(dev, i)
        s.wait_stream(main)
        with torch.cuda.stream(s):
            z.add_(0)
        main.wait_stream(s)


_PACK_TABLES: dict = {}
PACK_BATCH = os.environ.get("TQDNE_PACK_BATCH", "1") != "0"   # A/B switch: 0 = one launch per tensor (rounds 1-2 behaviour)


def pack_batch(lib, dev, jobs, stream, capturing=False):
    """Run a list of (src_ptr, dst_ptr, C_out, C_in, K, mode) weight packs / copies (mode 4: C_out floats) as ONE launch
    (tq_pack_jobs).  The device job table is built once per distinct list (a host-to-device copy) and cached; under stream capture
    an unseen list falls back to one launch per job."""
    if not jobs:
        return
    key = (str(dev), tuple(jobs))
    ent = _PACK_TABLES.get(key)
    # (tables are never evicted: a launch queued on another stream may still read one, and the caching allocator would hand its
    # block to the next allocation; they are ~40 bytes per job, and past 4096 distinct lists new ones run as per-tensor launches)
    if ent is None and (capturing or not PACK_BATCH or len(jobs) < 2 or len(_PACK_TABLES) >= 4096):
        ent = False
    if ent is None:
        arr = (_lib.TqPackJob * len(jobs))()
        total = 0
        for jb, (src, dst, co, ci, k, mode) in zip(arr, jobs):
            jb.src, jb.dst, jb.C_out, jb.C_in, jb.K, jb.mode, jb.block_begin = src, dst, co, ci, k, mode, total
            nb = lib.tq_pack_job_blocks(co, ci, k, mode)
            if nb <= 0:
                raise ValueError(f"bad pack job {(co, ci, k, mode)}")
            total += nb
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        ent = _PACK_TABLES[key] = (table, len(jobs), total)
    if ent is False:
        for src, dst, co, ci, k, mode in jobs:
            if mode == 4:
                _lib_copy_floats(dst, src, co, stream)
            else:
                check(lib.tq_pack_conv_weight(src, co, ci, k, mode, dst, stream), "pack")
        return
    table, n, total = ent
    check(lib.tq_pack_jobs(table.data_ptr(), n, total, stream), "pack jobs")


def _lib_copy_floats(dst, src, n, stream):
    import ctypes
    rt = ctypes.CDLL("libamdhip64.so")
    rc = rt.hipMemcpyAsync(ctypes.c_void_p(dst), ctypes.c_void_p(src), ctypes.c_size_t(4 * n), 3, ctypes.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"hipMemcpyAsync failed: {rc}")


class PackedStore:
    """The packed MFMA weight fragments (and the derived weight tensors) of ONE model on one device, shared by every execution
    plan of that model: the fragments depend on the weights only, not on the batch, the length or the sampler lane.  One copy
    instead of one per plan (the 4-lane sampler + the training plan held five 62 MB copies and re-packed each of them after every
    optimizer step), and one copy for the XCD L2s / the Infinity Cache to keep.

    Plans on different HIP streams use the store: a plan that finds stale fragments re-packs them on ITS stream after waiting for
    the last use recorded by every other stream; a plan on another stream than the packing one waits for the pack's event."""

    def __init__(self, device):
        self.dev = device
        self.entries = {}       # key -> {"buf": tensor, "ver": version tag of the weights the buffer was built from}
        self.event = None       # recorded behind the most recent pack
        self.gen = 0            # pack generation
        self.pack_stream = None
        self.users = {}         # stream handle -> event recorded behind that stream's most recent forward

    def entry(self, key, shape, dtype=torch.uint8):
        e = self.entries.get(key)
        if e is None:
            e = {"buf": torch.empty(shape, dtype=dtype, device=self.dev), "ver": None}
            self.entries[key] = e
        assert tuple(e["buf"].shape) == (tuple(shape) if isinstance(shape, (tuple, list)) else (shape,)), key
        return e


def get_store(model, device) -> PackedStore:
    if os.environ.get("TQDNE_SHARED_STORE", "1") == "0":   # A/B switch: a private store per plan (the round-1 behaviour)
        return PackedStore(device)
    stores = model.__dict__.setdefault("_packed_stores", {})
    st = stores.get(str(device))
    if st is None:
        st = stores[str(device)] = PackedStore(device)
    return st


class ConvSite:
    """One convolution's weights: torch parameter + packed bf16 hi/lo MFMA fragments (a buffer of the model's PackedStore)."""

    __slots__ = ("weight", "bias", "packed", "packed_t", "pack_mode_t", "C_out", "C_in", "K", "version", "name", "pack_mode", "entry", "tail")

    def __init__(self, name, weight, bias, device, lib, packed=None, tail_bytes=0, store=None):
        self.name = name
        self.weight, self.bias = weight, bias
        self.C_out, self.C_in, self.K = weight.shape
        nbytes = lib.tq_conv_weight_pack_bytes(self.C_out, self.C_in, self.K, 0)
        # tail_bytes: room for a second conv's fragments right behind this one's (fused skip conv, tq_conv1d_fwd_skip)
        self.entry = None      # store entry owning `packed` (None: a tail view of another site's buffer, or a private buffer)
        self.tail = None       # the site whose fragments follow in the same buffer
        if packed is not None:
            self.packed = packed
        elif store is not None:
            self.entry = store.entry("conv:" + name, nbytes + tail_bytes)
            self.packed = self.entry["buf"]
        else:
            self.packed = torch.empty(nbytes + tail_bytes, dtype=torch.uint8, device=device)
        self.pack_mode = 0     # tq_pack_conv_weight mode of `packed` (2: TQ_WFMT_F16_MX8), set by the plan builder
        self.packed_t = None  # transposed / tap-flipped fragments for the data gradient (training only)
        self.pack_mode_t = 1  # their tq_pack_conv_weight mode: 1 bf16x3, 5 fp16 + MX-fp6 (set by the backward plan)
        self.version = -1


def _recorded_event():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _check_head_limits(C_in: int, C_out: int, k: int):
    """The limits of tq_head_conv_fwd, asked of the library itself (tq_head_conv_lds_bytes: 0 = not built for the shape) when the
    plan is built, so that an unsupported model fails at construction with a message rather than at its first forward."""
    if _lib.load().tq_head_conv_lds_bytes(C_in, C_out, k) == 0:
        raise NotImplementedError(
            f"output conv {C_in} -> {C_out} channels, k = {k}: the HIP head kernel takes 16 | C_in <= 128, C_out <= 16, k in (1, 3, 5) "
            "and <= 64 KB of LDS")


def shared_range_flag(model, device) -> torch.Tensor:
    """The model's range-guard flag on ``device`` (int32[1]); created on first use."""
    flags = model.__dict__.setdefault("_range_flags", {})
    key = str(device)
    if key not in flags:
        flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return flags[key]


class Probe:
    """HIP-event timings of one op of the plan (events are recorded on the stream the kernel is launched on)."""

    def __init__(self, idx, name, flops):
        self.idx, self.name, self.flops, self.events = idx, name, flops, []

    def reset(self):
        self.events = []

    def result(self):
        if not self.events:
            return None, self.flops, self.name, 0
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.events]
        return sum(ms) / len(ms), self.flops, self.name, len(ms)


_PLAN_UID = __import__("itertools").count(1)


class UNetEngine:
    def __init__(self, model, B: int, T: int, device: torch.device, solo: bool = True):
        self.lib = _lib.load()
        self.m = model
        self.B, self.T, self.dev = B, T, device
        self.solo = solo   # False: one of several plans that run concurrently (sampler / training lanes)
        reserve_side_streams(device)
        self.store = get_store(model, device)
        self._seen_pack = {}
        self._clean_tag = None
        self.E = 4 * getattr(model, "model_channels", 0)
        self._keep = []          # ctypes structs / tensors referenced by raw pointers
        self.ops: List[Tuple] = []          # launches of a forward whose backward may follow
        self.ops_infer: List[Tuple] = []    # same order and length; inference-only variants where they exist
        self.op_bytes: List[int] = []       # same order: algorithmic bytes per launch
        # contraction scheme of the forward convs: "auto" = the fp16-range scheme (TQ_WFMT_F16_MX8) wherever it is built, "bf16x3" =
        # fp32 range everywhere (after the range guard fired, or on request: model._conv_scheme / TQDNE_CONV_SCHEME)
        self.scheme = "auto"
        self.plan_epoch = 0
        self.uid = next(_PLAN_UID)          # never reused: captured HIP graphs are keyed by it (a plan evicted from the bounded cache
                                            # and rebuilt for the same shape owns NEW buffers)
        self._wfmt_sites = []               # (descriptor, [conv sites packed for it], preferred wfmt)
        self.kv_v_format = _lib.attn_v_format()   # V planes of the inference attention pair (TQ_KV_V_*): a property of the plan
        self._vfmt_ops = []                 # indices into ops_infer of the launches that carry it as their last argument
        self.ckpt = bool(getattr(model, "use_checkpoint", False))   # block-internal activations shared + recomputed in the backward
        self._ckpt_pool = {}
        self._attn_train_ops = []           # (index into ops, index into tape, workspace bytes) of every tq_attention_fwd launch
        self._block_kv = False              # training forwards keep each attention block's K / V planes for its backward
        # set by the conv epilogues (TqConvDesc.range_flag); ONE flag per model and device, shared by every plan of the model (any
        # batch, length, lane): the optimizer launch of the trainer is predicated on it (tq_adam_ema_step_guarded)
        self.range_flag = shared_range_flag(model, device)
        self._range_host = torch.zeros(1, dtype=torch.int32).pin_memory() if device.type == "cuda" else torch.zeros(1, dtype=torch.int32)
        self._range_evt = None
        self._trace = None                  # list: HIP-event pairs around EVERY launch of the next forwards (measurement only)
        self.conv_sites: List[ConvSite] = []
        self.poly_sites: List[Tuple[ConvSite, ConvSite]] = []   # (derived two-phase k = 3 site, the Upsample conv it restates)
        self.dropout_descs: List[TqConvDesc] = []
        self.acts: List[Act] = []
        self.gn_bufs: List[torch.Tensor] = []   # every GroupNorm's folded coefficients (see poison_gn)
        self.gn_fused = 0                         # GroupNorms folded inside their producer's launch
        self.poison_gn = False               # tests set it on the plan object (tests/test_concurrency.py); no environment switch
        self._probe = None
        self.tape = []
        self.last_rec = None
        self._bwd = None
        self.dgrad_sites = []
        self._wt_version = None
        self._build()
        if getattr(model, "_conv_scheme", "auto") == "bf16x3":
            self._set_scheme_bf16x3()

    # ------------------------------------------------------------------ allocation helpers
    def _empty(self, *shape, dtype=torch.float32):
        t = torch.empty(*shape, dtype=dtype, device=self.dev)
        self._keep.append(t)
        return t

    def _act(self, C_: int, T_: int, stats: bool = True, slot: int = STAT_SLOT) -> Act:
        a = Act(self._empty(self.B, T_, C_), self._empty(self.B, (T_ + slot - 1) // slot, C_, 2) if stats else None, C_, T_, slot)
        self.acts.append(a)
        return a

    def _ckpt_act(self, tag: str, C_: int, T_: int, stats: bool = True, slot: int = STAT_SLOT) -> Act:
        """``use_checkpoint`` plans (reference nn.py:137-215: a block's intermediate activations are not kept for the backward but
        recomputed there): an activation INSIDE a ResBlock / AttentionBlock lives in a buffer shared by every block of the plan with
        that shape; the backward plan re-issues the block's forward launches in front of the block's backward.  Each block still gets
        an Act of its own (gradient bookkeeping), the tensors are shared."""
        key = (tag, C_, T_, stats, slot)
        first = self._ckpt_pool.get(key)
        if first is None:
            first = self._ckpt_pool[key] = self._act(C_, T_, stats, slot)
            return first
        a = Act(first.buf, first.stats, C_, T_, slot)
        self.acts.append(a)
        return a

    def _site(self, name: str, conv: torch.nn.Module) -> ConvSite:
        s = ConvSite(name, conv.weight, conv.bias, self.dev, self.lib, store=self.store)
        self.conv_sites.append(s)
        return s

    def _site_pair(self, name: str, conv, tail_name: str, tail_conv):
        """Two sites sharing one packed buffer: `conv`'s fragments followed by the 1x1 `tail_conv`'s (fused skip conv)."""
        lib = self.lib
        co, ci, k = conv.weight.shape
        tco, tci, tk = tail_conv.weight.shape
        assert tk == 1 and tco == co
        main_bytes = lib.tq_conv_weight_pack_bytes(co, ci, k, 0)
        tail_bytes = lib.tq_conv_weight_pack_bytes(tco, tci, 1, 0)
        s = ConvSite(name, conv.weight, conv.bias, self.dev, lib, tail_bytes=tail_bytes, store=self.store)
        t = ConvSite(tail_name, tail_conv.weight, tail_conv.bias, self.dev, lib, packed=s.packed[main_bytes:])
        s.tail = t
        self.conv_sites += [s, t]
        return s, t

    # ------------------------------------------------------------------ op builders
    def _emit(self, op, infer_op=None, nbytes=0):
        """append a launch to the plan; ``infer_op`` replaces it in forwards that no backward will follow.
        ``nbytes``: algorithmic HBM bytes of the launch (fp32 inputs read once + output written once), for the bench's tables"""
        self.ops.append(op)
        self.ops_infer.append(op if infer_op is None else infer_op)
        self.op_bytes.append(nbytes)

    def _gn(self, srcs: Sequence[Act], norm: torch.nn.GroupNorm, defer: bool = False):
        """Folded scale / shift (B, C) of a GroupNorm over the (concatenated) sources.  Where the most recent source is written by a
        conv launch that can do it, the fold rides in that launch (TqGnFuse: the workgroup completing a sample's statistics folds
        them); otherwise -- the stem's output, a tensor that is already the last source of another GroupNorm -- a tq_gn_finalize
        launch.  The coefficient buffers are registered in ``gn_bufs`` so that tests can poison them (``poison_gn``): a fold that
        did not happen then surfaces as NaN instead of as the previous evaluation's (nearly right) coefficients."""
        C_ = sum(s.C for s in srcs)
        gscale, gshift = self._empty(self.B, C_), self._empty(self.B, C_)
        mean_rstd = self._empty(self.B, 32, 2)
        self.gn_bufs += [gscale, gshift, mean_rstd]
        s0 = srcs[0]
        s1 = srcs[1] if len(srcs) > 1 else None
        prod = s0.prod if GN_FUSE else None
        if GN_FUSE and not _lib.has_experiments():
            raise RuntimeError("TQDNE_GN_FUSE=1 needs the experiments build of the library (TQDNE_BUILD_EXPERIMENTS=1)")
        if prod and all(not d.gn_fuse for d in prod):
            for d in prod:   # (one TqGnFuse and one ticket counter per launch form: the forms tile the tensor differently)
                f = _lib.TqGnFuse()
                counters = torch.zeros(self.B, dtype=torch.int64, device=self.dev)
                self._keep += [f, counters]
                f.counters = counters.data_ptr()
                f.partner_stats, f.C_partner, f.partner_first = (_p(s1.stats), s1.C, 0) if s1 is not None else (None, 0, 0)
                f.gamma, f.beta = _p(norm.weight), _p(norm.bias)
                f.gscale, f.gshift, f.mean_rstd = _p(gscale), _p(gshift), _p(mean_rstd)
                d.gn_fuse = C.pointer(f)
            self.gn_fused += 1
            return gscale, gshift, mean_rstd
        fin = ((self.lib.tq_gn_finalize, (
            _p(s0.stats), s0.C, _p(s1.stats) if s1 else None, s1.C if s1 else 0, self.B, s0.T,
            _p(norm.weight), _p(norm.bias), _p(gscale), _p(gshift), _p(mean_rstd), s0.slot, s1.slot if s1 else 0), "gn_finalize", 0),
            4 * self.B * (2 * nslots(s0.T) * C_ + 2 * C_))
        if defer:
            # the consuming conv decides (``_conv``): a small-tile launch folds these statistics itself, any other launch gets the
            # tq_gn_finalize launch in front of it
            self._pending_gn = dict(key=gscale.data_ptr(), fin=fin, srcs=(s0, s1), norm=norm, mean_rstd=mean_rstd)
            return gscale, gshift, mean_rstd
        self._emit(fin[0], nbytes=fin[1])
        return gscale, gshift, mean_rstd

    def _conv(self, srcs: Sequence[Act], site: ConvSite, *, gn=None, silu=False, emb_ptr=None, res: Optional[Act] = None,
              stats=True, stride=1, upsample=False, dropout_site: Optional[int] = None, launch: bool = True,
              skip: Optional[Tuple[Sequence[Act], ConvSite]] = None, qkv_planes=None, ckpt_tag: Optional[str] = None) -> Optional[Act]:
        """launch=False only records the conv (descriptor for its gradients): its product is formed by another launch.
        skip=(srcs, 1x1 site): fuse that convolution of the un-activated srcs into this launch (site.packed holds both)."""
        s0 = srcs[0]
        s1 = srcs[1] if len(srcs) > 1 else None
        if s1 is not None and s1.T != s0.T:
            # the reference fails in th.cat([h, hs.pop()], dim=1) (unet.py:396) when the length is not divisible by 2^levels
            raise RuntimeError(f"Sizes of tensors must match except in dimension 1: lengths {s0.T} and {s1.T} at {site.name} "
                               "(the signal length must be divisible by the UNet's total down-sampling factor)")
        T_in = s0.T
        if stride == 2:
            T_out = (T_in + 2 - site.K) // 2 + 1
            pad = site.K // 2
        else:
            T_out = 2 * T_in if upsample else T_in
            pad = site.K // 2
        srcs_c = [s0.C, (s1.C if s1 else 0)] + ([a.C for a in skip[0]] if skip is not None else [])
        k5_act = site.K == 5 and gn is not None and silu and stride == 1 and not upsample and qkv_planes is None
        wfmt = _lib.forward_wfmt(site.C_out, srcs_c, stride, upsample, fused_skip=skip is not None, k5_act=k5_act) if launch else 0
        # the small tile where it is built (see SMALL_TILE_B): the ResBlock convs of a small-batch plan
        small = (launch and (self.B <= SMALL_TILE_B or (self.solo and self.B * ((T_out + 127) // 128) <= SMALL_TILE_WGS)) and k5_act
                 and wfmt in (_lib.TQ_WFMT_BF16X3, _lib.TQ_WFMT_F16_MX6) and not GN_FUSE)
        if small and wfmt == _lib.TQ_WFMT_F16_MX6 and site.C_out % 128:
            wfmt = _lib.TQ_WFMT_BF16X3   # (the small tile's fp16 + MX-fp6 form is the 128-channel one)
        if launch and ckpt_tag is not None and self.ckpt:
            out = self._ckpt_act(ckpt_tag, site.C_out, T_out, stats, slot=32 if small else STAT_SLOT)
        else:
            out = self._act(site.C_out, T_out, stats, slot=32 if small else STAT_SLOT) if launch else None
        d = TqConvDesc()
        d.t_tile = 32 if small else 0
        pend = getattr(self, "_pending_gn", None)
        if pend is not None and gn is not None and pend["key"] == gn[0].data_ptr():
            self._pending_gn = None
            fold_default = (GN_FOLD_DEFAULT and not small and k5_act and wfmt == _lib.TQ_WFMT_F16_MX6 and self.scheme == "auto"
                            and getattr(self.m, "_conv_scheme", "auto") == "auto")
            if fold_default and launch and not self.ckpt and not GN_FUSE:
                # the tq_gn_finalize launch stays in the plan as a no-op: the range-guard fallback moves this conv to the three-product
                # scheme, whose default tiles do not fold -- it then gets its launch back (_set_scheme_bf16x3)
                fn, fargs, fname, ffl = pend["fin"][0]
                self._fold_default_ops = getattr(self, "_fold_default_ops", [])
                self._fold_default_ops.append((len(self.ops), fn, d))
                self._emit((_noop_launch, fargs, fname + " (folded into its consumer)", ffl), nbytes=0)
            if launch and ((small and GN_FOLD_SMALL) or fold_default) and not self.ckpt and not GN_FUSE:
                f = _lib.TqGnFold()
                ps0, ps1 = pend["srcs"]
                f.stats0, f.stats1 = _p(ps0.stats), (_p(ps1.stats) if ps1 is not None else None)
                f.slot0, f.slot1 = ps0.slot, (ps1.slot if ps1 is not None else 0)
                f.gamma, f.beta, f.mean_rstd = _p(pend["norm"].weight), _p(pend["norm"].bias), _p(pend["mean_rstd"])
                self._keep.append(f)
                d.gn_fold = C.pointer(f)
                self.gn_folded = getattr(self, "gn_folded", 0) + 1
            else:
                self._emit(pend["fin"][0], nbytes=pend["fin"][1])
        d.B, d.T_in, d.T_out = self.B, T_in, T_out
        d.C_in0, d.C_in1, d.C_out = s0.C, (s1.C if s1 else 0), site.C_out
        assert d.C_in0 + d.C_in1 == site.C_in, (site.name, d.C_in0, d.C_in1, site.C_in)
        d.ktaps, d.stride, d.pad, d.upsample = site.K, stride, pad, int(upsample)
        flags = 0
        if gn is not None:
            flags |= TQ_CONV_GN
        if silu:
            flags |= TQ_CONV_SILU
        if emb_ptr is not None:
            flags |= TQ_CONV_EMB
        if res is not None:
            flags |= TQ_CONV_RES
            assert res.C == site.C_out and res.T == T_out
        if stats:
            flags |= TQ_CONV_STATS
        if site.K == 1 and (self.B <= SMALL_TILE_B or (self.solo and self.B * ((T_out + 127) // 128) <= SMALL_TILE_WGS)):
            # launch-bound plans (the rule of the small tile): the qkv projection in its channel-tiled form -- the input-stationary one
            # has ONE workgroup per 128 positions (a 16-sample plan at T = 512: 64 on 256 compute units; 37 -> 21 us)
            flags |= _lib.TQ_CONV_CH_TILES
        d.flags = flags
        d.emb_stride = self.emb_total
        d.wfmt = wfmt
        if launch:
            site.pack_mode = _lib.PACK_MODE[d.wfmt]
            if skip is not None:
                skip[1].pack_mode = site.pack_mode
            self._wfmt_sites.append((d, [site] + ([skip[1]] if skip is not None else []), d.wfmt))
            if stats:
                d.range_flag = self.range_flag.data_ptr()
        d.dropout_site = dropout_site or 0
        d.dropout_p = 0.0
        d.dropout_seed = 0
        if dropout_site is not None:
            self.dropout_descs.append(d)
        self._keep.append(d)
        flops = 2 * site.C_in * site.C_out * site.K * T_out * self.B
        nbytes = 4 * self.B * (T_in * site.C_in + T_out * site.C_out + (T_out * site.C_out if res is not None else 0))
        if skip is not None:
            ksrcs, ksite = skip
            k0, k1 = ksrcs[0], (ksrcs[1] if len(ksrcs) > 1 else None)
            assert res is None and k0.T == T_out and ksite.C_out == site.C_out
            d.C_skip0, d.C_skip1 = k0.C, (k1.C if k1 else 0)
            self._emit((self.lib.tq_conv1d_fwd_skip, (
                C.byref(d), _p(s0.buf), _p(s1.buf) if s1 else None, _p(gn[0]) if gn else None, _p(gn[1]) if gn else None,
                _p(site.packed), _p(site.bias), emb_ptr, _p(k0.buf), _p(k1.buf) if k1 else None, _p(ksite.bias),
                _p(out.buf), _p(out.stats)),
                "conv:" + site.name + "+skip", flops + 2 * ksite.C_in * site.C_out * T_out * self.B),
                nbytes=nbytes + 4 * self.B * T_out * ksite.C_in)
        elif launch:
            op = (self.lib.tq_conv1d_fwd, (
                C.byref(d), _p(s0.buf), _p(s1.buf) if s1 else None, _p(gn[0]) if gn else None, _p(gn[1]) if gn else None,
                _p(site.packed), _p(site.bias), emb_ptr, _p(res.buf) if res else None, _p(out.buf), _p(out.stats)),
                "conv:" + site.name, flops)
            infer_op = None
            # (two-phase form: its 2 * ceil(T_in / 128) statistics slots must be the tensor's ceil(2 T_in / 128))
            if (upsample and site.K == 5 and (T_in % STAT_SLOT == 0 or T_in % STAT_SLOT > STAT_SLOT // 2) and site.C_out % 32 == 0 and gn is None and res is None
                    and emb_ptr is None and os.environ.get("TQDNE_POLYPHASE_UPSAMPLE", "1") != "0"):
                self._poly_desc = None
                infer_op = self._polyphase_op(site, d, s0, s1, out, flops)
                if POLY_TRAIN and s1 is None:
                    op = infer_op   # (one form for both kinds of forward; the backward plan differentiates that form)
            if qkv_planes is not None:  # (ws, H, D): K / V straight into the attention kernel's pre-split planes
                ws, H_, D_ = qkv_planes
                # V planes in the plan's format (fp16 hi / lo unless the plan is on the fp32-range scheme): under the range guard
                d.range_flag = self.range_flag.data_ptr()
                infer_op = (self.lib.tq_conv1d_fwd_qkv, (
                    C.byref(d), _p(s0.buf), _p(gn[0]) if gn else None, _p(gn[1]) if gn else None, _p(site.packed), _p(site.bias),
                    _p(out.buf), _p(ws), H_, D_, self.kv_v_format), "conv:" + site.name + "+split", flops)
                self._vfmt_ops.append(len(self.ops_infer))
            self._emit(op, infer_op, nbytes=nbytes)
            if stats and qkv_planes is None:   # (its inference form, the two-phase up-sampling conv, is a launch of its own shape)
                poly = getattr(self, "_poly_desc", None)
                out.prod = [d] + ([poly] if (infer_op is not None and poly is not None) else [])
                self._poly_desc = None
        if launch and skip is not None and stats:
            out.prod = [d]
        self.last_rec = ConvRec(site, d, list(srcs), gn, out, stride, upsample, silu, dropout_site is not None)
        if launch and upsample and POLY_TRAIN and s1 is None and self.poly_sites and self.poly_sites[-1][1] is site:
            self.last_rec.poly = (self.poly_sites[-1][0], self._last_poly_desc)
        return out

    def _polyphase_op(self, site: ConvSite, d: TqConvDesc, s0: Act, s1: Optional[Act], out: Act, flops: int):
        """Inference form of Upsample (blocks.py:56-66: F.interpolate(nearest, x2), then conv k = 5): both output phases as ONE k = 3
        conv over the un-upsampled rows (TQ_CONV_POLY2) -- even outputs use the taps (w0+w1, w2+w3, w4), odd ones (w0, w1+w2,
        w3+w4): 3/5 of the multiply-adds, same result up to the fp32 rounding of the tap sums.  The training forward keeps the k = 5
        launch its gradients are written for.  The two-phase weights are rebuilt by repack() with the packed fragments."""
        Cr = site.C_out
        w2 = self.store.entry("w:" + site.name + ":polyphase", (2 * Cr, site.C_in, 3), torch.float32)["buf"]
        ps = ConvSite(site.name + ":polyphase", w2, site.bias, self.dev, self.lib, store=self.store)
        self.poly_sites.append((ps, site))
        d2 = TqConvDesc()
        d2.B, d2.T_in, d2.T_out = d.B, d.T_in, d.T_in
        d2.C_in0, d2.C_in1, d2.C_out = d.C_in0, d.C_in1, 2 * Cr
        d2.ktaps, d2.stride, d2.pad, d2.upsample = 3, 1, 1, 0
        d2.flags = (d.flags & TQ_CONV_STATS) | _lib.TQ_CONV_POLY2
        d2.emb_stride = 0
        d2.wfmt = _lib.forward_wfmt(2 * Cr, [d.C_in0, d.C_in1])
        ps.pack_mode = _lib.PACK_MODE[d2.wfmt]
        self._wfmt_sites.append((d2, [ps], d2.wfmt))
        if d2.flags & TQ_CONV_STATS:
            d2.range_flag = self.range_flag.data_ptr()
        self._keep.append(d2)
        self._poly_desc = d2
        self._last_poly_desc = d2
        return (self.lib.tq_conv1d_fwd, (
            C.byref(d2), _p(s0.buf), _p(s1.buf) if s1 else None, None, None, _p(ps.packed), _p(site.bias), None, None,
            _p(out.buf), _p(out.stats)), "conv:" + site.name + "+polyphase", flops)

    # ------------------------------------------------------------------ graph construction
    def _build(self):
        m, B, T = self.m, self.B, self.T
        lib = self.lib
        # all ResBlocks, in execution order, for the batched embedding projection
        self.res_blocks = []
        for blk in list(m.input_blocks) + [m.middle_block] + list(m.output_blocks):
            for layer in blk:
                if getattr(layer, "kind", None) == "res":
                    self.res_blocks.append(layer)
        self.emb_offsets = {}
        off = 0
        for rb in self.res_blocks:
            self.emb_offsets[id(rb)] = off
            off += rb.out_channels
        self.emb_total = off
        self.emb = self._empty(B, self.E)
        self.silu_emb = self._empty(B, self.E)
        self.emb_hidden = self._empty(B, 2, self.E)
        self.emb_w = self.store.entry("emb_w", (self.emb_total, self.E), torch.float32)["buf"]
        self.emb_b = self.store.entry("emb_b", (self.emb_total,), torch.float32)["buf"]
        self.emb_entry = None
        # all 22 per-block Linear(SiLU(emb)) projections (unet.py:91-97) as ONE pointwise "conv" on the MFMA path: the B samples
        # are the positions of a single (1, B, E) channels-last sequence, the concatenated weight a (emb_total, E, 1) kernel
        self.emb_desc = None
        if self.emb_total > 0 and self.E % 32 == 0 and self.emb_total % 32 == 0:
            d = TqConvDesc()
            d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = 1, B, B, self.E, 0, self.emb_total
            d.ktaps, d.stride, d.pad, d.upsample, d.flags = 1, 1, 0, 0, 0
            d.wfmt = _lib.forward_wfmt(self.emb_total, [self.E])
            self.emb_pack_mode = _lib.PACK_MODE[d.wfmt]
            self.emb_entry = self.store.entry("emb_packed", lib.tq_conv_weight_pack_bytes(self.emb_total, self.E, 1, self.emb_pack_mode))
            self.emb_packed = self.emb_entry["buf"]
            self._keep.append(d)
            self.emb_desc = d
        self.emb_all = self._empty(B, self.emb_total)

        # stem (dynamic args: x, in_scale) -------------------------------------------------------
        stem = m.input_blocks[0][0]
        self.stem_out = self._act(stem.out_channels, T, True)
        hs = [self.stem_out]
        h = self.stem_out
        self._site_counter = 0

        def run_layers(layers, h, name):
            for li, layer in enumerate(layers):
                kind = getattr(layer, "kind", None)
                pfx = f"{name}.{li}"
                if kind == "res":
                    h = self._res_block(h, layer, pfx)
                elif kind == "attn":
                    h = self._attention(h[0] if isinstance(h, tuple) else h, layer, pfx)
                elif kind == "down":
                    x_in = h
                    h = self._conv([h], self._site(pfx + ".op", layer.op), stride=2)
                    self.tape.append(("down", dict(x=x_in, out=h, rec=self.last_rec)))
                elif kind == "up":
                    x_in = h
                    h = self._conv([h], self._site(pfx + ".conv", layer.conv), upsample=True)
                    self.tape.append(("up", dict(x=x_in, out=h, rec=self.last_rec)))
                else:
                    raise RuntimeError(f"unexpected layer {type(layer)} in {name}")
            return h

        for i, blk in enumerate(m.input_blocks):
            if i == 0:
                continue
            h = run_layers(blk, h, f"input_blocks.{i}")
            hs.append(h)
        h = run_layers(m.middle_block, h, "middle_block")
        for i, blk in enumerate(m.output_blocks):
            skip = hs.pop()
            h = run_layers(blk, (h, skip), f"output_blocks.{i}")
        self.final = h
        _check_head_limits(h.C, m.out[2].out_channels, m.out[2].kernel_size[0])
        self.head_gn = self._gn([h], m.out[0])
        self.out_nct = self._empty(B, m.out_channels, T)
        assert getattr(self, "_pending_gn", None) is None, "a deferred GroupNorm finalisation was never placed"

    def _res_block(self, x, rb, name: str) -> Act:
        srcs = list(x) if isinstance(x, tuple) else [x]
        emb_ptr = self.emb_all.data_ptr() + 4 * self.emb_offsets[id(rb)] if hasattr(rb, "emb_layers") else None
        g1 = self._gn(srcs, rb.in_layers[0], defer=not self.ckpt)   # (use_checkpoint plans keep their recompute lists: no consumer-side fold)
        i_conv1 = len(self.ops)
        h1 = self._conv(srcs, self._site(name + ".in_layers.2", rb.in_layers[2]), gn=g1, silu=True, emb_ptr=emb_ptr, ckpt_tag="h1")
        rec1 = self.last_rec
        i_gn2 = len(self.ops)
        g2 = self._gn([h1], rb.out_layers[0], defer=not self.ckpt)
        # use_checkpoint: h1 and its statistics live in a shared buffer; the backward re-issues [conv1, GroupNorm 2's fold] first
        recompute = list(range(i_conv1, len(self.ops))) if self.ckpt else None
        rec_sk = None
        conv2 = rb.out_layers[3]
        self._site_counter += 1
        if isinstance(rb.skip_connection, torch.nn.Identity):
            assert len(srcs) == 1
            out = self._conv([h1], self._site(name + ".out_layers.3", conv2), gn=g2, silu=True, res=srcs[0],
                             dropout_site=self._site_counter)
        elif FUSE_SKIP and conv2.weight.shape[2] == 5 and conv2.weight.shape[0] % FUSE_SKIP_CO == 0 and all(a.C % 32 == 0 for a in srcs):
            # the 1x1 skip conv rides in conv2's launch (extra K chunks); it is still recorded for its gradients
            site2, site_sk = self._site_pair(name + ".out_layers.3", conv2, name + ".skip_connection", rb.skip_connection)
            self._conv(srcs, site_sk, stats=False, launch=False)
            rec_sk = self.last_rec
            out = self._conv([h1], site2, gn=g2, silu=True, dropout_site=self._site_counter, skip=(srcs, site_sk))
        else:
            res = self._conv(srcs, self._site(name + ".skip_connection", rb.skip_connection), stats=False)
            rec_sk = self.last_rec
            out = self._conv([h1], self._site(name + ".out_layers.3", conv2), gn=g2, silu=True, res=res,
                             dropout_site=self._site_counter)
        self.tape.append(("res", dict(rb=rb, srcs=srcs, g1=g1, g2=g2, h1=h1, out=out, rec1=rec1, rec2=self.last_rec,
                                      rec_sk=rec_sk, recompute=recompute)))
        return out

    def _attention(self, x: Act, ab, name: str) -> Act:
        g = self._gn([x], ab.norm)
        D = ab.channels // ab.num_heads
        if D not in (32, 64, 128):
            raise NotImplementedError(f"attention head dim {D} (kernels exist for 32, 64 and 128)")
        ws = self._attn_workspace(self.lib.tq_attention_workspace_bytes(self.B, x.T, ab.num_heads, D))
        # inference forwards: the qkv projection writes K / V as the attention kernel's bf16 hi / lo planes itself (no fp32 K / V,
        # no split pass: -134 MB and one launch per block); forwards a backward may follow keep fp32 qkv for tq_attention_bwd
        split = (ws, ab.num_heads, D) if D in (32, 64) else None
        i_qkv = len(self.ops)
        qkv = self._conv([x], self._site(name + ".qkv", ab.qkv), gn=g, silu=False, stats=False, qkv_planes=split, ckpt_tag="qkv")
        rec_qkv = self.last_rec
        att = self._ckpt_act("att", ab.channels, x.T, False) if self.ckpt else self._act(ab.channels, x.T, False)
        if self.ckpt:
            lse = self._ckpt_pool.get(("lse", ab.num_heads, x.T))
            if lse is None:
                lse = self._ckpt_pool[("lse", ab.num_heads, x.T)] = self._empty(self.B, ab.num_heads, x.T)
        else:
            lse = self._empty(self.B, ab.num_heads, x.T)
        flops = 4 * ab.channels * x.T * x.T * self.B
        op = (self.lib.tq_attention_fwd, (_p(qkv.buf), _p(att.buf), _p(lse), _p(ws), self.B, x.T, ab.num_heads, D), "attention", flops)
        if split is not None and not self.ckpt:   # (see enable_block_kv: once a backward plan exists this launch gets a workspace of its own)
            self._attn_train_ops.append((len(self.ops), len(self.tape), ws.numel()))
        infer_op = None
        if split is not None:
            infer_op = (self.lib.tq_attention_fwd_presplit, (_p(qkv.buf), _p(ws), _p(att.buf), self.B, x.T, ab.num_heads, D,
                                                             self.kv_v_format), "attention", flops)
            self._vfmt_ops.append(len(self.ops_infer))
        self._emit(op, infer_op, nbytes=4 * self.B * x.T * 4 * ab.channels)
        # use_checkpoint: qkv, the attention output, the log-sum-exp and the K / V planes live in shared buffers; the backward re-issues
        # [qkv projection, attention core] first -- which also leaves THIS block's K / V planes in the shared workspace
        recompute = list(range(i_qkv, len(self.ops))) if self.ckpt else None
        out = self._conv([att], self._site(name + ".proj_out", ab.proj_out), res=x)
        entry = dict(ab=ab, x=x, g=g, qkv=qkv, att=att, lse=lse, out=out, rec_qkv=rec_qkv, rec_proj=self.last_rec, D=D,
                     recompute=recompute)
        if self.ckpt and split is not None:
            entry["kv_ws"], entry["kv_always"] = ws, True
        self.tape.append(("attn", entry))
        return out

    def _attn_workspace(self, nbytes: int):
        """pre-split K/V scratch, shared by the attention blocks of the plan that have one shape (they run back to back on one
        stream); a block that needs more (a Decoder with attention at several resolutions: T grows along up_blocks) gets a
        buffer of its own -- launches already emitted keep the pointer they were bound to"""
        ws = getattr(self, "_attn_ws", None)
        if ws is None or ws.numel() != nbytes:
            ws = torch.zeros(nbytes, dtype=torch.uint8, device=self.dev)  # padding rows (t >= T) stay zero
            self._keep.append(ws)
            self._attn_ws = ws
        return ws

    # ------------------------------------------------------------------ measurement
    def install_probe(self, name_prefix: str = "conv:"):
        """Bracket the heaviest launch (by algorithmic FLOP) with HIP events on the launch stream, every eager forward."""
        idx = max((i for i, op in enumerate(self.ops) if op[2].startswith(name_prefix)), key=lambda i: self.ops[i][3])
        self._probe = Probe(idx, self.ops[idx][2], self.ops[idx][3])
        return self._probe

    def _poison(self):
        """test mode (``plan.poison_gn = True``): NaN into every GroupNorm coefficient buffer before a forward, so that a
        fold that is skipped, raced or mis-addressed is a loud NaN in the output instead of the previous call's coefficients"""
        if self.poison_gn and not torch.cuda.is_current_stream_capturing():
            for t in self.gn_bufs:
                t.fill_(float("nan"))
            for a in self.acts:   # (the partial statistics too: a fold that ran before its sample was complete reads NaN)
                if a.stats is not None:
                    a.stats.fill_(float("nan"))

    # ------------------------------------------------------------------ range guard of the fp16-range scheme
    def _set_scheme_bf16x3(self):
        """Move every forward launch of the plan to the fp32-range three-product scheme (descriptors are edited in place, the
        weights are re-packed in that format on the next forward)."""
        for d, sites, _pref in self._wfmt_sites:
            d.wfmt = _lib.TQ_WFMT_BF16X3
            for st in sites:
                st.pack_mode = 0
        for i, fn, d in getattr(self, "_fold_default_ops", ()):   # default-tile convs that folded their own GroupNorm get the launch back
            for ops in (self.ops, self.ops_infer):
                _f, a_, w_, fl_ = ops[i]
                ops[i] = (fn, a_, "gn_finalize", fl_)
            d.gn_fold = None
        self._fold_default_ops = []
        if getattr(self, "emb_desc", None) is not None:
            self.emb_desc.wfmt = _lib.TQ_WFMT_BF16X3
            self.emb_pack_mode = 0
        # the inference attention pair leaves the fp16 V planes with the convs (its v_format is the last integer of both calls)
        self.set_kv_v_format(_lib.TQ_KV_V_BF16)
        self.scheme = "bf16x3"            # (the pack mode is part of every store entry's version tag: the next forward re-packs)
        self.plan_epoch += 1     # captured HIP graphs of this plan are stale

    def enable_block_kv(self):
        """Called once a backward plan exists: every attention block's TRAINING forward gets a K / V workspace of its own (the shared one
        is overwritten by the next block), so that the backward re-uses those planes instead of re-deriving them from qkv
        (tq_attention_bwd_ws_kv).  Forwards already run keep nothing: ``_last["block_kv"]`` says which kind the last one was."""
        if self._block_kv:
            return
        for i, ti, nbytes in self._attn_train_ops:
            ws = torch.zeros(nbytes, dtype=torch.uint8, device=self.dev)   # (padding rows t >= T stay zero)
            self._keep.append(ws)
            fn, args, *rest = self.ops[i]
            # (only D = 32 / 64 blocks are listed: their inference op is the pre-split launch, another entry point with another workspace)
            self.ops[i] = (fn, args[:3] + (ws.data_ptr(),) + args[4:], *rest)
            self.tape[ti][1]["kv_ws"] = ws
        self._block_kv = True

    def set_kv_v_format(self, fmt: int):
        """V planes of the inference attention pair (``_lib.TQ_KV_V_*``; the last integer argument of both launches of the pair)."""
        if fmt != self.kv_v_format:
            self.kv_v_format = fmt
            for i in self._vfmt_ops:
                fn, args, *rest = self.ops_infer[i]
                self.ops_infer[i] = (fn, tuple(args[:-1]) + (fmt,), *rest)
            self.plan_epoch += 1     # captured HIP graphs of this plan are stale

    def _range_fallback(self):
        import warnings
        warnings.warn("tqdne_amd: activations approach the fp16 range (a tensor's 128-position sum of squares reached (65504/2)^2); "
                      "the forward convolutions of this model now run in the fp32-range bf16x3 scheme", RuntimeWarning)
        self.range_flag.zero_()
        self.m._conv_scheme = "bf16x3"           # plans built later start there
        for eng in getattr(self.m, "_engine_cache", {}).values():
            if eng.scheme == "auto":
                eng._set_scheme_bf16x3()
        if self.scheme == "auto":
            self._set_scheme_bf16x3()

    def check_range(self) -> bool:
        """Read the range-guard flag now (synchronises).  True: the flag was set -- the plan has been moved to bf16x3 and the
        caller should repeat the computation whose launches raised it."""
        if self.scheme != "auto":
            return False
        if int(self.range_flag.item()) != 0:
            self._range_fallback()
            return True
        return False

    def _range_poll(self, begin: bool):
        """Deferred form for loops that must not synchronise (training steps): the flag is copied to pinned host memory after a
        forward and looked at before a later one, once the copy has completed."""
        if self.scheme != "auto" or torch.cuda.is_current_stream_capturing():
            return
        if begin:
            if self._range_evt is not None and self._range_evt.query():
                self._range_evt = None
                if int(self._range_host[0]) != 0:
                    self._range_fallback()
        elif self._range_evt is None:
            self._range_host.copy_(self.range_flag, non_blocking=True)
            self._range_evt = torch.cuda.Event()
            self._range_evt.record()

    # ------------------------------------------------------------------ weights
    def _stale(self):
        """(conv sites, polyphase sites, emb?) whose store buffers were not built from the current weights / pack mode"""
        sites = []
        for st in self.conv_sites:
            if st.entry is None:
                continue  # tail of a pair: packed with its head
            ver = (id(st.weight), st.weight._version, st.pack_mode) + ((id(st.tail.weight), st.tail.weight._version) if st.tail else ())
            if st.entry["ver"] != ver:
                sites.append((st, ver))
        polys = []
        for ps, src in self.poly_sites:
            ver = (id(src.weight), src.weight._version, ps.pack_mode)
            if ps.entry["ver"] != ver:
                polys.append((ps, src, ver))
        emb = None
        if self.emb_total > 0:
            ver = tuple((id(rb.emb_layers[1].weight), rb.emb_layers[1].weight._version, rb.emb_layers[1].bias._version)
                        for rb in self.res_blocks if hasattr(rb, "emb_layers")) + (getattr(self, "emb_pack_mode", 0),)
            e = self.store.entries["emb_w"]
            if e["ver"] != ver:
                emb = ver
        return sites, polys, emb

    def repack(self, stream: int, force: bool = False):
        """(Re)build the packed conv weights and the concatenated emb projection in the model's PackedStore where the parameters
        (or the contraction scheme) changed since they were built -- by whichever plan runs first after the change."""
        store, lib = self.store, self.lib
        capturing = torch.cuda.is_current_stream_capturing()
        # cheap test first: nothing moved since this plan last found every buffer current (sum of the version counters, the
        # store's pack generation, this plan's scheme)
        v = 0
        for st in self.conv_sites:
            v += st.weight._version
        for rb in self.res_blocks:
            if hasattr(rb, "emb_layers"):
                v += rb.emb_layers[1].weight._version + rb.emb_layers[1].bias._version
        tag = (v, store.gen, self.plan_epoch, stream)
        if not force and tag == self._clean_tag:
            return
        sites, polys, emb = self._stale()
        if (not capturing and store.event is not None and store.pack_stream != stream
                and self._seen_pack.get(stream) is not store.event):
            torch.cuda.current_stream(self.dev).wait_event(store.event)   # packed on another stream: order this one behind it
            self._seen_pack[stream] = store.event
        if sites or polys or emb is not None:
            cur = torch.cuda.current_stream(self.dev)
            if not capturing:
                for sid, ev in store.users.items():   # nobody may still be reading the fragments about to be overwritten
                    if sid != stream:
                        cur.wait_event(ev)
            jobs = []
            for st, ver in sites:
                jobs.append((st.weight.data_ptr(), st.packed.data_ptr(), st.C_out, st.C_in, st.K, st.pack_mode))
                if st.tail is not None:
                    t = st.tail
                    jobs.append((t.weight.data_ptr(), t.packed.data_ptr(), t.C_out, t.C_in, t.K, t.pack_mode))
                st.entry["ver"] = ver
            if emb is not None:   # gather of the ResBlocks' embedding projections into the concatenated (emb_total, E) buffers
                for rb in self.res_blocks:
                    if hasattr(rb, "emb_layers"):
                        o = self.emb_offsets[id(rb)]
                        w, b_ = rb.emb_layers[1].weight, rb.emb_layers[1].bias
                        jobs.append((w.data_ptr(), self.emb_w.data_ptr() + 4 * o * self.E, w.numel(), 0, 0, 4))
                        jobs.append((b_.data_ptr(), self.emb_b.data_ptr() + 4 * o, b_.numel(), 0, 0, 4))
            pack_batch(lib, self.dev, jobs, stream, capturing)
            with torch.no_grad():
                for ps, src, ver in polys:  # two-phase k = 3 restatement of the upsampling convs (see _polyphase_op)
                    w, Cr = src.weight, src.C_out
                    ps.weight[:Cr, :, 0] = w[:, :, 0] + w[:, :, 1]
                    ps.weight[:Cr, :, 1] = w[:, :, 2] + w[:, :, 3]
                    ps.weight[:Cr, :, 2] = w[:, :, 4]
                    ps.weight[Cr:, :, 0] = w[:, :, 0]
                    ps.weight[Cr:, :, 1] = w[:, :, 1] + w[:, :, 2]
                    ps.weight[Cr:, :, 2] = w[:, :, 3] + w[:, :, 4]
                    check(lib.tq_pack_conv_weight(ps.weight.data_ptr(), ps.C_out, ps.C_in, 3, ps.pack_mode, ps.packed.data_ptr(), stream),
                          "pack " + ps.name)
                    ps.entry["ver"] = ver
                if emb is not None:
                    if getattr(self, "emb_desc", None) is not None:   # (reads emb_w: stream-ordered behind the gather above)
                        check(lib.tq_pack_conv_weight(self.emb_w.data_ptr(), self.emb_total, self.E, 1, self.emb_pack_mode,
                                                      self.emb_packed.data_ptr(), stream), "pack emb projections")
                    store.entries["emb_w"]["ver"] = emb
            store.gen += 1
            if not capturing:
                store.event = torch.cuda.Event()
                store.event.record(cur)
                store.pack_stream = stream
        self._clean_tag = (v, store.gen, self.plan_epoch, stream)

    def _mark_use(self, stream: int):
        """record, behind the launches of this forward, that ``stream`` has read the store (see repack)"""
        if torch.cuda.is_current_stream_capturing():
            return
        ev = self.store.users.get(stream)
        if ev is None:
            ev = self.store.users[stream] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))

    def repack_transposed(self, stream: int):
        """Transposed / tap-flipped fragments for the data-gradient launches (training only)."""
        v = (sum(s.weight._version for s in self.dgrad_sites), sum(s.pack_mode_t for s in self.dgrad_sites))
        if v == self._wt_version:
            return
        pack_batch(self.lib, self.dev, [(s.weight.data_ptr(), s.packed_t.data_ptr(), s.C_out, s.C_in, s.K, s.pack_mode_t)
                                        for s in self.dgrad_sites], stream, torch.cuda.is_current_stream_capturing())
        self._wt_version = v

    # ------------------------------------------------------------------ run
    def forward(self, x, timesteps, cond=None, *, in_scale=None, c_out=None, c_skip=None, skip_src=None,
                train: bool = False, dropout_seed: int = 0, infer: bool = False):
        """Run the UNet.  Returns the static (B, C_out, T) output buffer (overwritten by the next call).
        ``infer``: no backward will follow this forward -- launches may skip what only the backward reads (``ops_infer``)."""
        m, lib, B, T = self.m, self.lib, self.B, self.T
        if tuple(x.shape) != (B, m.in_channels, T):
            raise ValueError(f"plan was built for {(B, m.in_channels, T)}, got {tuple(x.shape)}")
        x = x.contiguous()
        timesteps = timesteps.contiguous().float()
        if timesteps.shape != (B,):
            raise ValueError("timesteps must have shape (N,)")
        ncond = 0
        if cond is not None:
            cond = cond.contiguous().float()
            ncond = cond.shape[1]
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        self._range_poll(True)
        self._poison()
        self.repack(stream)
        p = float(m.dropout) if train else 0.0
        for d in self.dropout_descs:
            if p > 0.0:
                d.flags |= TQ_CONV_DROPOUT
                d.dropout_p, d.dropout_seed = p, dropout_seed
            else:
                d.flags &= ~TQ_CONV_DROPOUT
        trace = None if torch.cuda.is_current_stream_capturing() else self._trace
        ev = (lambda: _recorded_event()) if trace is not None else None
        e0 = ev() if ev else None
        tm, cm = m.time_mlp, (m.cond_mlp if m.cond_features is not None else None)
        check(lib.tq_embed_fwd(
            _p(timesteps), _p(cond), _p(m.time_embed.W), _p(tm[0].weight), _p(tm[0].bias), _p(tm[2].weight), _p(tm[2].bias),
            _p(cm[0].weight) if cm else None, _p(cm[0].bias) if cm else None, _p(cm[2].weight) if cm else None,
            _p(cm[2].bias) if cm else None, _p(self.emb), _p(self.silu_emb), _p(self.emb_hidden), B, m.model_channels,
            ncond, stream), "embed")
        if self.emb_desc is not None:
            check(lib.tq_conv1d_fwd(C.byref(self.emb_desc), _p(self.silu_emb), None, None, None, _p(self.emb_packed), _p(self.emb_b),
                                    None, None, _p(self.emb_all), None, stream), "emb projections")
        else:
            check(lib.tq_linear_fwd(_p(self.silu_emb), _p(self.emb_w), _p(self.emb_b), _p(self.emb_all), B, self.E,
                                    self.emb_total, stream), "emb projections")
        if ev:
            e1 = ev()
            trace.append(("embed", 2 * B * self.E * (self.emb_total + 2 * self.E), 4 * (self.emb_total * self.E + B * self.emb_total), e0, e1))
            e0 = e1
        stem = m.input_blocks[0][0]
        check(lib.tq_stem_conv_fwd(_p(x), _p(in_scale), _p(stem.weight), _p(stem.bias), _p(self.stem_out.buf),
                                   _p(self.stem_out.stats), B, m.in_channels, T, stem.out_channels, stem.kernel_size[0],
                                   stream), "stem conv")
        if ev:
            e1 = ev()
            trace.append(("stem", 2 * B * T * m.in_channels * stem.out_channels * stem.kernel_size[0],
                          4 * B * T * (m.in_channels + stem.out_channels), e0, e1))
        probe = self._probe
        ops = self.ops_infer if (infer and not train) else self.ops
        if trace is not None:
            for i, (fn, args, what, fl) in enumerate(ops):
                a = ev()
                rc = fn(*args, stream)
                b = ev()
                trace.append((what, fl, self.op_bytes[i], a, b))
                if rc:
                    check(rc, what)
        elif probe is None or torch.cuda.is_current_stream_capturing():
            for fn, args, what, _ in ops:
                rc = fn(*args, stream)
                if rc:
                    check(rc, what)
        else:
            for i, (fn, args, what, _) in enumerate(ops):
                if i == probe.idx:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = fn(*args, stream)
                    e1.record()
                    probe.events.append((e0, e1))
                else:
                    rc = fn(*args, stream)
                if rc:
                    check(rc, what)
        self._fwd_count = getattr(self, "_fwd_count", 0) + 1
        self._last = dict(x=x, in_scale=in_scale, c_out=c_out, timesteps=timesteps, cond=cond, train=train,
                          dropout_p=p, dropout_seed=dropout_seed, infer=infer and not train, block_kv=self._block_kv)
        head = m.out[2]
        e0 = ev() if ev else None
        check(lib.tq_head_conv_fwd(_p(self.final.buf), _p(self.head_gn[0]), _p(self.head_gn[1]), _p(head.weight),
                                   _p(head.bias), _p(c_out), _p(c_skip), _p(skip_src), _p(self.out_nct), B, T,
                                   self.final.C, m.out_channels, head.kernel_size[0], stream), "head conv")
        if ev:
            trace.append(("head", 2 * B * T * self.final.C * m.out_channels * head.kernel_size[0],
                          4 * B * T * (self.final.C + 2 * m.out_channels), e0, ev()))
        if train:
            self._range_poll(False)
        self._mark_use(stream)
        return self.out_nct

    def release(self):
        """Called by the bounded plan cache when this plan is evicted (after the device has been synchronised): the plan and its
        backward plan reference each other (``_bwd`` <-> ``BackwardPlan.e``, and the backward plan's attention closures hold the
        plan), so without this the activation, gradient and scratch buffers -- several GB at B = 64 -- would outlive the eviction
        until a cyclic-GC pass, which the caching allocator never asks for.  The plan stays usable for whoever still holds it (an
        autograd graph whose backward has not run yet: ``ctx.eng``): a later ``backward`` builds a fresh backward plan."""
        bwd, self._bwd = self._bwd, None
        if bwd is not None:
            bwd.e = None
            bwd.ops = []

    # ------------------------------------------------------------------ backward
    def backward(self, dpred: torch.Tensor, gloss: torch.Tensor, c_out=None, in_scale=None, clone: bool = True, on_bucket=None,
                 bucket_elems: int = 4 << 20, tail_fill=None, want_dx: bool = False):
        """Gradients of every UNet parameter for d loss / d pred = gloss * dpred, for the last train-mode forward.
        Returns a list aligned with ``model.parameters()`` (None for frozen parameters).  ``on_bucket``: see BackwardPlan.run.
        ``want_dx``: the gradient with respect to the forward's input as well (``self._bwd.last_dx``)."""
        if self._bwd is None:
            self._bwd = BackwardPlan(self)
        return self._bwd.run(dpred, gloss, clone=clone, on_bucket=on_bucket, bucket_elems=bucket_elems, tail_fill=tail_fill,
                             want_dx=want_dx)


class SeqEngine(UNetEngine):
    """Plan for the VAE Encoder / Decoder (blocks.py:263-436): NCW stem conv -> sequence of un-conditioned ResBlocks /
    attention / down / up blocks -> output conv -> NCW.  Reuses the UNet plan's op builders (no embedding, no skip stack)."""

    def _build(self):
        m, B, T = self.m, self.B, self.T
        self.res_blocks, self.emb_offsets, self.emb_total = [], {}, 0
        self._site_counter = 0
        stem = m.input_layer
        if stem.in_channels > 16:
            raise NotImplementedError("input layer with more than 16 channels")
        self.stem_out = self._act(stem.out_channels, T, True)
        h = self.stem_out
        blocks = getattr(m, m.blocks_attr)
        for li, layer in enumerate(blocks):
            kind = getattr(layer, "kind", None)
            pfx = f"{m.blocks_attr}.{li}"
            if kind == "res":
                h = self._res_block(h, layer, pfx)
            elif kind == "attn":
                h = self._attention(h, layer, pfx)
            elif kind == "down":
                x_in = h
                h = self._conv([h], self._site(pfx + ".op", layer.op), stride=2)
                self.tape.append(("down", dict(x=x_in, out=h, rec=self.last_rec)))
            elif kind == "up":
                x_in = h
                h = self._conv([h], self._site(pfx + ".conv", layer.conv), upsample=True)
                self.tape.append(("up", dict(x=x_in, out=h, rec=self.last_rec)))
            else:
                raise RuntimeError(f"unexpected layer {type(layer)}")
        self.final = h
        out = m.output_layer
        self.out_nct = self._empty(B, out.out_channels, h.T)
        if out.out_channels <= 16:
            self.out_mode = "head"
        else:  # wide output (encoder: 2 x latent channels): fused conv to channels-last, then a layout flip
            self.out_mode = "conv"
            self.out_btc = self._conv([h], self._site("output_layer", out), stats=False)
            self.out_rec = self.last_rec

    def forward(self, x, train: bool = False, dropout_seed: int = 0):
        m, lib, B, T = self.m, self.lib, self.B, self.T
        if tuple(x.shape) != (B, m.in_channels, T):
            raise ValueError(f"plan was built for {(B, m.in_channels, T)}, got {tuple(x.shape)}")
        x = x.contiguous()
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        self._poison()
        self.repack(stream)
        p = float(getattr(m, "dropout", 0.0)) if train else 0.0
        for d in self.dropout_descs:
            if p > 0.0:
                d.flags |= TQ_CONV_DROPOUT
                d.dropout_p, d.dropout_seed = p, dropout_seed
            else:
                d.flags &= ~TQ_CONV_DROPOUT
        self._last = dict(x=x, train=train, dropout_p=p, dropout_seed=dropout_seed, block_kv=self._block_kv)
        stem = m.input_layer
        trace = None if torch.cuda.is_current_stream_capturing() else self._trace   # (measurement only: HIP events around every launch)
        ev = _recorded_event if trace is not None else None
        e0 = ev() if ev else None
        check(lib.tq_stem_conv_fwd(_p(x), None, _p(stem.weight), _p(stem.bias), _p(self.stem_out.buf), _p(self.stem_out.stats), B,
                                   m.in_channels, T, stem.out_channels, stem.kernel_size[0], stream), "input layer")
        if ev:
            trace.append(("input layer", 2 * B * T * m.in_channels * stem.out_channels * stem.kernel_size[0],
                          4 * B * T * (m.in_channels + stem.out_channels), e0, ev()))
        for i, (fn, args, what, fl) in enumerate(self.ops):
            a = ev() if ev else None
            rc = fn(*args, stream)
            if ev:
                trace.append((what, fl, self.op_bytes[i], a, ev()))
            if rc:
                check(rc, what)
        out = m.output_layer
        e0 = ev() if ev else None
        if self.out_mode == "head":
            check(lib.tq_head_conv_fwd(_p(self.final.buf), None, None, _p(out.weight), _p(out.bias), None, None, None,
                                       _p(self.out_nct), B, self.final.T, self.final.C, out.out_channels, out.kernel_size[0],
                                       stream), "output layer")
        else:
            self.out_nct.copy_(self.out_btc.buf.permute(0, 2, 1))  # (B,T,C) -> (B,C,T): 1/60 of the encoder's traffic
        if ev:
            trace.append(("output layer" if self.out_mode == "head" else "output layout flip", 2 * B * self.final.T * self.final.C * out.out_channels
                          * out.kernel_size[0] if self.out_mode == "head" else 0, 4 * B * self.final.T * (self.final.C + out.out_channels), e0, ev()))
        self._mark_use(stream)
        return self.out_nct

    def backward(self, dout: torch.Tensor, want_dx: bool = False, clone: bool = True):
        """Gradients of every parameter of the Encoder / Decoder for d loss / d output = ``dout`` (B, C_out, T_out), for the
        last forward; with ``want_dx`` also d loss / d input (B, C_in, T).  Returns (list aligned with parameters(), dx|None)."""
        from .engine_bwd import SeqBackwardPlan
        if self._bwd is None:
            self._bwd = SeqBackwardPlan(self)
        return self._bwd.run_seq(dout, want_dx=want_dx, clone=clone)
