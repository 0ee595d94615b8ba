"""CPU-side checks of the product: library loads and exports the header's symbols, the module surface matches the
reference (constructor, state_dict schema, initialisation), and the hot path refuses to run without a GPU."""

import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_header_symbols():
    from tqdne_amd import _build, _lib
    _build.build(verbose=False)
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "tqdne_hip.h")).read()
    declared = set(re.findall(r"\b(tq_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/tqdne_hip.h but not exported"
    assert set(_lib.exported_symbols()) <= declared


def test_state_dict_schema_and_layout():
    from tqdne_amd import UNetModel, paper_1d_unet_config, tiny_1d_unet_config
    m = UNetModel(**paper_1d_unet_config())
    sd = m.state_dict()
    assert len(sd) == 311 and sum(p.numel() for p in m.parameters()) == 15581347
    assert sd["input_blocks.1.0.in_layers.2.weight"].shape == (64, 64, 5)
    assert sd["input_blocks.3.0.op.weight"].shape == (64, 64, 3)
    assert sd["output_blocks.2.2.conv.weight"].shape == (256, 256, 5)
    assert sd["middle_block.1.qkv.weight"].shape == (768, 256, 1)
    assert sd["time_embed.W"].shape == (32,) and not m.time_embed.W.requires_grad
    assert torch.count_nonzero(sd["out.2.weight"]) == 0  # zero_module
    t = UNetModel(**tiny_1d_unet_config())
    assert sum(p.numel() for p in t.parameters()) == 3558867


@pytest.mark.skipif(not os.path.isdir("/root/reference/tqdne"), reason="reference only exists in the build container")
def test_same_seed_same_weights_as_reference():
    import sys
    sys.path.insert(0, "/root/reference")
    from tqdne.unet import UNetModel as Ref
    from tqdne_amd import UNetModel, paper_1d_unet_config
    cfg = paper_1d_unet_config()
    torch.manual_seed(3)
    a = UNetModel(**cfg).state_dict()
    torch.manual_seed(3)
    b = Ref(**cfg).state_dict()
    assert list(a) == list(b)
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_cpu_tensors_are_refused_not_silently_computed():
    from tqdne_amd import LightningEDM, UNetModel, tiny_1d_unet_config
    m = UNetModel(**tiny_1d_unet_config())
    with pytest.raises(RuntimeError, match="HIP kernels only"):
        m(torch.zeros(1, 3, 256), torch.zeros(1))
    e = LightningEDM(tiny_1d_unet_config(), {"learning_rate": 1e-4, "max_steps": 1, "eta_min": 0})
    with pytest.raises(RuntimeError):
        e(torch.zeros(1, 3, 256), torch.ones(1))
    with pytest.raises(AssertionError, match="must specify cond"):
        m(torch.zeros(1, 3, 256), torch.zeros(1), cond=torch.zeros(1, 5))


def test_unsupported_options_fail_loudly():
    from tqdne_amd import UNetModel
    for kw in (dict(dims=3), dict(dims=1, use_scale_shift_norm=True), dict(dims=1, use_causal_mask=True),
               dict(dims=2, use_causal_mask=True)):
        with pytest.raises(NotImplementedError):
            UNetModel(3, 32, 3, 1, **kw)


def test_edm_surface():
    from tqdne_amd import EDM, LightningEDM, tiny_1d_unet_config
    e = LightningEDM(tiny_1d_unet_config(), {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0})
    for attr in ("unet", "edm", "autoencoder", "config", "optimizer_params", "num_sampling_steps", "deterministic_sampling"):
        assert hasattr(e, attr)
    assert e.num_sampling_steps == 25 and e.deterministic_sampling
    opt = e.configure_optimizers()
    assert isinstance(opt["optimizer"], torch.optim.Adam) and opt["lr_scheduler"]["interval"] == "step"
    s = EDM().sampling_sigmas(18)
    assert s.shape == (19,) and s[-1] == 0
    assert all(k.startswith("unet.") for k in e.state_dict())


def test_gpu_only_components_fail_loudly_on_cpu():
    """no CPU fallbacks: the optimizer launch and the GPU representation refuse CPU tensors instead of computing elsewhere"""
    import numpy as np
    import pytest
    import torch
    from tqdne_amd.optim import FusedAdamEMA
    from tqdne_amd.representation import Identity, MovingAverageEnvelope, Normalization

    p = torch.nn.Parameter(torch.zeros(8))
    with pytest.raises(RuntimeError):
        FusedAdamEMA([("p", p)], lr=1e-3)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            MovingAverageEnvelope().get_representation(np.zeros((3, 256), np.float32))
    x = np.arange(6.0).reshape(2, 3)
    assert Identity().invert_representation(Identity().get_representation(x)) is x
    n = Normalization(1.0, 2.0)
    assert np.allclose(n.invert_representation(n.get_representation(x)), x)


def test_sampler_lane_rule():
    import os
    from tqdne_amd.edm import sampler_lanes
    old = os.environ.pop("TQDNE_SAMPLER_LANES", None)
    try:
        assert [sampler_lanes(b) for b in (1, 8, 16, 31, 32, 48, 64, 128, 66)] == [1, 1, 1, 1, 2, 2, 4, 4, 2]
        os.environ["TQDNE_SAMPLER_LANES"] = "1"
        assert sampler_lanes(64) == 1
    finally:
        os.environ.pop("TQDNE_SAMPLER_LANES", None)
        if old is not None:
            os.environ["TQDNE_SAMPLER_LANES"] = old


def test_c_abi_argument_checks_return_error_codes_without_a_gpu():
    """entry points validate descriptors / pointers before touching the device: TQ_ERR_ARG (-1) / TQ_ERR_SHAPE (-2)"""
    import ctypes as C
    from tqdne_amd import _lib
    lib = _lib.load()
    d = _lib.TqConvDesc()
    d.B, d.T_in, d.T_out, d.C_in0, d.C_in1, d.C_out = 2, 128, 128, 64, 0, 64
    d.ktaps, d.stride, d.pad, d.upsample, d.flags = 5, 1, 2, 0, 0
    fake = 0x1000  # never dereferenced: every call below is rejected during validation
    args = lambda **kw: [C.byref(d)] + [kw.get(k, fake) for k in ("x0", "x1", "gs", "gh", "w", "bias", "emb", "res", "y", "st")] + [None]
    assert lib.tq_conv1d_fwd(None, *args()[1:]) == -1
    assert lib.tq_conv1d_fwd(*args(x0=None)) == -1
    d.C_in0 = 48
    assert lib.tq_conv1d_fwd(*args()) == -2              # channels not a multiple of 32
    d.C_in0, d.ktaps = 64, 7
    d.pad = 3
    assert lib.tq_conv1d_fwd(*args()) == -2              # unsupported kernel size
    d.ktaps, d.pad, d.flags = 5, 2, _lib.TQ_CONV_GN
    assert lib.tq_conv1d_fwd(*args(gs=None)) == -1       # GN prologue without coefficients
    d.flags, d.wfmt = 0, 7
    assert lib.tq_conv1d_fwd(*args()) == -1              # unknown weight format
    d.wfmt, d.C_skip0 = 0, 64
    assert lib.tq_conv1d_fwd(*args()) == -1              # fused-skip descriptors go through tq_conv1d_fwd_skip
    assert lib.tq_pack_conv_weight(fake, 64, 64, 5, 9, fake, None) == -1
    assert lib.tq_attention_fwd(fake, fake, None, None, 2, 128, 4, 48, None) == -2   # head dim without a kernel
    assert lib.tq_envelope_fwd(fake, fake, 2, 3, 100, 128, 1e-6, 1e-6, None) == -2   # window longer than the signal
    assert lib.tq_adam_ema_step(None, 3, 1e-3, 0.9, 0.999, 1e-8, 1.0, 0.0, 1.0, 1.0, None) == -1
    assert lib.tq_conv_weight_pack_bytes(256, 256, 5, 0) == lib.tq_conv_weight_pack_bytes(256, 256, 5, 2) > 0


def test_hardware_queue_default_is_set_before_the_device_is_touched():
    """four sampler lanes + any other live stream (RCCL's, the backward plan's) need more than ROCm's default of 4 hardware queues:
    tools/hwq_probe.py measured 163.9 -> 227.2 ms for the 18-step sample with one extra stream (tqdne_amd/__init__.py)"""
    import os
    import tqdne_amd  # noqa: F401
    assert int(os.environ["GPU_MAX_HW_QUEUES"]) >= 8


def test_plan_cache_is_bounded_and_evicts_least_recently_used_shape_groups():
    """tqdne_amd/_cache.py: the plans of a model are cached per (B, T, device, lane); the cache keeps the most recently used
    (B, T, device) GROUPS (all lanes of a shape together) and tells its owner which keys went (round-4 verdict: the cache was unbounded)."""
    from tqdne_amd._cache import PlanCache, plan_cache, PLAN_SHAPES
    gone = []
    c = PlanCache(3, group=lambda k: k[:3], on_evict=lambda items: gone.extend(k for k, _ in items))
    for lane in range(4):
        c[(16, 4096, "cuda:0", 64 + lane)] = f"lane{lane}"
    c[(64, 4096, "cuda:0", 0)] = "train"
    c[(8, 4096, "cuda:0", 0)] = "ragged"
    assert len(c) == 6 and len(c.groups()) == 3 and not gone
    assert c.get((16, 4096, "cuda:0", 65)) == "lane1"        # touches the lane group: now the most recently used
    c[(4, 4096, "cuda:0", 0)] = "new"                        # evicts the B = 64 group (least recently used), not the lanes
    assert gone == [(64, 4096, "cuda:0", 0)]
    assert c.get((64, 4096, "cuda:0", 0)) is None and (16, 4096, "cuda:0", 66) in c
    c[(2, 4096, "cuda:0", 0)] = "newer"                      # ... then the ragged one
    assert gone[-1] == (8, 4096, "cuda:0", 0) and len(c.groups()) == 3 and c.evictions == 2
    assert sorted(v for v in c.values() if v.startswith("lane")) == ["lane0", "lane1", "lane2", "lane3"]
    assert plan_cache().cap == PLAN_SHAPES >= 2
    # the model classes use it
    from tqdne_amd import UNetModel, tiny_1d_unet_config
    assert isinstance(UNetModel(**tiny_1d_unet_config())._engine_cache, PlanCache)


def test_plan_cache_hands_over_live_values_releases_them_and_postpones_under_capture():
    """round-5 advisor findings: (1) the owner's callback sees the evicted VALUES while they are still referenced (it synchronises, then
    breaks the plan <-> backward-plan cycle through ``release()``; only then are the buffers let go); (2) no eviction -- hence no
    device synchronisation -- while ``can_evict()`` says no (a stream capture is in progress): the cache runs over its cap and catches
    up at the next insertion."""
    import gc
    import weakref
    from tqdne_amd._cache import PlanCache, _sync_on_evict

    class Plan:   # a plan and its backward plan reference each other (engine.UNetEngine._bwd <-> engine_bwd.BackwardPlan.e)
        def __init__(self):
            self._bwd = type("Bwd", (), {})()
            self._bwd.e = self
            self.released = False

        def release(self):
            self._bwd.e = None
            self._bwd = None
            self.released = True

    gc.disable()
    try:
        seen = []
        c = PlanCache(2, on_evict=lambda items: (seen.extend(v.released for _, v in items), _sync_on_evict(items)))
        a = Plan()
        ra = weakref.ref(a)
        c["a"] = a
        del a
        c["b"] = Plan()
        c["c"] = Plan()          # evicts "a": its cycle is broken by release(), so it dies WITHOUT a cyclic-GC pass
        assert seen == [False] and ra() is None
        allow = [False]
        c2 = PlanCache(2, can_evict=lambda: allow[0], on_evict=_sync_on_evict)
        for k in "abcd":
            c2[k] = Plan()
        assert len(c2) == 4 and c2.evictions == 0
        allow[0] = True
        c2["e"] = Plan()
        assert len(c2) == 2 and c2.evictions == 3 and "e" in c2 and "d" in c2
    finally:
        gc.enable()


# ---------------------------------------------------------------------------------------------------------------------------------
# The three statements of every C-ABI struct -- include/tqdne_hip.h, tqdne_amd/_lib.py (ctypes) and the maintainer's stub printed
# in INTEGRATION.md -- must describe the same bytes: a struct built from a stale stub is read past its end by the library.

_C_TYPES = {"int32_t": "c_int", "uint32_t": "c_uint", "float": "c_float", "uint64_t": "c_ulong", "int": "c_int",
            "unsigned long long": "c_ulong", "double": "c_double"}


def _header_struct_fields(name):
    """[(field, ctypes type name)] of ``typedef struct <name> {...}`` in include/tqdne_hip.h (pointers -> c_void_p)"""
    import ctypes
    header = open(os.path.join(ROOT, "include", "tqdne_hip.h")).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(?:const )?((?:unsigned long long|[A-Za-z_][A-Za-z0-9_]*))\s*(\*?)\s*(.*)$", decl)
        ctype, star, names = m.group(1), m.group(2), m.group(3)
        for n in names.split(","):
            n = n.strip()
            ptr = bool(star) or n.startswith("*")
            n = n.lstrip("* ")
            fields.append((n, ctypes.c_void_p if ptr else getattr(ctypes, _C_TYPES[ctype])))
    return fields


def _layout(struct):
    import ctypes
    return [(n, getattr(struct, n).offset, getattr(struct, n).size) for n, *_ in struct._fields_], ctypes.sizeof(struct)


@pytest.mark.parametrize("name", ["TqConvDesc", "TqConvBwdDesc", "TqPackJob", "TqGnFuse", "TqGnFold"])
def test_ctypes_structs_match_the_header(name):
    import ctypes
    from tqdne_amd import _lib
    ref = type(name + "_h", (ctypes.Structure,), {"_fields_": _header_struct_fields(name)})
    assert _layout(getattr(_lib, name)) == _layout(ref)


def test_integration_md_stub_matches_the_header_and_the_binding():
    """INTEGRATION.md's ``class TqConvDesc(ctypes.Structure)`` (the stub a maintainer of the reference would copy) is executed as
    printed and held against the binding: same field names, offsets, sizes and total size."""
    import ctypes
    from tqdne_amd import _lib
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", doc, re.S)
    stub = next(b for b in blocks if "class TqConvDesc(ctypes.Structure)" in b)
    classes = re.findall(r"^(class (\w+)\(ctypes\.Structure\):.*?)(?=^\S)", stub, re.S | re.M)
    assert classes, "no ctypes.Structure in the stub"
    for src, name in classes:
        ns = {"ctypes": ctypes}
        exec(src, ns)
        assert _layout(ns[name]) == _layout(getattr(_lib, name)), name
    # the argument list the stub binds tq_conv1d_fwd with is the binding's
    n_ptr = int(re.search(r"tq_conv1d_fwd\.argtypes = \[ctypes\.POINTER\(TqConvDesc\)\] \+ \[ctypes\.c_void_p\] \* (\d+)", stub).group(1))
    assert n_ptr == len(_lib._PROTOS["tq_conv1d_fwd"][1]) - 1
