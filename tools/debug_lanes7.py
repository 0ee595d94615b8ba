"""Stress: 4 plans on 4 streams, after the allocator has been churned like the test-suite does; prints details of mismatches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tqdne_amd import LightningEDM, paper_1d_unet_config
from oracle import edm as OE
dev = torch.device("cuda:0")

def make():
    torch.manual_seed(0)
    edm = LightningEDM(paper_1d_unet_config(), {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0}, num_sampling_steps=18)
    edm.unet.load_state_dict(bench.perturbed_state(edm.unet, 17))
    return edm.to(dev).eval()

# churn: what test_bench_config_parity does first (B = 64 sampler with 4 lanes, then drop everything)
if "--churn" in sys.argv:
    edm = make()
    g = torch.Generator().manual_seed(1234)
    start = torch.randn(64, 3, 4096, generator=g, dtype=torch.float64)
    cond = torch.randn(64, 5, generator=g)
    sig = OE.sampling_sigmas(OE.EDMParams(), 18)
    eps = (start * sig[0]).to(dev)
    for lanes in (4, 1):
        edm.sample_deterministically(eps, sig[:3].to(dev), None, cond.to(dev), lanes=lanes)
    torch.cuda.synchronize()
    del edm, eps
    import gc; gc.collect()

edm = make()
T, h, L = 4096, 16, 4
g = torch.Generator().manual_seed(1)
x = (3.0 * torch.randn(h, 3, T, generator=g)).to(dev)
cond = torch.randn(h, 5, generator=g).to(dev)
sig = torch.full((h,), 2.0, device=dev)
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(L - 1)]

def fwd(lane):
    edm._lane = lane
    try:
        with torch.no_grad():
            return edm._denoise_static(x, sig, 1, cond, infer=True)
    finally:
        edm._lane = 0

def tensors(eng):
    qkv = {id(t["qkv"]) for kind, t in eng.tape if kind == "attn"}
    out = [("out", eng.out_nct)]
    for i, a in enumerate(eng.acts):
        out.append((f"act{i}", a.buf[:, :, : a.C // 3] if id(a) in qkv else a.buf))
        if a.stats is not None:
            out.append((f"act{i}.stats", a.stats))
    return out

for lane in range(L):
    fwd(lane)
torch.cuda.synchronize()
e0 = edm.unet._engine(h, T, dev, 0)
ref = [t.clone() for _, t in tensors(e0)]
producer = {}
for op in e0.ops_infer:
    for a in op[1]:
        if isinstance(a, int):
            producer.setdefault(a, op[2])
names = {}
for i, a in enumerate(e0.acts):
    names[f"act{i}"] = [op[2] for op in e0.ops_infer if any(isinstance(v, int) and v == a.buf.data_ptr() for v in op[1])]
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30):
    for s in streams[1:]:
        s.wait_stream(streams[0])
    for _ in range(3):
        for lane, s in enumerate(streams):
            with torch.cuda.stream(s):
                fwd(lane)
    torch.cuda.synchronize()
    for lane in range(L):
        for (name, t), r in zip(tensors(edm.unet._engine(h, T, dev, lane)), ref):
            if not torch.equal(t, r):
                d = (t.float() - r.float()).abs()
                idx = torch.nonzero(d > 0)
                print(f"round {it} lane {lane}: {name} {tuple(t.shape)} (ops touching it: {names.get(name.split('.')[0], '?')[:3]}) max diff {float(d.max()):.3e} of {float(r.abs().max()):.3e}; "
                      f"{idx.shape[0]} elems; samples {sorted(set(idx[:, 0].tolist()))[:8]}; dim1 range {int(idx[:, 1].min())}-{int(idx[:, 1].max())}; dim2 range {int(idx[:, 2].min())}-{int(idx[:, 2].max())}")
                bad += 1
                break
print("mismatches:", bad)
