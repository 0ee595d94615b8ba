"""Which launches of the plan are corrupted by a concurrently running attention kernel?  Victim: each op of plan 0 re-run alone on
stream A (its inputs are the plan's static buffers, already populated); aggressor: plan 1's attention op on stream B."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tqdne_amd import LightningEDM, paper_1d_unet_config, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
edm = LightningEDM(paper_1d_unet_config(), {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0})
edm.unet.load_state_dict(bench.perturbed_state(edm.unet, 17))
edm = edm.to(dev).eval()
T, h = 4096, 16
g = torch.Generator().manual_seed(1)
x = (3.0 * torch.randn(h, 3, T, generator=g)).to(dev)
cond = torch.randn(h, 5, generator=g).to(dev)
sig = torch.full((h,), 2.0, device=dev)
for lane in (0, 1):
    edm._lane = lane
    with torch.no_grad():
        edm._denoise_static(x, sig, 1, cond, infer=True)
edm._lane = 0
torch.cuda.synchronize()
e0, e1 = edm.unet._engine(h, T, dev, 0), edm.unet._engine(h, T, dev, 1)

def tensors(eng):
    out = []
    for i, a in enumerate(eng.acts):
        out.append((f"act{i}.buf", a.buf))
        if a.stats is not None:
            out.append((f"act{i}.stats", a.stats))
    for j, (kind, t) in enumerate(eng.tape):
        for key in ("g1", "g2", "g"):
            if key in t and t[key] is not None:
                for q, nm in zip(t[key], ("scale", "shift", "mean_rstd")):
                    out.append((f"tape{j}.{kind}.{key}.{nm}", q))
    for q, nm in zip(eng.head_gn, ("scale", "shift", "mean_rstd")):
        out.append((f"head_gn.{nm}", q))
    return out

named = tensors(e0)
ref = [t.clone() for _, t in named]
att = [op for op in e1.ops_infer if op[2] == "attention"][0]
A, Bs = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
aggr = sys.argv[1] if len(sys.argv) > 1 else "attention"
for i, (fn, args, name, _) in enumerate(e0.ops_infer):
    bad = set()
    for rep in range(4):
        for k in range(6):
            att[0](*att[1], Bs.cuda_stream)
            fn(*args, A.cuda_stream)
        torch.cuda.synchronize()
        for (nm, t), r in zip(named, ref):
            if not torch.equal(t, r):
                bad.add(nm)
                t.copy_(r)
        torch.cuda.synchronize()
    if bad:
        print(f"victim op {i} {name}: corrupted {sorted(bad)[:4]}")
print("done")
