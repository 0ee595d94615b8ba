"""Concurrency check: the same 16-sample forward on 4 plans / 4 streams at once, repeated; every intermediate activation of
every lane is compared with a reference run made alone.  Reports the first differing activation (in plan order)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tqdne_amd import LightningEDM, paper_1d_unet_config
dev = torch.device("cuda:0")
torch.manual_seed(0)
edm = LightningEDM(paper_1d_unet_config(), {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0})
edm.unet.load_state_dict(bench.perturbed_state(edm.unet, 17))
edm = edm.to(dev).eval()
T, h, L = 4096, 16, 4
g = torch.Generator().manual_seed(1)
x = (3.0 * torch.randn(h, 3, T, generator=g)).to(dev)
cond = torch.randn(h, 5, generator=g).to(dev)
sig = torch.full((h,), 2.0, device=dev)
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(L - 1)]

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
    out.append(("emb_all", eng.emb_all))
    out.append(("out_nct", eng.out_nct))
    return out


def snapshot(eng):
    return [t.clone() for _, t in tensors(eng)]


def fwd(lane):
    edm._lane = lane
    with torch.no_grad():
        y = edm._denoise_static(x, sig, 1, cond, infer=True)
    edm._lane = 0
    return y

# reference: each lane alone
ref = []
for l in range(L):
    fwd(l); torch.cuda.synchronize()
    eng = edm.unet._engine(h, T, dev, l)
    ref.append(snapshot(eng))
for l in range(1, L):
    assert all(torch.equal(a, b) for a, b in zip(ref[0], ref[l])), "lanes differ even alone"
names = []
eng0 = edm.unet._engine(h, T, dev, 0)
ptr2name = {}
for op in eng0.ops_infer:
    pass
bad_total = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    for s in streams[1:]:
        s.wait_stream(streams[0])
    for rep in range(3):
        for l, s in enumerate(streams):
            with torch.cuda.stream(s):
                fwd(l)
    torch.cuda.synchronize()
    for l in range(L):
        eng = edm.unet._engine(h, T, dev, l)
        named = tensors(eng)
        cur = [t for _, t in named]
        for i, (a, b) in enumerate(zip(cur, ref[0])):
            if not torch.equal(a, b):
                print("   ->", named[i][0])
                d = (a - b).abs()
                nzb = torch.nonzero(d.flatten(1).amax(1) > 0).flatten().tolist()[:8]
                print(f"iter {it} lane {l}: first differing activation #{i} of {len(cur)} shape {tuple(a.shape)} max diff {float(d.max()):.3e} "
                      f"(max {float(b.abs().max()):.3e}); samples {nzb}; n elems {int((d > 0).sum())}")
                bad_total += 1
                break
print("mismatching (iteration, lane) pairs:", bad_total)
print("activation order:", [(i, tuple(a.buf.shape)) for i, a in enumerate(eng0.acts)][:12], "...")
