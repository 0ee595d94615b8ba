"""Snapshot the head conv's inputs ON ITS STREAM right before it runs, in the concurrent 4-lane setting."""
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
T, h, L = 4096, 16, 4
g = torch.Generator().manual_seed(1)
x = (3.0 * torch.randn(h, 3, T, generator=g)).to(dev)
cond = torch.randn(h, 5, generator=g).to(dev)
sig = torch.full((h,), 2.0, device=dev)
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(L - 1)]
lib = _lib.load()
orig = lib.tq_head_conv_fwd
snaps = []
cur_lane = [0]

def wrapper(*args):
    eng = edm.unet._engine(h, T, dev, cur_lane[0])
    sc = edm._scal[(h, str(dev), cur_lane[0])]
    pre = [eng.final.buf.clone(), eng.head_gn[0].clone(), eng.head_gn[1].clone(), sc.clone(), x.clone()]
    rc = orig(*args)
    snaps.append((cur_lane[0], pre, eng.out_nct.clone()))
    return rc

def fwd(lane):
    edm._lane = lane
    cur_lane[0] = lane
    with torch.no_grad():
        y = edm._denoise_static(x, sig, 1, cond, infer=True)
    edm._lane = 0
    return y

for l in range(L):
    fwd(l)
torch.cuda.synchronize()
lib.tq_head_conv_fwd = wrapper
fwd(0); torch.cuda.synchronize()
ref_lane, ref_pre, ref_out = snaps.pop()
names = ["final.buf", "gscale", "gshift", "sc", "x"]
bad = 0
for it in range(12):
    for s in streams[1:]:
        s.wait_stream(streams[0])
    for rep in range(3):
        for l, s in enumerate(streams):
            with torch.cuda.stream(s):
                fwd(l)
    torch.cuda.synchronize()
    for lane, pre, out in snaps:
        ineq = [n for n, a, b in zip(names, pre, ref_pre) if not torch.equal(a, b)]
        oeq = torch.equal(out, ref_out)
        if ineq or not oeq:
            bad += 1
            print(f"iter {it} lane {lane}: inputs differing at head time: {ineq}; output equal: {oeq}")
    snaps.clear()
print("bad:", bad)
