"""Which launch of the plan corrupts a concurrently running head conv?  Head (plan 0, stream A) against each op of plan 1 (stream B)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tqdne_amd import LightningEDM, paper_1d_unet_config, _lib
from tqdne_amd.engine import _p
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
lib = _lib.load()
e0, e1 = edm.unet._engine(h, T, dev, 0), edm.unet._engine(h, T, dev, 1)
ref = e0.out_nct.clone()
sc = edm._scal[(h, str(dev), 0)]
m = edm.unet
head = m.out[2]
A, Bs = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

def run_head(stream):
    return lib.tq_head_conv_fwd(_p(e0.final.buf), _p(e0.head_gn[0]), _p(e0.head_gn[1]), _p(head.weight), _p(head.bias), _p(sc[1]), _p(sc[2]),
                                _p(x), _p(e0.out_nct), h, T, e0.final.C, m.out_channels, head.kernel_size[0], stream)

seen = {}
for i, (fn, args, name, _) in enumerate(e1.ops_infer):
    bad = 0
    for rep in range(6):
        for k in range(8):
            fn(*args, Bs.cuda_stream)
            run_head(A.cuda_stream)
            torch.cuda.current_stream(dev)  # no-op
        torch.cuda.synchronize()
        bad += int(not torch.equal(e0.out_nct, ref))
        e0.out_nct.zero_()
    key = name.split(".")[-1] if name.startswith("conv:") else name
    if bad:
        print(f"op {i} {name}: head output corrupted in {bad} of 6 rounds")
print("done")
