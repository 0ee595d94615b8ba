"""Localise a batch-size dependence of the forward: run the paper UNet at B = 64 and at B = 16 on the first 16 samples and report
the first intermediate activation that differs; also run-to-run determinism of each."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tqdne_amd import LightningEDM, paper_1d_unet_config

dev = torch.device("cuda:0")
torch.manual_seed(0)
edm = LightningEDM(paper_1d_unet_config(), {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0})
edm.unet.load_state_dict(bench.perturbed_state(edm.unet, 17))
edm = edm.to(dev).eval()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator().manual_seed(1)
x = (80.0 * torch.randn(64, 3, T, generator=g)).to(dev)
cond = torch.randn(64, 5, generator=g).to(dev)
sig = torch.full((64,), 80.0, device=dev)

def run(B, lane):
    edm._lane = lane
    with torch.no_grad():
        y = edm._denoise_static(x[:B].contiguous(), sig[:B].contiguous(), 1, cond[:B].contiguous(), infer=True).clone()
    eng = edm.unet._engine(B, T, dev, lane)
    torch.cuda.synchronize()
    acts = [(a.buf[:16].clone(), None if a.stats is None else a.stats[:16].clone()) for a in eng.acts]
    edm._lane = 0
    return y[:16].clone(), acts, eng

y64, a64, e64 = run(64, 0)
y64b, a64b, _ = run(64, 0)
y16, a16, e16 = run(16, 1)
y16b, a16b, _ = run(16, 1)
print("run-to-run B=64:", torch.equal(y64, y64b), " B=16:", torch.equal(y16, y16b), " B=64 vs B=16:", torch.equal(y64, y16),
      float((y64 - y16).abs().max()), float(y64.abs().max()))
names = []
for kind, t in e64.tape:
    names.append(kind)
ops64 = [op[2] for op in e64.ops_infer]
print(len(a64), "activations;", len(ops64), "ops")
bad = 0
for i, ((b64, s64), (b16, s16)) in enumerate(zip(a64, a16)):
    eq = torch.equal(b64, b16)
    seq = True if s64 is None else torch.equal(s64, s16)
    if not (eq and seq):
        d = (b64 - b16).abs()
        idx = torch.nonzero(d > 0)
        print(f"act {i}: shape {tuple(b64.shape)} buf equal {eq} stats equal {seq}; max diff {float(d.max()):.3e} of {float(b64.abs().max()):.3e}; "
              f"first diffs at {idx[:3].tolist()} count {idx.shape[0]}")
        bad += 1
        if bad > 6:
            break
# which op produced each act: acts are created in build order = op order (convs / attention outputs)
