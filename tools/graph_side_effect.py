#!/usr/bin/env python3
"""Does a HIP-graph capture earlier in the process slow the 4-lane sampler down?  (developer probe, GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tqdne_amd import LightningEDM, paper_1d_unet_config
dev = torch.device("cuda:0")
torch.manual_seed(0)
edm = LightningEDM(paper_1d_unet_config(), {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0}, num_sampling_steps=18).to(dev).eval()
with torch.no_grad():
    for p in edm.unet.parameters():
        if torch.count_nonzero(p) == 0:
            p.normal_(0, 0.02)
sig = edm.edm.sampling_sigmas(18).to(dev)
g = torch.Generator().manual_seed(1)
def mk(B):
    return (torch.randn(B, 3, 4096, generator=g, dtype=torch.float64).to(dev) * sig[0], torch.randn(B, 5, generator=g).to(dev))
e64, c64 = mk(64); e4, c4 = mk(4)
def t64(tag):
    edm.sample_deterministically(e64, sig, None, c64); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); edm.sample_deterministically(e64, sig, None, c64); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{tag}: B=64 4-lane sample {sorted(ts)[1]:.1f} ms", flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
if not mode.endswith("_first"):
    t64("fresh process")
mode = mode.replace("_first", "")
if mode == "graph":
    edm.sample_deterministically(e4, sig, None, c4, use_graph=True)
elif mode == "eager4":
    edm.sample_deterministically(e4, sig, None, c4, use_graph=False)
elif mode == "rawgraph":
    x = torch.zeros(16, device=dev); gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        x.add_(1)
    gr.replay()
torch.cuda.synchronize()
t64(f"after {mode}")
