#!/usr/bin/env python3
"""Is the 4-lane sampler at B = 64 fed by the host fast enough?  Prints the 18-step sample time for 1, 2, 4, 8 lanes, launched eagerly
(argument 2 = 0) or with one HIP-graph replay per network evaluation (argument 2 = d; needs the `lanes = 1` override in
LightningEDM._sample_det lifted).  Round 3, same box: eager 170.1 / 161.8 / 161.0 / 261.6 ms, replayed 170.6 / 162.5 / 160.7 / 350.7 ms
-- identical up to 4 lanes, i.e. the sampler is bound by the GPU, not by the ~560 launches per evaluation round the host issues.
(The "enqueue done" column includes the sample call's closing range-flag read, a synchronisation.)  (developer probe, GPU box)"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eps = torch.randn(B, 3, 4096, generator=g, dtype=torch.float64).to(dev) * sig[0]
cond = torch.randn(B, 5, generator=g).to(dev)
mode = {"0": False, "d": "denoiser"}[sys.argv[2] if len(sys.argv) > 2 else "0"]
for lanes in (1, 2, 4, 8):
    edm.sample_deterministically(eps, sig, None, cond, lanes=lanes, use_graph=mode)
    torch.cuda.synchronize()
    rows = []
    for _ in range(3):
        t0 = time.perf_counter()
        edm.sample_deterministically(eps, sig, None, cond, lanes=lanes, use_graph=mode)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    rows.sort(key=lambda r: r[1])
    print(f"graph={mode} lanes={lanes}: host enqueue done after {rows[1][0]:.1f} ms, GPU done after {rows[1][1]:.1f} ms", flush=True)
