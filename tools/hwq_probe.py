#!/usr/bin/env python3
"""How do live HIP streams and GPU_MAX_HW_QUEUES interact with the 4-lane sampler?  usage: hwq_probe.py <extra streams>
Creates (and uses once) that many extra streams BEFORE the sampler's lanes exist -- the position a backward plan's or RCCL's
stream would take -- then times the 18-step sample at B = 64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tqdne_amd import LightningEDM, paper_1d_unet_config
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_hip_unet import perturbed_state
extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda:0")
keep = []
for _ in range(extra):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        keep.append(torch.zeros(1024, device=dev) + 1)
    keep.append(s)
torch.cuda.synchronize()
torch.manual_seed(0)
edm = LightningEDM(paper_1d_unet_config(), {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0}, num_sampling_steps=18)
edm.unet.load_state_dict(perturbed_state(edm.unet, 17))
edm = edm.to(dev).eval()
B, T = 64, 4096
g = torch.Generator().manual_seed(1)
cond = torch.randn(B, 5, generator=g).to(dev)
sig = edm.edm.sampling_sigmas(18).to(dev)
eps = torch.randn(B, 3, T, generator=g, dtype=torch.float64).to(dev) * sig[0]
for _ in range(2):
    edm.sample_deterministically(eps, sig, None, cond)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t = time.perf_counter()
    edm.sample_deterministically(eps, sig, None, cond)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t))
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')} extra_streams={extra}: sample {sorted(ts)[1]:.1f} ms")
