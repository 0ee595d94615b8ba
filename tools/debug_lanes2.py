import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tqdne_amd import LightningEDM, paper_1d_unet_config
from oracle import edm as OE
dev = torch.device("cuda:0")
torch.manual_seed(0)
edm = LightningEDM(paper_1d_unet_config(), {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0}, num_sampling_steps=18)
edm.unet.load_state_dict(bench.perturbed_state(edm.unet, 17))
edm = edm.to(dev).eval()
B, T = 64, 4096
g = torch.Generator().manual_seed(1234)
start = torch.randn(B, 3, T, generator=g, dtype=torch.float64)
cond = torch.randn(B, 5, generator=g).to(dev)
sig = OE.sampling_sigmas(OE.EDMParams(), 18)
eps = (start * sig[0]).to(dev)
for nsteps in (1, 2):
    s2 = sig[: nsteps + 1].to(dev)
    r = {}
    for name, lanes in (("4a", 4), ("4b", 4), ("1a", 1), ("1b", 1), ("2a", 2)):
        r[name] = edm.sample_deterministically(eps, s2, None, cond, lanes=lanes).clone()
        torch.cuda.synchronize()
    # 16-sample sub-batches one after the other on the main stream (no concurrency)
    seq = torch.cat([edm.sample_deterministically(eps[i * 16:(i + 1) * 16].contiguous(), s2, None, cond[i * 16:(i + 1) * 16].contiguous(), lanes=1) for i in range(4)])
    torch.cuda.synchronize()
    print(f"steps {nsteps}: 4a==4b {torch.equal(r['4a'], r['4b'])}  1a==1b {torch.equal(r['1a'], r['1b'])}  4a==1a {torch.equal(r['4a'], r['1a'])}  "
          f"2a==1a {torch.equal(r['2a'], r['1a'])}  seq16==1a {torch.equal(seq, r['1a'])}  seq16==4a {torch.equal(seq, r['4a'])}")
    d = (r['4a'] - r['1a']).abs()
    nz = torch.nonzero(d.amax(dim=(1, 2)) > 0).flatten().tolist()
    print("   samples differing (4 lanes vs 1):", nz[:20], "max abs diff", float(d.max()))
