import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from tqdne_amd import UNetModel, paper_1d_unet_config
from tqdne_amd.consistency_model import LithningConsistencyModel
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = UNetModel(**paper_1d_unet_config())
with torch.no_grad():
    for p in net.parameters():
        if torch.count_nonzero(p) == 0: p.normal_(0, 0.02)
cm = LithningConsistencyModel(net).to(dev).eval()
B = 64
g = torch.Generator().manual_seed(1)
eps = torch.randn(B, 3, 4096, generator=g).to(dev); cond = torch.randn(B, 5, generator=g).to(dev)
outs = {}
for lanes in (1, 4, 2, 1, 4):
    os.environ["TQDNE_SAMPLER_LANES"] = str(lanes)
    for _ in range(2): y = cm.sample_from(eps, [], [], None, cond)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): y = cm.sample_from(eps, [], [], None, cond)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    outs[lanes] = y.clone()
    print(f"consistency 1-step, B=64, lanes {lanes}: {dt*1e3:.2f} ms = {B/dt:.0f} waveforms/s")
print("bit-identical:", torch.equal(outs[1], outs[4]), torch.equal(outs[1], outs[2]))
