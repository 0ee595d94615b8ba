"""Per-launch table of one UNet forward (HIP events around every launch of the plan, on the launch stream):
name, time, algorithmic GFLOP and MB, fraction of the dense bf16 MFMA peak and of the HBM peak.
usage: python tools/layer_table.py [B] [T] [reps] [train]      (LAYER_TABLE_CONFIG=tiny: cfg0; LAYER_TABLE_CHANNELS=6 | 16: the paper UNet
with that many input / output channels -- the envelope representation's 6 x 4064, the latent EDM's 16 x 4096)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tqdne_amd import LightningEDM, paper_1d_unet_config, tiny_1d_unet_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
train = len(sys.argv) > 4 and sys.argv[4] == "train"
tiny = os.environ.get("LAYER_TABLE_CONFIG", "paper") == "tiny"   # (BASELINE configs[0]: unconditioned, 32 base channels)
dev = torch.device("cuda:0")
torch.manual_seed(0)
nch = int(os.environ.get("LAYER_TABLE_CHANNELS", "3"))
edm = LightningEDM(tiny_1d_unet_config() if tiny else paper_1d_unet_config(in_channels=nch, out_channels=nch),
                   {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0})
edm.unet.load_state_dict(bench.perturbed_state(edm.unet, 17))
edm = edm.to(dev)
g = torch.Generator().manual_seed(1)
x = (0.5 * torch.randn(B, 3 if tiny else nch, T, generator=g)).to(dev)
cond = None if tiny else torch.randn(B, 5, generator=g).to(dev)
sig = torch.full((B,), 0.7, device=dev)
eng = edm.unet._engine(B, T, dev)
batch = {"signal": x} if tiny else {"signal": x, "cond": cond}
acc = {}
order = []
for r in range(reps + 2):
    if train:
        edm.train()
        if r >= 1:
            eng._trace = []
            if eng._bwd is not None:
                eng._bwd._trace = []
        edm.step_and_backward(batch)
        tr = (eng._trace or []) + ((eng._bwd._trace or []) if eng._bwd is not None else [])
    else:
        edm.eval()
        eng._trace = []
        with torch.no_grad():
            edm(x, sig, None, cond)
        tr = eng._trace
    torch.cuda.synchronize()
    if r < 2:
        continue
    for i, (name, fl, nb, e0, e1) in enumerate(tr):
        key = (i, name)
        if key not in acc:
            acc[key] = [0.0, fl, nb]
            order.append(key)
        acc[key][0] += e0.elapsed_time(e1) / reps
tot = sum(v[0] for v in acc.values())
print(f"# B={B} T={T} {'train step (fwd + bwd)' if train else 'inference forward'}: {tot:.3f} ms over {len(order)} launches")
print(f"{'launch':58s} {'us':>8s} {'GFLOP':>8s} {'MB':>8s} {'mfma':>6s} {'hbm':>6s}")
for key in order:
    ms, fl, nb = acc[key]
    print(f"{key[1][:58]:58s} {ms*1e3:8.1f} {fl/1e9:8.2f} {nb/1e6:8.1f} {fl/1e9/ms/2500 if ms else 0:6.3f} {nb/1e6/ms/8000 if ms else 0:6.3f}")
