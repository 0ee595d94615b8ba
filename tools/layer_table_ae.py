"""Per-launch table of the autoencoder of BASELINE configs[3] (`get_1d_autoencoder_configs`: 3 x 16384 -> 32 x 4096 (mean | log std) and
16 x 4096 -> 3 x 16384): HIP events around every launch of the encode and of the decode plan, on the launch stream.
usage: python tools/layer_table_ae.py [B] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tqdne_amd import LightningAutoencoder
from tqdne_amd.autoencoder import _seq_engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
T = 16384
dev = torch.device("cuda:0")
torch.manual_seed(0)
ae = LightningAutoencoder(dict(bench.AE_BASE, in_channels=3, out_channels=32), dict(bench.AE_BASE, in_channels=16, out_channels=3),
                          {"learning_rate": 1e-4, "max_steps": 1000, "eta_min": 0.0})
ae.load_state_dict(bench.perturbed_state(ae, 19))
ae = ae.to(dev).eval()
g = torch.Generator().manual_seed(1)
x = (0.5 * torch.randn(B, 3, T, generator=g)).to(dev)
with torch.no_grad():
    z = ae.encode(x)
for name, mod, inp in (("encoder", ae.encoder, x), ("decoder", ae.decoder, z)):
    eng = _seq_engine(mod, inp)
    acc, order = {}, []
    for r in range(reps + 2):
        eng._trace = []
        with torch.no_grad():
            mod(inp)
        torch.cuda.synchronize()
        tr, eng._trace = eng._trace, None
        if r < 2:
            continue
        for i, (nm, fl, nb, e0, e1) in enumerate(tr):
            key = (i, nm)
            if key not in acc:
                acc[key] = [0.0, fl, nb]
                order.append(key)
            acc[key][0] += e0.elapsed_time(e1) / reps
    tot = sum(v[0] for v in acc.values())
    fl_t, nb_t = sum(v[1] for v in acc.values()), sum(v[2] for v in acc.values())
    print(f"# {name}, B={B}, {tuple(inp.shape[1:])}: {tot:.3f} ms over {len(order)} launches; {fl_t / 1e9:.1f} GFLOP, {nb_t / 1e6:.0f} MB algorithmic "
          f"-> {fl_t / 1e9 / tot / 2500:.3f} of the MFMA peak, {nb_t / 1e6 / tot / 8000:.3f} of the HBM peak")
    print(f"{'launch':58s} {'us':>8s} {'GFLOP':>8s} {'MB':>8s} {'mfma':>6s} {'hbm':>6s}")
    for key in order:
        ms, fl, nb = acc[key]
        print(f"{key[1][:58]:58s} {ms * 1e3:8.1f} {fl / 1e9:8.2f} {nb / 1e6:8.1f} {fl / 1e9 / ms / 2500 if ms else 0:6.3f} {nb / 1e6 / ms / 8000 if ms else 0:6.3f}")
