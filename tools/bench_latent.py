#!/usr/bin/env python3
"""BASELINE configs[3]: 1-D latent EDM -- autoencoder encode / decode (3 x 16384 <-> 16 x 4096) + latent UNet (paper shape,
in/out 16): one latent-EDM train step and one 18-step sample + decode, B = 16.  Developer timing tool (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tqdne_amd import LightningAutoencoder, LightningEDM, paper_1d_unet_config
from tqdne_amd.trainer import DataParallelTrainer

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
base = dict(model_channels=64, channel_mult=(1, 2, 4), attention_resolutions=(), num_res_blocks=2, dims=1, conv_kernel_size=5, dropout=0.1)
torch.manual_seed(0)
ae = LightningAutoencoder(dict(base, in_channels=3, out_channels=32), dict(base, in_channels=16, out_channels=3),
                          {"learning_rate": 1e-4, "max_steps": 1000, "eta_min": 0.0})
with torch.no_grad():
    for p in ae.parameters():
        if torch.count_nonzero(p) == 0:
            p.normal_(0, 0.02)
ae = ae.to(dev).eval()
cfg = paper_1d_unet_config(in_channels=16, out_channels=16)
edm = LightningEDM(cfg, {"learning_rate": 1e-4, "max_steps": 1000, "eta_min": 0.0}, num_sampling_steps=18, autoencoder=ae)
with torch.no_grad():
    for p in edm.unet.parameters():
        if torch.count_nonzero(p) == 0:
            p.normal_(0, 0.02)
edm = edm.to(dev)
g = torch.Generator().manual_seed(1)
batch = {"signal": (0.5 * torch.randn(B, 3, 16384, generator=g)).to(dev), "cond": torch.randn(B, 5, generator=g).to(dev)}
tr = DataParallelTrainer(edm, world_size=1)


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


edm.train()
t_train = timeit(lambda: tr.train_step(batch))
edm.eval()
t_sample = timeit(lambda: edm.sample((B, 3, 16384), cond=batch["cond"]))
t_enc = timeit(lambda: ae.encode(batch["signal"]))
z = ae.encode(batch["signal"])
t_dec = timeit(lambda: ae.decode(z))
print(f"latent EDM, B={B}: train step {t_train:.1f} ms (encode + latent UNet fwd/bwd + Adam), 18-step sample + decode {t_sample:.1f} ms, "
      f"encode {t_enc:.2f} ms, decode {t_dec:.2f} ms -> {B / ((t_train + t_sample) * 1e-3):.1f} waveforms/s (3 x 16384)")
