import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tqdne_amd import (LightningAutoencoder, LightningEDM, UNetModel, get_1d_autoencoder_configs, paper_1d_unet_config)
from tqdne_amd.consistency_model import LithningConsistencyModel
from tqdne_amd.trainer import DataParallelTrainer
dev = torch.device("cuda:0")
def perturb(m):
    with torch.no_grad():
        for p in m.parameters():
            if torch.count_nonzero(p) == 0: p.normal_(0, 0.02)
class C: channels, latent_channels = 3, 16
enc_cfg, dec_cfg = get_1d_autoencoder_configs(C)
torch.manual_seed(0)
ae = LightningAutoencoder(enc_cfg, dec_cfg, {"learning_rate": 1e-4, "max_steps": 100, "eta_min": 0}); perturb(ae); ae = ae.to(dev).train()
g = torch.Generator().manual_seed(1)
x = (0.5 * torch.randn(2, 3, 16384, generator=g)).to(dev)
tr = DataParallelTrainer(ae, world_size=1)
l0 = float(tr.train_step({"signal": x})); l1 = float(tr.train_step({"signal": x}))
print("AE real config train steps:", l0, l1)
import math
def fin(*v):   # (this walk is the suite's only end-to-end pass over EVERY configuration: at least nothing may come out non-finite)
    for t in v:
        assert (torch.isfinite(t).all() if torch.is_tensor(t) else math.isfinite(t)), "non-finite value in a configuration's end-to-end pass"
fin(l0, l1); assert l1 < l0 * 1.5
loss = ae.step({"signal": x, "cond_signal": x * 0.5}); loss.backward(); print("AE step with cond_signal + autograd backward ok", float(loss))
net = UNetModel(**paper_1d_unet_config()); perturb(net)
cm = LithningConsistencyModel(net).to(dev).train(); cm.max_steps, cm.global_step = 1000, 10
l = cm.step({"signal": x[:, :, :4096].contiguous(), "cond": torch.randn(2, 5, generator=g).to(dev)}); l.backward()
print("iCT step on the paper UNet ok", float(l), sum(p.grad is not None for p in net.parameters()))
fin(float(l), *[p.grad for p in net.parameters() if p.grad is not None]); assert sum(p.grad is not None for p in net.parameters()) == len(list(net.parameters())) - 1   # (all but the frozen Fourier frequencies)
opt = cm.configure_optimizers(); opt.step(); print("RAdam step ok")
edm = LightningEDM(paper_1d_unet_config(in_channels=16, out_channels=16), {"learning_rate": 1e-4, "max_steps": 100, "eta_min": 0.0}, num_sampling_steps=3, autoencoder=ae.eval())
perturb(edm.unet); edm = edm.to(dev).train()
tr2 = DataParallelTrainer(edm, world_size=1, ema_decay=0.999)
lt = float(tr2.train_step({"signal": x, "cond": torch.randn(2, 5, generator=g).to(dev)})); fin(lt)
print("latent EDM train step (frozen AE encode inside):", lt)
edm.eval(); s_det = edm.sample((2, 3, 16384), cond=torch.randn(2, 5, generator=g).to(dev)); print("latent sample:", s_det.shape)
edm.deterministic_sampling = False; s_sto = edm.sample((2, 3, 16384), cond=torch.randn(2, 5, generator=g).to(dev)); print("stochastic latent sample:", s_sto.shape)
fin(s_det, s_sto); assert tuple(s_det.shape) == (2, 3, 16384) == tuple(s_sto.shape)
from tqdne_amd import checkpoint
import tempfile, os
d = tempfile.mkdtemp(); f = os.path.join(d, "l.ckpt")
checkpoint.save_checkpoint(edm, f, ema_state=tr2.ema_state(), optimizer=tr2.optimizer, lr_scheduler=tr2.scheduler, global_step=1)
m2 = LightningEDM.load_from_checkpoint(f, autoencoder=ae, ema=True)
print("latent EDM checkpoint round trip ok:", type(m2.edm).__name__, len(m2.state_dict()))
ema, sd2, ncmp = tr2.ema_state(), m2.state_dict(), 0
for k_, v_ in ema.items():   # (ema=True: the loaded weights are the EMA's, bit for bit)
    if k_ in sd2:
        assert torch.equal(sd2[k_].cpu(), v_.cpu()), k_
        ncmp += 1
assert ncmp >= 100, (ncmp, list(ema)[:3], list(sd2)[:3])
# the experiments' real data shape (experiments/config.py:61-67): MovingAverageEnvelope representation, 6 channels x 4064
from tqdne_amd.representation import MovingAverageEnvelope
rep = MovingAverageEnvelope()
wave = (torch.randn(2, 3, 4064, generator=g) * torch.linspace(0.1, 2.0, 4064)).to(dev)
sig6 = rep.get_representation(wave)
edm6 = LightningEDM(paper_1d_unet_config(in_channels=6, out_channels=6), {"learning_rate": 1e-4, "max_steps": 100, "eta_min": 0.0}, num_sampling_steps=3)
perturb(edm6.unet); edm6 = edm6.to(dev).train()
tr6 = DataParallelTrainer(edm6, world_size=1)
l6 = float(tr6.train_step({"signal": sig6, "cond": torch.randn(2, 5, generator=g).to(dev)})); fin(l6)
print("6 x 4064 train step:", l6)
edm6.eval()
out6 = edm6.sample((2, 6, 4064), cond=torch.randn(2, 5, generator=g).to(dev))
w6 = rep.invert_representation(out6); fin(out6, w6); assert tuple(w6.shape) == (2, 3, 4064)
print("6 x 4064 sample -> waveform:", w6.shape)
