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
loss = ae.step({"signal": x, "cond_signal": x * 0.5}); loss.backward(); print("AE step with cond_signal + autograd backward ok", float(loss))
net = UNetModel(**paper_1d_unet_config()); perturb(net)
cm = LithningConsistencyModel(net).to(dev).train(); cm.max_steps, cm.global_step = 1000, 10
l = cm.step({"signal": x[:, :, :4096].contiguous(), "cond": torch.randn(2, 5, generator=g).to(dev)}); l.backward()
print("iCT step on the paper UNet ok", float(l), sum(p.grad is not None for p in net.parameters()))
opt = cm.configure_optimizers(); opt.step(); print("RAdam step ok")
edm = LightningEDM(paper_1d_unet_config(in_channels=16, out_channels=16), {"learning_rate": 1e-4, "max_steps": 100, "eta_min": 0.0}, num_sampling_steps=3, autoencoder=ae.eval())
perturb(edm.unet); edm = edm.to(dev).train()
tr2 = DataParallelTrainer(edm, world_size=1, ema_decay=0.999)
print("latent EDM train step (frozen AE encode inside):", float(tr2.train_step({"signal": x, "cond": torch.randn(2, 5, generator=g).to(dev)})))
edm.eval(); print("latent sample:", edm.sample((2, 3, 16384), cond=torch.randn(2, 5, generator=g).to(dev)).shape)
edm.deterministic_sampling = False; print("stochastic latent sample:", edm.sample((2, 3, 16384), cond=torch.randn(2, 5, generator=g).to(dev)).shape)
from tqdne_amd import checkpoint
import tempfile, os
d = tempfile.mkdtemp(); f = os.path.join(d, "l.ckpt")
checkpoint.save_checkpoint(edm, f, ema_state=tr2.ema_state(), optimizer=tr2.optimizer, lr_scheduler=tr2.scheduler, global_step=1)
m2 = LightningEDM.load_from_checkpoint(f, autoencoder=ae, ema=True)
print("latent EDM checkpoint round trip ok:", type(m2.edm).__name__, len(m2.state_dict()))
# the experiments' real data shape (experiments/config.py:61-67): MovingAverageEnvelope representation, 6 channels x 4064
from tqdne_amd.representation import MovingAverageEnvelope
rep = MovingAverageEnvelope()
wave = (torch.randn(2, 3, 4064, generator=g) * torch.linspace(0.1, 2.0, 4064)).to(dev)
sig6 = rep.get_representation(wave)
edm6 = LightningEDM(paper_1d_unet_config(in_channels=6, out_channels=6), {"learning_rate": 1e-4, "max_steps": 100, "eta_min": 0.0}, num_sampling_steps=3)
perturb(edm6.unet); edm6 = edm6.to(dev).train()
tr6 = DataParallelTrainer(edm6, world_size=1)
print("6 x 4064 train step:", float(tr6.train_step({"signal": sig6, "cond": torch.randn(2, 5, generator=g).to(dev)})))
edm6.eval()
out6 = edm6.sample((2, 6, 4064), cond=torch.randn(2, 5, generator=g).to(dev))
print("6 x 4064 sample -> waveform:", rep.invert_representation(out6).shape)
