import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import rel_err
from test_hip_unet import perturbed_state
from oracle import edm as OE
from tqdne_amd import LightningEDM, paper_1d_unet_config, tiny_1d_unet_config
which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg = dict(paper_1d_unet_config() if which == "paper" else tiny_1d_unet_config(), dropout=0.0)
torch.manual_seed(0)
edm = LightningEDM(cfg, {"learning_rate": 1e-4, "max_steps": 10, "eta_min": 0.0})
sd = perturbed_state(edm.unet, 23); edm.unet.load_state_dict(sd)
dev = torch.device("cuda:0"); edm = edm.to(dev).train()
g = torch.Generator().manual_seed(77); B = 2
sig = 0.5 * torch.randn(B, 3, T, generator=g)
cond = torch.randn(B, 5, generator=g) if cfg["cond_features"] else None
eps, noise = torch.randn(B, generator=g), torch.randn(B, 3, T, generator=g)
loss = edm.step_with_noise(sig.to(dev), eps.to(dev), noise.to(dev), cond=cond.to(dev) if cond is not None else None)
loss.backward()
params = {("unet." + k): v.clone().requires_grad_(k != "time_embed.W") for k, v in sd.items()}
lo = OE.loss_step(OE.EDMParams(), OE.make_net(params, cfg), sig, eps, noise, cond=cond); lo.backward()
errs = []
for name, p in edm.unet.named_parameters():
    if p.requires_grad:
        errs.append((rel_err(p.grad.cpu(), params["unet." + name].grad), name))
for e, n in sorted(errs, reverse=True)[:25]:
    print(f"{e:.2e} {n}")
