#!/usr/bin/env python3
"""Generate tests/golden/nano_edm.ckpt + nano_edm_expected.npz by running the *reference* in this container.

The .ckpt has the layout of a Lightning checkpoint of the reference's LightningEDM trained with the EMA callback
(tqdne/training.py:54-65, tqdne/ema.py:50-54): state_dict, hyper_parameters (with a pickled tqdne.edm.EDM), ema_state,
optimizer / scheduler states.  pytorch_lightning is not installed here, so the dict is assembled by hand from the reference
objects (the keys are Lightning's documented checkpoint keys); every object inside is the reference's own.
The .npz holds an input and the reference module's outputs with the plain and the EMA weights.

Run:  python tools/make_ckpt_golden.py      (needs /root/reference; CPU only)
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402

NANO = dict(in_channels=3, out_channels=3, model_channels=32, channel_mult=(1,), num_res_blocks=1,
            attention_resolutions=(), num_heads=1, conv_kernel_size=5, dims=1, cond_features=5, dropout=0.0,
            flash_attention=False)


def main():
    sys.path.insert(0, mg.REF)
    mg.install_lightning_standin()
    import tqdne.edm as redm

    torch.manual_seed(3)
    opt_params = dict(learning_rate=1e-4, max_steps=100, eta_min=0.0)
    edm_consts = redm.EDM()
    edm_consts.sigma_max = 60.0  # a non-default constant, so the test sees the pickled instance and not the class defaults
    m = redm.LightningEDM(NANO, opt_params, num_sampling_steps=6, deterministic_sampling=True, edm=edm_consts).eval()
    mg.perturb_(m.unet, 17)
    g = torch.Generator().manual_seed(5)
    ema = OrderedDict((n, (p.detach() + 0.01 * torch.randn(p.shape, generator=g)).clone())
                      for n, p in m.named_parameters() if p.requires_grad)
    opt = m.configure_optimizers()
    ckpt = OrderedDict()
    ckpt["epoch"] = 3
    ckpt["global_step"] = 42
    ckpt["pytorch-lightning_version"] = "2.5.1"
    ckpt["state_dict"] = m.state_dict()
    ckpt["loops"] = {}
    ckpt["callbacks"] = {}
    ckpt["optimizer_states"] = [opt["optimizer"].state_dict()]
    ckpt["lr_schedulers"] = [opt["lr_scheduler"]["scheduler"].state_dict()]
    ckpt["hparams_name"] = "kwargs"
    ckpt["hyper_parameters"] = dict(unet_config=NANO, optimizer_params=opt_params, num_sampling_steps=6,
                                    deterministic_sampling=True, edm=edm_consts)
    ckpt["ema_state"] = ema
    torch.save(ckpt, os.path.join(mg.OUT, "nano_edm.ckpt"))

    x = torch.randn(2, 3, 128, generator=g)
    sigma = torch.tensor([0.3, 7.0])
    cond = torch.randn(2, 5, generator=g)
    with torch.no_grad():
        y = m(x, sigma, cond=cond)
        m.load_state_dict(ema, strict=False)
        y_ema = m(x, sigma, cond=cond)
    np.savez_compressed(os.path.join(mg.OUT, "nano_edm_expected.npz"), x=x.numpy(), sigma=sigma.numpy(), cond=cond.numpy(),
                        y=y.numpy(), y_ema=y_ema.numpy())
    print("wrote", os.path.getsize(os.path.join(mg.OUT, "nano_edm.ckpt")), "bytes")


if __name__ == "__main__":
    main()
