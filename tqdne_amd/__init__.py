"""tqdne_amd: MI355X (gfx950) implementation of the tqdne 1-D EDM hot path behind the reference's own
module surface (UNetModel / LightningEDM / LithningConsistencyModel / architecture config dicts)."""

import os as _os

# ROCm multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues per device (default 4).  The sampler integrates a batch as
# four lanes on four streams; with one more live stream in the process (RCCL's, a backward plan's) two lanes share a queue and
# serialise: measured 163.9 -> 227.2 ms for the 18-step sample at B = 64 (tools/hwq_probe.py), and no penalty with 8 queues.
# Read by the runtime when the device is first touched, so it must be in the environment before the first torch.cuda call.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .architectures import (get_1d_autoencoder_configs, get_1d_unet_config, get_2d_autoencoder_configs, get_2d_unet_config,
                            paper_1d_unet_config, tiny_1d_unet_config)
from .autoencoder import Decoder, Encoder, LightningAutoencoder
from .consistency_model import LithningConsistencyModel
from .diffusion import DDPMScheduler, LightningDDMP
from .edm import EDM, LightningEDM
from .unet import UNetModel

__version__ = "0.1.0"
__all__ = ["UNetModel", "EDM", "LightningEDM", "LightningDDMP", "DDPMScheduler", "LithningConsistencyModel", "LightningAutoencoder", "Encoder", "Decoder", "get_1d_unet_config",
           "get_1d_autoencoder_configs", "get_2d_unet_config", "get_2d_autoencoder_configs", "paper_1d_unet_config",
           "tiny_1d_unet_config"]
