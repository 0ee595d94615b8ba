"""tqdne_amd: MI355X (gfx950) implementation of the tqdne 1-D EDM hot path behind the reference's own
module surface (UNetModel / LightningEDM / LithningConsistencyModel / architecture config dicts)."""

from .architectures import (get_1d_autoencoder_configs, get_1d_unet_config, get_2d_autoencoder_configs, get_2d_unet_config,
                            paper_1d_unet_config, tiny_1d_unet_config)
from .autoencoder import Decoder, Encoder, LightningAutoencoder
from .consistency_model import LithningConsistencyModel
from .edm import EDM, LightningEDM
from .unet import UNetModel

__version__ = "0.1.0"
__all__ = ["UNetModel", "EDM", "LightningEDM", "LithningConsistencyModel", "LightningAutoencoder", "Encoder", "Decoder", "get_1d_unet_config",
           "get_1d_autoencoder_configs", "get_2d_unet_config", "get_2d_autoencoder_configs", "paper_1d_unet_config",
           "tiny_1d_unet_config"]
