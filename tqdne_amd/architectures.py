"""Config surface: the dict builders of reference tqdne/architectures.py:1-79 (1-D hot path, 2-D family), returning the
keyword dictionaries that ``UNetModel`` / ``Encoder`` / ``Decoder`` are constructed from."""


def get_1d_unet_config(config, in_channels, out_channels):
    """Paper 1-D UNet (architectures.py:22-37); ``config.features_keys`` gives the conditioning width."""
    return dict(
        in_channels=in_channels,
        out_channels=out_channels,
        cond_features=len(config.features_keys),
        dims=1,
        conv_kernel_size=5,
        model_channels=64,
        channel_mult=(1, 2, 4, 4),
        attention_resolutions=(8,),
        num_res_blocks=2,
        num_heads=4,
        dropout=0.1,
        flash_attention=False,
    )


def get_1d_autoencoder_configs(config):
    """1-D VAE encoder / decoder (architectures.py:1-19)."""
    shared = dict(model_channels=64, channel_mult=(1, 2, 4), attention_resolutions=(), num_res_blocks=2, dims=1,
                  conv_kernel_size=5, dropout=0.1)
    enc = dict(shared, in_channels=config.channels, out_channels=config.latent_channels * 2)
    dec = dict(shared, in_channels=config.latent_channels, out_channels=config.channels)
    return enc, dec


def get_2d_unet_config(config, in_channels, out_channels, model_channels=128, use_causal_mask=False):
    """2-D UNet of the generate_waveforms.py family (architectures.py:61-79); built on stock PyTorch operators (family2d.py)."""
    return dict(get_1d_unet_config(config, in_channels, out_channels), dims=2, conv_kernel_size=3,
                model_channels=model_channels, use_causal_mask=use_causal_mask)


def get_2d_autoencoder_configs(config):
    """2-D VAE encoder / decoder (architectures.py:40-58)."""
    enc, dec = get_1d_autoencoder_configs(config)
    return dict(enc, dims=2, conv_kernel_size=3), dict(dec, dims=2, conv_kernel_size=3)


def paper_1d_unet_config(in_channels=3, out_channels=3, cond_features=5):
    """BASELINE.json cfg1/cfg2/cfg4: the paper UNet on 3 x 4096 synthetic waveforms."""

    class _C:
        features_keys = tuple(range(cond_features))

    cfg = get_1d_unet_config(_C, in_channels, out_channels)
    if cond_features is None or cond_features == 0:
        cfg["cond_features"] = None
    return cfg


def tiny_1d_unet_config(in_channels=3, out_channels=3):
    """BASELINE.json cfg0: tiny UNet (32 base channels, 2 res blocks, no attention, unconditioned)."""
    return dict(in_channels=in_channels, out_channels=out_channels, cond_features=None, dims=1, conv_kernel_size=5,
                model_channels=32, channel_mult=(1, 2, 4, 4), attention_resolutions=(), num_res_blocks=2, num_heads=1,
                dropout=0.1, flash_attention=False)
