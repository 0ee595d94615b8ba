"""UNetModel: drop-in for ``tqdne.unet.UNetModel`` (reference tqdne/unet.py:146-398) whose forward runs on
hand-written gfx950 kernels through the C ABI of ``libtqdne_hip.so``.

Same constructor keywords, same ``forward(x, timesteps, cond=None)`` contract (incl. the assert of
unet.py:378-380), same ``state_dict`` keys / shapes / registration order, and the same
default initialisation (parameters are created by the same torch initialisers in the same order, so
``torch.manual_seed(s); UNetModel(**cfg)`` yields bit-identical weights to the reference).

For ``dims=1`` (the hot path) the torch ``nn.Conv1d`` / ``nn.GroupNorm`` / ``nn.Linear`` objects below are *parameter
containers only*: their ``forward`` is never called.  There is no CPU or ATen fallback for that path: tensors must live on a
ROCm device and the HIP library must be built, otherwise ``forward`` raises.

``dims=2`` builds the reference's other model family (``generate_waveforms.py``, ``architectures.py:40-79``) with ``nn.Conv2d``
containers and runs it on stock PyTorch operators (``family2d.py``) -- a separate family, not a
fallback of the 1-D path.
"""

from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import nn

from . import engine
from ._cache import plan_cache

GN_GROUPS = 32


def _gn(ch: int) -> nn.GroupNorm:  # reference nn.py:90-105 (GroupNorm32(32, ch))
    return nn.GroupNorm(GN_GROUPS, ch)


def _conv(dims: int):  # reference nn.py:16-24
    """dims=1: the hot path's parameter container; dims=2: a stock ``nn.Conv2d`` that family2d.py calls."""
    return {1: nn.Conv1d, 2: nn.Conv2d}[dims]


def _zero(m: nn.Module) -> nn.Module:  # reference nn.py:59-63
    for p in m.parameters():
        p.detach().zero_()
    return m


class FourierParams(nn.Module):
    """blocks.py:15-26: frozen W ~ N(0,1)*scale."""

    def __init__(self, channels: int, scale: float = 0.02):
        super().__init__()
        self.W = nn.Parameter(torch.randn(channels // 2) * scale, requires_grad=False)


class ResBlockParams(nn.Module):
    """Parameters of the time-conditioned ResBlock (unet.py:42-143) / plain ResBlock (blocks.py:233-260)."""

    kind = "res"

    def __init__(self, channels, emb_channels, dropout, out_channels=None, kernel_size=3, dims=1):
        super().__init__()
        conv = _conv(dims)
        out_channels = out_channels or channels
        self.channels, self.out_channels, self.kernel_size, self.dropout = channels, out_channels, kernel_size, dropout
        self.in_layers = nn.Sequential(_gn(channels), nn.SiLU(), conv(channels, out_channels, kernel_size, padding="same"))
        if emb_channels is not None:
            self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, out_channels))
        self.out_layers = nn.Sequential(
            _gn(out_channels), nn.SiLU(), nn.Dropout(p=dropout),
            _zero(conv(out_channels, out_channels, kernel_size, padding="same")),
        )
        if out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = conv(channels, out_channels, 1)


class AttentionParams(nn.Module):
    """blocks.py:111-145 (QKVAttention variant; the flash_attn variant needs a package the reference never enables)."""

    kind = "attn"

    def __init__(self, channels, num_heads=1, dims=1):
        super().__init__()
        self.channels, self.num_heads = channels, num_heads
        self.norm = _gn(channels)
        self.qkv = _conv(dims)(channels, channels * 3, 1)
        self.proj_out = _zero(_conv(dims)(channels, channels, 1))


class DownsampleParams(nn.Module):
    """blocks.py:69-108 with use_conv=True: conv k (default 3), stride 2, padding k//2."""

    kind = "down"

    def __init__(self, channels, out_channels=None, kernel_size=3, dims=1):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.op = _conv(dims)(channels, self.out_channels, kernel_size, stride=2, padding=kernel_size // 2)


class UpsampleParams(nn.Module):
    """blocks.py:29-66 with use_conv=True: nearest x2 then "same" conv."""

    kind = "up"

    def __init__(self, channels, out_channels=None, kernel_size=3, dims=1):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.conv = _conv(dims)(channels, self.out_channels, kernel_size, padding="same")


class BlockSeq(nn.Sequential):
    """Stands where the reference has TimestepEmbedSequential (unet.py:27-39): an indexable list of layers."""


class UNetModel(nn.Module):
    def __init__(
        self,
        in_channels,
        model_channels,
        out_channels,
        num_res_blocks,
        attention_resolutions=(8, 16, 32),
        dropout=0,
        channel_mult=(1, 2, 4, 8),
        conv_kernel_size=3,
        conv_resample=True,
        dims=2,
        cond_features=None,
        cond_emb_scale=None,
        use_checkpoint=False,
        num_heads=1,
        use_scale_shift_norm=False,
        flash_attention=True,
        use_causal_mask=False,
    ):
        super().__init__()
        if dims not in (1, 2):
            raise NotImplementedError("tqdne_amd.UNetModel: dims=1 (the HIP hot path) or dims=2 (stock-PyTorch family, family2d.py)")
        if use_scale_shift_norm or cond_emb_scale is not None or not conv_resample or use_causal_mask:
            raise NotImplementedError("option unused by every reference config of the supported families and not implemented")
        self.dims = dims
        if dims == 2:
            from . import family2d
            family2d.announce()
        # flash_attention is accepted and ignored: the fused kernel is always flash-style (blocks.py:193-230 needs flash_attn)
        # use_checkpoint (unet.py:129, blocks.py:137, nn.py:137-215): plans of this model keep no activation INSIDE a ResBlock /
        # AttentionBlock for the backward (shared buffers, engine._ckpt_act) and recompute them there (engine_bwd._recompute); dims=1 only
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, tuple(attention_resolutions)
        self.dropout, self.channel_mult, self.conv_kernel_size = dropout, tuple(channel_mult), conv_kernel_size
        self.num_heads, self.use_checkpoint = num_heads, use_checkpoint

        embed_dim = model_channels * 4
        self.time_embed = FourierParams(model_channels)
        self.time_mlp = nn.Sequential(nn.Linear(model_channels, embed_dim), nn.SiLU(), nn.Linear(embed_dim, embed_dim))
        self.cond_features = cond_features
        if cond_features is not None:
            self.cond_embed = None
            self.cond_mlp = nn.Sequential(nn.Linear(cond_features, embed_dim), nn.SiLU(), nn.Linear(embed_dim, embed_dim))

        k = conv_kernel_size
        ch = input_ch = int(channel_mult[0] * model_channels)
        self.input_blocks = nn.ModuleList([BlockSeq(_conv(dims)(in_channels, ch, k, padding="same"))])
        skip_chans = [ch]
        ds = 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlockParams(ch, embed_dim, dropout, int(mult * model_channels), k, dims)]
                ch = int(mult * model_channels)
                if ds in self.attention_resolutions:
                    layers.append(AttentionParams(ch, num_heads, dims))
                self.input_blocks.append(BlockSeq(*layers))
                skip_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(BlockSeq(DownsampleParams(ch, ch, dims=dims)))  # kernel 3: unet.py:273 passes none
                skip_chans.append(ch)
                ds *= 2

        self.middle_block = BlockSeq(
            ResBlockParams(ch, embed_dim, dropout, None, k, dims), AttentionParams(ch, num_heads, dims),
            ResBlockParams(ch, embed_dim, dropout, None, k, dims),
        )

        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = skip_chans.pop()
                layers = [ResBlockParams(ch + ich, embed_dim, dropout, int(model_channels * mult), k, dims)]
                ch = int(model_channels * mult)
                if ds in self.attention_resolutions:
                    layers.append(AttentionParams(ch, num_heads, dims))
                if level and i == num_res_blocks:
                    layers.append(UpsampleParams(ch, ch, k, dims))
                    ds //= 2
                self.output_blocks.append(BlockSeq(*layers))

        self.out = nn.Sequential(_gn(ch), nn.SiLU(), _zero(_conv(dims)(input_ch, out_channels, k, padding="same")))
        self._engine_cache = plan_cache()   # bounded: the least recently used (B, T, device) group of plans goes when a 7th shape arrives
        self._conv_scheme = "auto"   # "bf16x3" once the range guard of the fp16-range scheme has fired (engine.py)

    # ------------------------------------------------------------------ execution
    def _engine(self, B: int, T: int, device: torch.device, lane: int = 0) -> "engine.UNetEngine":
        """The execution plan for this shape.  ``lane`` > 0: an independent plan (own static buffers) for a second stream."""
        key = (B, T, str(device), lane)
        eng = self._engine_cache.get(key)
        if eng is None:
            # lanes >= engine.CONCURRENT_LANE0 are the sub-batch plans of a multi-lane sampler / training step: they run next to each
            # other, so grid-filling choices made for a plan that has the device to itself do not apply to them
            eng = engine.UNetEngine(self, B, T, device, solo=lane < engine.CONCURRENT_LANE0)
            self._engine_cache[key] = eng
        return eng

    def forward(self, x, timesteps, cond=None):
        """x (N, C, T) fp32 on a ROCm device, timesteps (N,), cond (N, cond_features) or None -> (N, C_out, T)."""
        assert (cond is not None) == (self.cond_features is not None), (
            "must specify cond if and only if the model is conditioned"
        )
        if self.dims == 2:   # the generate_waveforms.py family: stock PyTorch operators (family2d.py), never the 1-D path
            from . import family2d
            return family2d.unet_forward(self, x, timesteps, cond)
        engine.require_device(x)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            # an ordinary differentiable module, like the reference's (unet.py:360-398): ``loss = f(unet(x, t, c)); loss.backward()``
            # runs the HIP backward (parameter gradients; d / d x when x asks for it)
            from .autograd import unet_with_grad
            return unet_with_grad(self, x, timesteps, cond)
        eng = self._engine(x.shape[0], x.shape[2], x.device)
        y = eng.forward(x, timesteps, cond, train=False, infer=True).clone()
        if eng.check_range():   # activations near the fp16 range: the plan is on bf16x3 now, repeat
            y = eng.forward(x, timesteps, cond, train=False, infer=True).clone()
        return y

    def _apply(self, fn, *a, **k):  # parameters moved (.to / .cuda): compiled plans hold stale pointers
        self._engine_cache = plan_cache()
        self.__dict__.pop("_packed_stores", None)
        return super()._apply(fn, *a, **k)
