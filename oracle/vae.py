"""Torch-only restatement of the ENCODER half of diffusers==0.27.2 ``AutoencoderKL`` (SD v1.x VAE; oracle, test
infrastructure only).

The reference maps both image batches to latents with the frozen VAE before the SISS loss
(delete_sd.py:464-468 load, :879-888 ``vae.encode(x).latent_dist.sample() * vae.config.scaling_factor``).
**Parity unpinned by the reference**: diffusers is a pip dependency (environment.yml:232), not vendored and not
installed here; this file restates the published architecture and is pinned by the exact parameter count of the
CompVis/stable-diffusion-v1-4 ``vae`` encoder + quant_conv (34,163,592 + 72) and its state-dict key names.
"""
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet import Attention, Downsample2D


@dataclass
class VAEConfig:
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215

    @staticmethod
    def sd_v1():
        return VAEConfig()

    @staticmethod
    def tiny():
        return VAEConfig(block_out_channels=(64, 128), layers_per_block=1)


class EncResnet(nn.Module):
    """ResnetBlock2D with temb_channels=None."""

    def __init__(self, cin, cout, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class EncDownBlock(nn.Module):
    def __init__(self, cin, cout, cfg, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([EncResnet(cin if i == 0 else cout, cout, cfg.norm_num_groups, cfg.norm_eps)
                                      for i in range(cfg.layers_per_block)])
        self.add_down = add_down
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout, 0)])     # F.pad (0,1,0,1) + stride 2, pad 0

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.downsamplers[0](x) if self.add_down else x


class EncMidBlock(nn.Module):
    def __init__(self, ch, cfg):
        super().__init__()
        self.resnets = nn.ModuleList([EncResnet(ch, ch, cfg.norm_num_groups, cfg.norm_eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Attention(ch, ch, cfg.norm_num_groups, cfg.norm_eps)])   # one head of `ch`

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch = cfg.block_out_channels
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i, c in enumerate(ch):
            cin, out = out, c
            self.down_blocks.append(EncDownBlock(cin, out, cfg, i != len(ch) - 1))
        self.mid_block = EncMidBlock(ch[-1], cfg)
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[-1], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(ch[-1], 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class OracleVAEEncoder(nn.Module):
    """``vae.encode(x).latent_dist``: moments = quant_conv(encoder(x)); mean, logvar = chunk(moments)."""

    def __init__(self, cfg: VAEConfig):
        super().__init__()
        self.cfg = cfg
        self.encoder = Encoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)

    def moments(self, x):
        m = self.quant_conv(self.encoder(x))
        mean, logvar = m.chunk(2, dim=1)
        return mean, logvar.clamp(-30.0, 20.0)

    def encode(self, x, eps):
        """mean + exp(0.5 logvar) * eps, times the scaling factor (delete_sd.py:879-888); eps ~ N(0, 1) given."""
        mean, logvar = self.moments(x)
        return (mean + torch.exp(0.5 * logvar) * eps) * self.cfg.scaling_factor
