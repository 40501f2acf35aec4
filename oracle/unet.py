"""Torch-only restatement of diffusers==0.27.2 ``UNet2DModel`` (oracle; test infrastructure only).

**Parity unpinned by the reference**: diffusers is a pip dependency of the
reference (environment.yml:232), not vendored under /root/reference and not
installed in this image.  This file restates the published architecture
(SURVEY.md §8 a-U and Appendix A2-A6); it is pinned by the exact parameter
count / tensor count / state-dict key names of google/ddpm-celebahq-256
(113,673,219 params, 450 tensors) -- see tests/test_oracle_unet.py.

Reference call sites: losses/ddpm_deletion_loss.py:24,61,64 (``unet(x, t,
return_dict=False)[0]``), delete_celeb.py:181-186 (load).
Parameter names are the diffusers state-dict keys, so a real checkpoint's
safetensors load with ``load_state_dict`` unchanged.
"""
import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class UNetConfig:
    sample_size: int = 256
    in_channels: int = 3
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 128, 256, 256, 512, 512)
    down_block_types: Tuple[str, ...] = ("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D")
    up_block_types: Tuple[str, ...] = ("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4
    layers_per_block: int = 2
    attention_head_dim: Optional[int] = None
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    downsample_padding: int = 0
    flip_sin_to_cos: bool = False
    freq_shift: int = 1
    act_fn: str = "silu"

    @staticmethod
    def celebahq256():
        """google/ddpm-celebahq-256 unet/config.json (config/delete_celeb.yaml:7)."""
        return UNetConfig()

    @staticmethod
    def mnist_tshirt():
        """config/train_tshirt_mnist.yaml:25-41 + UNet2DModel defaults."""
        return UNetConfig(sample_size=28, in_channels=1, out_channels=1,
                          block_out_channels=(64, 128, 256),
                          down_block_types=("DownBlock2D", "AttnDownBlock2D", "DownBlock2D"),
                          up_block_types=("UpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
                          attention_head_dim=8, norm_eps=1e-5, downsample_padding=1,
                          flip_sin_to_cos=True, freq_shift=0)

    @staticmethod
    def tiny(ch=(32, 64), attn=True, sample_size=16, in_channels=3, head_dim=None):
        """Small celeb-shaped config for parity tests (same block kinds, 2 levels)."""
        return UNetConfig(sample_size=sample_size, in_channels=in_channels, out_channels=in_channels,
                          block_out_channels=tuple(ch),
                          down_block_types=("DownBlock2D", "AttnDownBlock2D" if attn else "DownBlock2D"),
                          up_block_types=("AttnUpBlock2D" if attn else "UpBlock2D", "UpBlock2D"),
                          attention_head_dim=head_dim)


def timestep_embedding(t, dim, flip_sin_to_cos, freq_shift, max_period=10000):
    """Appendix A2: sinusoidal embedding; [sin | cos] unless flipped."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device)
    exponent = exponent / (half - freq_shift)
    arg = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, cemb):
        super().__init__()
        self.linear_1 = nn.Linear(cin, cemb)
        self.linear_2 = nn.Linear(cemb, cemb)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    """Appendix A4."""

    def __init__(self, cin, cout, temb, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, emb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(emb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class AttnOut(nn.ModuleList):
    """``to_out`` = [Linear, Dropout(0)] in diffusers; only index 0 has parameters."""


class Attention(nn.Module):
    """Appendix A6: GroupNorm -> q,k,v -> softmax(QK^T/sqrt(d))V -> out -> + residual."""

    def __init__(self, ch, head_dim, groups, eps):
        super().__init__()
        self.heads = ch // head_dim
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = AttnOut([nn.Linear(ch, ch)])

    def forward(self, x):
        b, c, hh, ww = x.shape
        h = self.group_norm(x).reshape(b, c, hh * ww).transpose(1, 2)        # [B, S, C]
        d = c // self.heads

        def split(t):
            return t.reshape(b, hh * ww, self.heads, d).transpose(1, 2)      # [B, heads, S, d]
        q, k, v = split(self.to_q(h)), split(self.to_k(h)), split(self.to_v(h))
        p = torch.softmax((q @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1)
        o = (p @ v).transpose(1, 2).reshape(b, hh * ww, c)
        o = self.to_out[0](o).transpose(1, 2).reshape(b, c, hh, ww)
        return o + x


class Downsample2D(nn.Module):
    """Appendix A5."""

    def __init__(self, ch, padding):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:
            x = F.pad(x, (0, 1, 0, 1))
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, cfg, attn, add_down, head_dim):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb, cfg.norm_num_groups, cfg.norm_eps)
             for i in range(cfg.layers_per_block)])
        if attn:
            self.attentions = nn.ModuleList(
                [Attention(cout, head_dim, cfg.norm_num_groups, cfg.norm_eps)
                 for _ in range(cfg.layers_per_block)])
        self.has_attn = attn
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout, cfg.downsample_padding)])
        self.add_down = add_down

    def forward(self, x, emb):
        outs = ()
        for i, r in enumerate(self.resnets):
            x = r(x, emb)
            if self.has_attn:
                x = self.attentions[i](x)
            outs += (x,)
        if self.add_down:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, ch, temb, cfg, head_dim):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb, cfg.norm_num_groups, cfg.norm_eps)
                                      for _ in range(2)])
        self.attentions = nn.ModuleList([Attention(ch, head_dim, cfg.norm_num_groups, cfg.norm_eps)])

    def forward(self, x, emb):
        x = self.resnets[0](x, emb)
        x = self.attentions[0](x)
        return self.resnets[1](x, emb)


class UpBlock(nn.Module):
    def __init__(self, cin, cout, cprev, temb, cfg, attn, add_up, head_dim):
        super().__init__()
        n = cfg.layers_per_block + 1
        rs = []
        for i in range(n):
            skip = cin if i == n - 1 else cout
            rin = cprev if i == 0 else cout
            rs.append(ResnetBlock2D(rin + skip, cout, temb, cfg.norm_num_groups, cfg.norm_eps))
        self.resnets = nn.ModuleList(rs)
        if attn:
            self.attentions = nn.ModuleList(
                [Attention(cout, head_dim, cfg.norm_num_groups, cfg.norm_eps) for _ in range(n)])
        self.has_attn = attn
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])
        self.add_up = add_up

    def forward(self, x, skips, emb):
        for i, r in enumerate(self.resnets):
            x = torch.cat([x, skips[-1]], dim=1)          # hidden FIRST (Appendix A3)
            skips = skips[:-1]
            x = r(x, emb)
            if self.has_attn:
                x = self.attentions[i](x)
        if self.add_up:
            x = self.upsamplers[0](x)
        return x


class OracleUNet2D(nn.Module):
    """Appendix A3."""

    def __init__(self, cfg: UNetConfig):
        super().__init__()
        self.cfg = cfg
        ch = cfg.block_out_channels
        temb = ch[0] * 4
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb)

        def hd(c):
            return cfg.attention_head_dim if cfg.attention_head_dim is not None else c

        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i, kind in enumerate(cfg.down_block_types):
            cin, out = out, ch[i]
            self.down_blocks.append(DownBlock(cin, out, temb, cfg, kind.startswith("Attn"),
                                              i != len(ch) - 1, hd(out)))
        self.mid_block = MidBlock(ch[-1], temb, cfg, hd(ch[-1]))
        rev = list(reversed(ch))
        self.up_blocks = nn.ModuleList()
        out = rev[0]
        for i, kind in enumerate(cfg.up_block_types):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, len(ch) - 1)]
            self.up_blocks.append(UpBlock(cin, out, prev, temb, cfg, kind.startswith("Attn"),
                                          i != len(ch) - 1, hd(out)))
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)

    @property
    def config(self):
        return self.cfg

    def forward(self, sample, timestep, return_dict=False, **unused):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=sample.device)
        if t.dim() == 0:
            t = t[None]
        t = t.expand(sample.shape[0])
        emb = timestep_embedding(t, self.cfg.block_out_channels[0], self.cfg.flip_sin_to_cos,
                                 self.cfg.freq_shift).to(sample.dtype)
        emb = self.time_embedding(emb)
        x = self.conv_in(sample)
        skips = (x,)
        for blk in self.down_blocks:
            x, outs = blk(x, emb)
            skips += outs
        x = self.mid_block(x, emb)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            x = blk(x, skips[-n:], emb)
            skips = skips[:-n]
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return (x,)
