"""Torch-only restatement of diffusers==0.27.2 ``UNet2DConditionModel`` (SD v1.x; oracle, test infrastructure only).

**Parity unpinned by the reference**: diffusers is a pip dependency of the reference (environment.yml:232), not
vendored under /root/reference and not installed in this image.  This file restates the published architecture
(SURVEY.md §8 a-U "SD UNet (config 5)" and Appendix A7); it is pinned by the exact parameter count of the
runwayml/stable-diffusion-v1-5 UNet -- 859,520,964 -- and by its state-dict key names (tests/test_oracle_unet.py).

Reference call sites: losses/ddpm_deletion_loss.py:24 with ``conditioning = {'encoder_hidden_states': [B,77,768]}``
(delete_sd.py:968-985), load at delete_sd.py:458-462.  Parameter names are the diffusers state-dict keys.
"""
import math
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet import Downsample2D, ResnetBlock2D, TimestepEmbedding, Upsample2D, timestep_embedding


@dataclass
class UNetCondConfig:
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",)
    up_block_types: Tuple[str, ...] = ("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3
    layers_per_block: int = 2
    attention_head_dim: int = 8          # diffusers quirk: for this model the field is the NUMBER of heads
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    downsample_padding: int = 1
    flip_sin_to_cos: bool = True
    freq_shift: int = 0
    act_fn: str = "silu"

    @staticmethod
    def sd15():
        """runwayml/stable-diffusion-v1-5 unet/config.json (config/delete_sd.yaml pretrained_model_name_or_path)."""
        return UNetCondConfig()

    @staticmethod
    def tiny(ch=(64, 128), heads=2, cross_dim=64, sample_size=16, in_channels=4):
        """Small SD-shaped config for parity tests (one cross-attention level, one plain level)."""
        return UNetCondConfig(sample_size=sample_size, in_channels=in_channels, out_channels=in_channels,
                              block_out_channels=tuple(ch),
                              down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                              up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
                              attention_head_dim=heads, cross_attention_dim=cross_dim)


class AttnOut(nn.ModuleList):
    """``to_out`` = [Linear, Dropout(0)]; only index 0 has parameters."""


class CrossAttention(nn.Module):
    """Appendix A7: q from x, k/v from context (x itself for self-attention); q/k/v without bias, out with bias."""

    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(ctx_dim, dim, bias=False)
        self.to_v = nn.Linear(ctx_dim, dim, bias=False)
        self.to_out = AttnOut([nn.Linear(dim, dim)])

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        b, s, c = x.shape
        d = c // self.heads

        def split(t):
            return t.reshape(b, t.shape[1], self.heads, d).transpose(1, 2)
        q, k, v = split(self.to_q(x)), split(self.to_k(ctx)), split(self.to_v(ctx))
        p = torch.softmax((q @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1)
        o = (p @ v).transpose(1, 2).reshape(b, s, c)
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(g)


class FeedForward(nn.Module):
    """``net`` = [GEGLU, Dropout(0), Linear]: parameters at net.0.proj and net.2."""

    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, 4 * dim), nn.Identity(), nn.Linear(4 * dim, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = CrossAttention(dim, dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = CrossAttention(dim, cross_dim, heads)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ctx)
        return x + self.ff(self.norm3(x))


class Transformer2DModel(nn.Module):
    """GroupNorm(32, eps 1e-6) -> proj_in (1x1 conv) -> [B, HW, C] -> block -> proj_out (1x1 conv) -> + input."""

    def __init__(self, ch, heads, cross_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.proj_in = nn.Conv2d(ch, ch, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(ch, heads, cross_dim)])
        self.proj_out = nn.Conv2d(ch, ch, 1)

    def forward(self, x, ctx):
        b, c, h, w = x.shape
        y = self.proj_in(self.norm(x))
        y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
        for blk in self.transformer_blocks:
            y = blk(y, ctx)
        y = y.reshape(b, h, w, c).permute(0, 3, 1, 2)
        return self.proj_out(y) + x


class CondDownBlock(nn.Module):
    def __init__(self, cin, cout, temb, cfg, attn, add_down):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb, cfg.norm_num_groups, cfg.norm_eps)
             for i in range(cfg.layers_per_block)])
        if attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout, cfg.attention_head_dim, cfg.cross_attention_dim, cfg.norm_num_groups)
                 for _ in range(cfg.layers_per_block)])
        self.has_attn = attn
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout, cfg.downsample_padding)])
        self.add_down = add_down

    def forward(self, x, emb, ctx):
        outs = ()
        for i, r in enumerate(self.resnets):
            x = r(x, emb)
            if self.has_attn:
                x = self.attentions[i](x, ctx)
            outs += (x,)
        if self.add_down:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class CondMidBlock(nn.Module):
    def __init__(self, ch, temb, cfg):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb, cfg.norm_num_groups, cfg.norm_eps) for _ in range(2)])
        self.attentions = nn.ModuleList(
            [Transformer2DModel(ch, cfg.attention_head_dim, cfg.cross_attention_dim, cfg.norm_num_groups)])

    def forward(self, x, emb, ctx):
        x = self.resnets[0](x, emb)
        x = self.attentions[0](x, ctx)
        return self.resnets[1](x, emb)


class CondUpBlock(nn.Module):
    def __init__(self, cin, cout, cprev, temb, cfg, attn, add_up):
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
                [Transformer2DModel(cout, cfg.attention_head_dim, cfg.cross_attention_dim, cfg.norm_num_groups)
                 for _ in range(n)])
        self.has_attn = attn
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])
        self.add_up = add_up

    def forward(self, x, skips, emb, ctx):
        for i, r in enumerate(self.resnets):
            x = torch.cat([x, skips[-1]], dim=1)
            skips = skips[:-1]
            x = r(x, emb)
            if self.has_attn:
                x = self.attentions[i](x, ctx)
        if self.add_up:
            x = self.upsamplers[0](x)
        return x


class OracleUNet2DCondition(nn.Module):
    def __init__(self, cfg: UNetCondConfig):
        super().__init__()
        self.cfg = cfg
        ch = cfg.block_out_channels
        temb = ch[0] * 4
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i, kind in enumerate(cfg.down_block_types):
            cin, out = out, ch[i]
            self.down_blocks.append(CondDownBlock(cin, out, temb, cfg, kind.startswith("CrossAttn"), i != len(ch) - 1))
        self.mid_block = CondMidBlock(ch[-1], temb, cfg)
        rev = list(reversed(ch))
        self.up_blocks = nn.ModuleList()
        out = rev[0]
        for i, kind in enumerate(cfg.up_block_types):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, len(ch) - 1)]
            self.up_blocks.append(CondUpBlock(cin, out, prev, temb, cfg, kind.startswith("CrossAttn"), i != len(ch) - 1))
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)

    @property
    def config(self):
        return self.cfg

    def forward(self, sample, timestep, encoder_hidden_states, return_dict=False, **unused):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=sample.device)
        if t.dim() == 0:
            t = t[None]
        t = t.expand(sample.shape[0])
        emb = timestep_embedding(t, self.cfg.block_out_channels[0], self.cfg.flip_sin_to_cos,
                                 self.cfg.freq_shift).to(sample.dtype)
        emb = self.time_embedding(emb)
        ctx = encoder_hidden_states
        x = self.conv_in(sample)
        skips = (x,)
        for blk in self.down_blocks:
            x, outs = blk(x, emb, ctx)
            skips += outs
        x = self.mid_block(x, emb, ctx)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            x = blk(x, skips[-n:], emb, ctx)
            skips = skips[:-n]
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return (x,)
