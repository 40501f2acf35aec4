"""VAE encoder (diffusers ``AutoencoderKL.encode``, SD v1.x) forward on HIP -- the frozen front end that maps both
image batches to latents before the SISS step (delete_sd.py:464-468 load, :879-888
``vae.encode(x).latent_dist.sample() * vae.config.scaling_factor``).

Forward only (the VAE is frozen, delete_sd.py:476).  It is the UNet engine's own machinery on a different graph:
padded-NHWC bf16 activations, GroupNorm+SiLU kernels, 3x3 / 1x1 / stride-2 convolutions and the single-head
attention as MFMA GEMMs.  ``quant_conv`` (1x1, 8 -> 8) is folded into ``conv_out`` when the weights are loaded
(both are linear: W' = W_q W_out, b' = W_q b_out + b_q), so the moments come out of one 3x3 GEMM.
Parameter names are the diffusers state-dict keys (``encoder.*``, ``quant_conv.*``).
"""
from dataclasses import dataclass
from typing import Tuple

import torch

from . import lib
from .unet import ParamStore, UNetEngine


@dataclass
class VAEEncoderConfig:
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215
    downsample_padding: int = 0          # Downsample2D(padding=0): F.pad (0,1,0,1) then stride 2

    @staticmethod
    def from_dict(d):
        names = set(VAEEncoderConfig.__dataclass_fields__)
        return VAEEncoderConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items() if k in names})

    def head_dim(self, channels):
        return channels                  # mid-block attention: one head as wide as the block


class VAEEncoder(UNetEngine):
    forward_only = True

    def __init__(self, cfg: VAEEncoderConfig = None, device="cuda"):
        lib.load()
        self.cfg = cfg or VAEEncoderConfig()
        self.device = torch.device(device)
        lib.ensure_workspace(self.device)
        self.ps = ParamStore()
        self._declare_params()
        self.ps.allocate(self.device, nsets=1)
        self.wT, self._wds, self._acts, self._bufs, self._pool = {}, {}, {}, {}, {}
        self.tape, self.gmap, self._uid = [], {}, 0
        self.on_early_grads_final = None
        self.adt, self.f32 = torch.bfloat16, False         # (forward-only front end: the bf16 path)
        self._wq, self._held, self._held_release = [], {}, []
        self._pair1 = []
        self._wq_post = []
        self._side, self._side_busy, self._side_held, self._side_release, self._side_mark = None, False, {}, [], None
        self._side_phase = False
        self._prep_pending, self._wT_stale = False, False
        self._up_w = {}

    # ------------------------------------------------------------------ parameters
    def _declare_enc_resnet(self, pre, cin, cout):
        a = self.ps.add
        a(f"{pre}.norm1.weight", "vec", (cin,)); a(f"{pre}.norm1.bias", "vec", (cin,))
        a(f"{pre}.conv1.weight", "conv3", (cout, cin, 3, 3)); a(f"{pre}.conv1.bias", "vec", (cout,))
        a(f"{pre}.norm2.weight", "vec", (cout,)); a(f"{pre}.norm2.bias", "vec", (cout,))
        a(f"{pre}.conv2.weight", "conv3", (cout, cout, 3, 3)); a(f"{pre}.conv2.bias", "vec", (cout,))
        if cin != cout:
            a(f"{pre}.conv_shortcut.weight", "conv1", (cout, cin, 1, 1)); a(f"{pre}.conv_shortcut.bias", "vec", (cout,))

    def _declare_params(self):
        cfg, a = self.cfg, self.ps.add
        ch = cfg.block_out_channels
        a("conv_in.weight", "conv_in", (ch[0], cfg.in_channels, 3, 3)); a("conv_in.bias", "vec", (ch[0],))
        self.plan = []
        out = ch[0]
        for i, c in enumerate(ch):
            cin, out = out, c
            down = i != len(ch) - 1
            for j in range(cfg.layers_per_block):
                self._declare_enc_resnet(f"down_blocks.{i}.resnets.{j}", cin if j == 0 else out, out)
            if down:
                a(f"down_blocks.{i}.downsamplers.0.conv.weight", "conv3", (out, out, 3, 3))
                a(f"down_blocks.{i}.downsamplers.0.conv.bias", "vec", (out,))
            self.plan.append((i, down))
        c = ch[-1]
        self._declare_enc_resnet("mid_block.resnets.0", c, c)
        self._declare_attn("mid_block.attentions.0", c)
        self._declare_enc_resnet("mid_block.resnets.1", c, c)
        a("conv_norm_out.weight", "vec", (c,)); a("conv_norm_out.bias", "vec", (c,))
        # conv_out with quant_conv folded in (see load_state_dict)
        a("conv_out.weight", "conv3", (2 * cfg.latent_channels, c, 3, 3)); a("conv_out.bias", "vec", (2 * cfg.latent_channels,))

    def load_state_dict(self, sd, strict=True):
        """diffusers AutoencoderKL state dict (decoder / post_quant_conv entries are ignored)."""
        enc = {k[len("encoder."):]: v.float() for k, v in sd.items() if k.startswith("encoder.")}
        wq = sd["quant_conv.weight"].float()[:, :, 0, 0]
        enc["conv_out.bias"] = wq @ enc["conv_out.bias"] + sd["quant_conv.bias"].float()
        enc["conv_out.weight"] = torch.einsum("om,mckl->ockl", wq, enc["conv_out.weight"])
        self.ps.load_state_dict(enc, strict)
        self.refresh_weights(cast_shadow=True)

    def refresh_weights(self, cast_shadow=False):
        ps = self.ps
        lib.call("siss_cast_f32_bf16", ps.flat, ps.shadow, ps.total)       # frozen: no dgrad copies needed

    @classmethod
    def from_pretrained(cls, path, subfolder="vae", device="cuda"):
        import json
        import os
        from safetensors.torch import load_file
        d = os.path.join(path, subfolder) if subfolder else path
        m = cls(VAEEncoderConfig.from_dict(json.load(open(os.path.join(d, "config.json")))), device)
        m.load_state_dict(load_file(os.path.join(d, "diffusion_pytorch_model.safetensors")))
        return m

    # ------------------------------------------------------------------ graph
    def _enc_resnet(self, x, pre):
        a1, _ = self.gn(x, pre + ".norm1", True)
        h, _ = self.conv(a1, pre + ".conv1")
        a2, _ = self.gn(h, pre + ".norm2", True)
        if (pre + ".conv_shortcut.weight") in self.ps.specs:
            res, _ = self.conv(x, pre + ".conv_shortcut", ksize=1)
        else:
            res = x
        out, _ = self.conv(a2, pre + ".conv2", residual=res)
        return out

    @torch.no_grad()
    def moments(self, x):
        """x [N, 3, H, W] images in [-1, 1] (f32 / bf16, device).  Returns (mean, logvar) [N, 4, H/8, W/8] f32."""
        cfg = self.cfg
        assert x.is_cuda and x.dim() == 4 and x.shape[1] == cfg.in_channels
        self.tape, self.gmap, self._uid = [], {}, 0
        self.nf = x.shape[0]
        h = self._conv_in(x.contiguous())
        for i, down in self.plan:
            for j in range(cfg.layers_per_block):
                h = self._enc_resnet(h, f"down_blocks.{i}.resnets.{j}")
            if down:
                h = self.downsample(h, f"down_blocks.{i}.downsamplers.0")
        h = self._enc_resnet(h, "mid_block.resnets.0")
        h = self.attention(h, "mid_block.attentions.0")
        h = self._enc_resnet(h, "mid_block.resnets.1")
        a, _ = self.gn(h, "conv_norm_out", True)
        m, _ = self.conv(a, "conv_out")
        self.tape = []                                   # forward only: drop the backward closures
        mom = m.to_nchw()
        mean, logvar = mom.chunk(2, dim=1)
        return mean.contiguous(), logvar.clamp(-30.0, 20.0).contiguous()

    @torch.no_grad()
    def encode(self, x, eps=None, generator=None):
        """``vae.encode(x).latent_dist.sample() * scaling_factor`` (delete_sd.py:879-888)."""
        mean, logvar = self.moments(x)
        if eps is None:
            eps = torch.randn(mean.shape, device=mean.device, generator=generator)
        return (mean + torch.exp(0.5 * logvar) * eps.to(mean.device)) * self.cfg.scaling_factor
