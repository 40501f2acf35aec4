"""SISS loss layer on HIP: fused mixture / IS-weight kernel, fused loss-seed kernel, and the
drop-in ``DDPMDeletionLoss`` class surface of the reference (losses/ddpm_deletion_loss.py:3-122).
"""
from dataclasses import dataclass
from typing import Optional

import torch

from . import lib


@dataclass
class Mixture:
    x_mix: torch.Tensor          # [B,C,H,W], dtype of the inputs (bf16 in mixed-precision mode)
    gamma_t: torch.Tensor        # [B] f32
    sigma_t: torch.Tensor
    dist_x: torch.Tensor
    dist_a: torch.Tensor
    iw_x: torch.Tensor
    iw_a: torch.Tensor


@dataclass
class LossSeed:
    c_x: Optional[torch.Tensor]      # d(sum weighted_loss_x * scale)/d pred
    c_a: Optional[torch.Tensor]
    loss_x: Optional[torch.Tensor]
    loss_a: Optional[torch.Tensor]
    sum_loss_x: torch.Tensor         # [B] per-sample sums over (c,h,w)
    sum_loss_a: torch.Tensor


def _partials(B, chw, dev):
    return torch.empty(lib.query("siss_loss_partials_words", B, chw), dtype=torch.float64, device=dev)


def mixture_fwd(x0, a0, noise, t, u, alphas_cumprod, gamma_tab, sigma_tab, lambd, partials=None):
    """Fused q_sample(x0) / q_sample(a0) with the SAME noise (delete_celeb.py:602-603), defensive
    mixture row select (ddpm_deletion_loss.py:18-23) and IS weights (:33-45)."""
    assert x0.is_cuda and x0.shape == a0.shape == noise.shape and x0.dtype == a0.dtype == noise.dtype
    assert x0.dtype in (torch.float32, torch.bfloat16)
    x0, a0, noise = x0.contiguous(), a0.contiguous(), noise.contiguous()
    B = x0.shape[0]
    chw = x0[0].numel()
    dev = x0.device
    t = t.to(device=dev, dtype=torch.int64).contiguous()
    u = u.to(device=dev, dtype=torch.float32).contiguous()
    x_mix = torch.empty_like(x0)
    f = lambda: torch.empty(B, dtype=torch.float32, device=dev)
    m = Mixture(x_mix, f(), f(), f(), f(), f(), f())
    if partials is None:
        partials = _partials(B, chw, dev)
    lib.call("siss_mixture_fwd", x0, a0, noise, int(x0.dtype == torch.bfloat16), t, u, alphas_cumprod,
             gamma_tab, sigma_tab, float(lambd), B, chw, x_mix, m.gamma_t, m.sigma_t, m.dist_x, m.dist_a,
             m.iw_x, m.iw_a, partials)
    return m


def loss_bwd_seed(pred, m: Mixture, x0, a0, scale, want_losses=False, want_cotangents=True, partials=None,
                  c_out=None):
    """Cotangents c_x, c_a = d/dpred [ sum(iw * (pred - eps)^2) * scale ]  (delete_celeb.py:686-687;
    scale = 1/(train_batch_size * grad_accum)) + per-sample loss sums for the stats block (:626-663)."""
    assert pred.dtype == torch.float32 and pred.is_contiguous()
    B = pred.shape[0]
    chw = pred[0].numel()
    dev = pred.device
    e = lambda: torch.empty_like(pred)
    if c_out is not None:          # caller-provided stacked [2B,...] cotangent buffer: rows [0,B)=c_x, [B,2B)=c_a
        assert c_out.shape[0] == 2 * B and c_out.dtype == torch.float32 and c_out.is_contiguous()
        cx, ca = c_out[:B], c_out[B:]
    else:
        cx, ca = (e(), e()) if want_cotangents else (None, None)
    s = LossSeed(cx, ca,
                 e() if want_losses else None, e() if want_losses else None,
                 torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.float32, device=dev))
    if partials is None:
        partials = _partials(B, chw, dev)
    lib.call("siss_loss_bwd_seed", pred, m.x_mix, x0.contiguous(), a0.contiguous(),
             int(m.x_mix.dtype == torch.bfloat16), m.gamma_t, m.sigma_t, m.iw_x, m.iw_a, float(scale), B, chw,
             s.c_x, s.c_a, s.loss_x, s.loss_a, s.sum_loss_x, s.sum_loss_a, partials)
    return s


def mse_bwd_seed(pred, target, scale, want_loss=False, partials=None):
    """c = 2*scale*(pred - target), loss = (pred-target)^2 (ddpm_deletion_loss.py:62,65,84,93)."""
    assert pred.dtype == torch.float32 and pred.is_contiguous()
    B, chw, dev = pred.shape[0], pred[0].numel(), pred.device
    c = torch.empty_like(pred)
    loss = torch.empty_like(pred) if want_loss else None
    sums = torch.empty(B, dtype=torch.float32, device=dev)
    if partials is None:
        partials = _partials(B, chw, dev)
    lib.call("siss_mse_bwd_seed", pred, target.contiguous(), int(target.dtype == torch.bfloat16), float(scale),
             B, chw, c, loss, sums, partials)
    return c, loss, sums
