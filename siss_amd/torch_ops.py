"""``torch.ops.siss.*``: the self-contained tensor kernels of the path registered as PyTorch custom ops
(BASELINE.json north_star: "driven from Python through PyTorch-ROCm custom ops").

These are thin ``torch.library`` registrations over the same C ABI the engine drives directly
(``siss_amd/lib.py``): tensors in, tensors out, launched on torch's current stream, with shape-only fake
implementations so that they trace / compile.  The UNet itself is not an op -- it is an engine with persistent
buffers (``siss_amd/unet.py``) -- but everything around it that the reference's loop does with ~25 elementwise
launches per call is:

    siss::mixture_fwd      x0, a0, noise, t, u, alphas_cumprod, lambd -> x_mix, iw_x, iw_a, gamma_t, sigma_t
    siss::loss_bwd_seed    pred, x_mix, x0, a0, gamma_t, sigma_t, iw_x, iw_a, scale -> c_x, c_a, sum_loss_x, sum_loss_a
    siss::mse_bwd_seed     pred, target, scale -> c, sum_loss
    siss::ddpm_step        x, eps, noise, sqrt_a, sqrt_b, c_x0, c_xt, sigma, clip -> x_prev
    siss::recombine_clip_adamw_   g_x, g_a, p, m, v, scalars, partials, ... -> ()   (in place: p, m, v, scalars)

Import this module to register them (``import siss_amd.torch_ops``).
"""
from typing import Tuple

import torch

from . import lib
from .loss import _partials

Tensor = torch.Tensor


@torch.library.custom_op("siss::mixture_fwd", mutates_args=())
def mixture_fwd(x0: Tensor, a0: Tensor, noise: Tensor, t: Tensor, u: Tensor, alphas_cumprod: Tensor,
                lambd: float) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """ddpm_deletion_loss.py:12-45 + delete_celeb.py:602-603 in one launch."""
    assert x0.is_cuda and x0.shape == a0.shape == noise.shape and x0.dtype == a0.dtype == noise.dtype
    x0, a0, noise = x0.contiguous(), a0.contiguous(), noise.contiguous()
    B, chw, dev = x0.shape[0], x0[0].numel(), x0.device
    ac = alphas_cumprod.to(device=dev, dtype=torch.float32).contiguous()
    gam, sig = (ac ** 0.5).contiguous(), ((1 - ac) ** 0.5).contiguous()
    x_mix = torch.empty_like(x0)
    f = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(6)]
    lib.call("siss_mixture_fwd", x0, a0, noise, int(x0.dtype == torch.bfloat16),
             t.to(device=dev, dtype=torch.int64).contiguous(), u.to(device=dev, dtype=torch.float32).contiguous(), ac, gam,
             sig, float(lambd), B, chw, x_mix, f[0], f[1], f[2], f[3], f[4], f[5], _partials(B, chw, dev))
    return x_mix, f[4], f[5], f[0], f[1]


@mixture_fwd.register_fake
def _(x0, a0, noise, t, u, alphas_cumprod, lambd):
    v = lambda: x0.new_empty((x0.shape[0],), dtype=torch.float32)
    return torch.empty_like(x0), v(), v(), v(), v()


@torch.library.custom_op("siss::loss_bwd_seed", mutates_args=())
def loss_bwd_seed(pred: Tensor, x_mix: Tensor, x0: Tensor, a0: Tensor, gamma_t: Tensor, sigma_t: Tensor, iw_x: Tensor,
                  iw_a: Tensor, scale: float) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """c_x, c_a = d/dpred sum(iw (pred - eps)^2) * scale; per-sample loss sums (ddpm_deletion_loss.py:26-53)."""
    assert pred.is_cuda and pred.dtype == torch.float32
    pred = pred.contiguous()
    B, chw, dev = pred.shape[0], pred[0].numel(), pred.device
    cx, ca = torch.empty_like(pred), torch.empty_like(pred)
    sx, sa = torch.empty(B, device=dev), torch.empty(B, device=dev)
    lib.call("siss_loss_bwd_seed", pred, x_mix.contiguous(), x0.contiguous(), a0.contiguous(),
             int(x_mix.dtype == torch.bfloat16), gamma_t, sigma_t, iw_x, iw_a, float(scale), B, chw, cx, ca, None, None,
             sx, sa, _partials(B, chw, dev))
    return cx, ca, sx, sa


@loss_bwd_seed.register_fake
def _(pred, x_mix, x0, a0, gamma_t, sigma_t, iw_x, iw_a, scale):
    v = lambda: pred.new_empty((pred.shape[0],))
    return torch.empty_like(pred), torch.empty_like(pred), v(), v()


@torch.library.custom_op("siss::mse_bwd_seed", mutates_args=())
def mse_bwd_seed(pred: Tensor, target: Tensor, scale: float) -> Tuple[Tensor, Tensor]:
    """c = 2 scale (pred - target), per-sample sums of (pred - target)^2 (ddpm_deletion_loss.py:62,65,84,93)."""
    assert pred.is_cuda and pred.dtype == torch.float32
    pred, target = pred.contiguous(), target.contiguous()
    B, chw, dev = pred.shape[0], pred[0].numel(), pred.device
    c, s = torch.empty_like(pred), torch.empty(B, device=dev)
    lib.call("siss_mse_bwd_seed", pred, target, int(target.dtype == torch.bfloat16), float(scale), B, chw, c, None, s,
             _partials(B, chw, dev))
    return c, s


@mse_bwd_seed.register_fake
def _(pred, target, scale):
    return torch.empty_like(pred), pred.new_empty((pred.shape[0],))


@torch.library.custom_op("siss::ddpm_step", mutates_args=())
def ddpm_step(x: Tensor, eps: Tensor, noise: Tensor, sqrt_a: float, sqrt_b: float, c_x0: float, c_xt: float,
              sigma: float, clip: bool) -> Tensor:
    """x_prev = c_x0 * clamp((x - sqrt_b eps) / sqrt_a) + c_xt * x + sigma * noise (DDPMScheduler.step)."""
    out = torch.empty_like(x)
    lib.call("siss_ddpm_step", x.contiguous(), eps.contiguous(), noise.contiguous() if sigma != 0 else None, out,
             x.numel(), float(sqrt_a), float(sqrt_b), float(c_x0), float(c_xt), float(sigma), int(clip))
    return out


@ddpm_step.register_fake
def _(x, eps, noise, sqrt_a, sqrt_b, c_x0, c_xt, sigma, clip):
    return torch.empty_like(x)


@torch.library.custom_op("siss::recombine_clip_adamw_", mutates_args=("p", "m", "v", "scalars", "partials"))
def recombine_clip_adamw_(g_x: Tensor, g_a: Tensor, p: Tensor, m: Tensor, v: Tensor, scalars: Tensor, partials: Tensor,
                          scaling_norm: float, max_grad_norm: float, lr: float, beta1: float, beta2: float, eps: float,
                          weight_decay: float) -> None:
    """delete_celeb.py:714-773 on flat f32 buffers: norms, s = scaling_norm / |g_a|, g = g_x - s g_a, clip, AdamW."""
    n = p.numel()
    lib.call("siss_grad_norms_scale", g_x, g_a, n, 0, float(scaling_norm), float(max_grad_norm), float(beta1),
             float(beta2), partials, scalars)
    lib.call("siss_recombine_clip_adamw", g_x, g_a, p, m, v, None, None, n, float(lr), float(beta1), float(beta2),
             float(eps), float(weight_decay), scalars)
