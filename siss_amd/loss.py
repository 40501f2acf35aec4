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
    assert alphas_cumprod.is_cuda and gamma_tab.is_cuda and sigma_tab.is_cuda, "the schedule tables are read by the kernel: device tensors"
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


# --------------------------------------------------------------------------------------------
# Drop-in class surface of the reference (losses/ddpm_deletion_loss.py:3-122)
# --------------------------------------------------------------------------------------------
def mixture_select(noisy_keep, noisy_forget, x0, a0, t, u, gamma_tab, sigma_tab, lambd):
    assert noisy_keep.is_cuda and noisy_keep.shape == noisy_forget.shape == x0.shape == a0.shape
    dt = noisy_keep.dtype
    nk, nf, x0, a0 = (v.to(dt).contiguous() for v in (noisy_keep, noisy_forget, x0, a0))
    B, chw, dev = nk.shape[0], nk[0].numel(), nk.device
    f = lambda: torch.empty(B, dtype=torch.float32, device=dev)
    m = Mixture(torch.empty_like(nk), f(), f(), f(), f(), f(), f())
    lib.call("siss_mixture_select", nk, nf, x0, a0, int(dt == torch.bfloat16),
             t.to(device=dev, dtype=torch.int64).contiguous(), u.to(device=dev, dtype=torch.float32).contiguous(),
             gamma_tab, sigma_tab, float(lambd), B, chw, m.x_mix, m.gamma_t, m.sigma_t, m.dist_x, m.dist_a,
             m.iw_x, m.iw_a, _partials(B, chw, dev))
    return m


class _SissTerms(torch.autograd.Function):
    """(loss_x, loss_a) = ((pred-eps_x)^2, (pred-eps_a)^2) from the fused HIP kernel; autograd-connected to pred."""

    @staticmethod
    def forward(ctx, pred, m, x0, a0):
        pred32 = pred.float().contiguous()
        one = torch.ones_like(m.iw_x)
        unit = Mixture(m.x_mix, m.gamma_t, m.sigma_t, m.dist_x, m.dist_a, one, one)
        s = loss_bwd_seed(pred32, unit, x0, a0, scale=1.0, want_losses=True)
        ctx.save_for_backward(s.c_x, s.c_a)          # = 2 (pred - eps_x), 2 (pred - eps_a)
        return s.loss_x, s.loss_a

    @staticmethod
    def backward(ctx, gx, ga):
        cx, ca = ctx.saved_tensors
        return gx * cx + ga * ca, None, None, None


class _MseTerm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        c, loss, _ = mse_bwd_seed(pred.float().contiguous(), target, scale=1.0, want_loss=True)
        ctx.save_for_backward(c)
        return loss

    @staticmethod
    def backward(ctx, g):
        return g * ctx.saved_tensors[0], None


def _b(v, like):
    return v.reshape(-1, *([1] * (like.dim() - 1)))


class DDPMDeletionLoss:
    """Same constructor, method names, parameter names and 7-tuple returns as the reference class.
    ``unet`` is any callable with the diffusers contract (``siss_amd.model.UNet2DModel`` for the HIP
    network).  The elementwise math runs in the fused HIP kernels; results are autograd-connected."""

    def __init__(self, gamma, sigma):
        self.all_gamma = gamma
        self.all_sigma = sigma

    def _keep_uniforms(self, batch_size):
        return torch.rand(batch_size)        # CPU default generator, as ddpm_deletion_loss.py:18 / :101

    def importance_sampling_with_mixture(self, unet, timesteps, noise, conditioning, all_samples_dict,
                                         deletion_samples_dict, lambd):
        nk, nf = all_samples_dict["noisy_latents"], deletion_samples_dict["noisy_latents"]
        u = self._keep_uniforms(nk.shape[0])
        dev = nk.device
        m = mixture_select(nk, nf, all_samples_dict["og_latents"], deletion_samples_dict["og_latents"], timesteps,
                           u, self.all_gamma.to(dev).float().contiguous(), self.all_sigma.to(dev).float().contiguous(),
                           lambd)
        pred = unet(m.x_mix, timesteps, **conditioning, return_dict=False)[0]
        dt = m.x_mix.dtype
        loss_x, loss_a = _SissTerms.apply(pred, m, all_samples_dict["og_latents"].to(dt).contiguous(),
                                          deletion_samples_dict["og_latents"].to(dt).contiguous())
        return (None, loss_x, loss_a, m.iw_x, m.iw_a, _b(m.iw_x, loss_x) * loss_x, _b(m.iw_a, loss_a) * loss_a)

    def double_forward_with_neg_del(self, unet, timesteps, noise, conditioning, all_samples_dict,
                                    deletion_samples_dict):
        lx = _MseTerm.apply(unet(all_samples_dict["noisy_latents"], timesteps, **conditioning, return_dict=False)[0], noise)
        la = _MseTerm.apply(unet(deletion_samples_dict["noisy_latents"], timesteps, **conditioning, return_dict=False)[0], noise)
        return None, lx, la, None, None, lx, la

    def erasediff(self, unet, timesteps, noise, conditioning, all_samples_dict, deletion_samples_dict):
        lx = _MseTerm.apply(unet(all_samples_dict["noisy_latents"], timesteps, **conditioning, return_dict=False)[0], noise)
        pa = unet(deletion_samples_dict["noisy_latents"], timesteps, **conditioning, return_dict=False)[0]
        la = _MseTerm.apply(pa, torch.rand_like(pa))
        return None, lx, la, None, None, lx, la

    def simple_neg_del(self, unet, timesteps, noise, conditioning, all_samples_dict, deletion_samples_dict,
                       superfactor):
        la = _MseTerm.apply(unet(deletion_samples_dict["noisy_latents"], timesteps, **conditioning, return_dict=False)[0], noise)
        return -superfactor * la, None, la, None, None, None, None

    def naive_del(self, unet, timesteps, noise, conditioning, all_samples_dict, deletion_samples_dict):
        lx = _MseTerm.apply(unet(all_samples_dict["noisy_latents"], timesteps, **conditioning, return_dict=False)[0], noise)
        return lx, lx, None, None, None, None, None

    def subscore_bernoulli(self, unet, timesteps, noise, conditioning, all_samples_dict, deletion_samples_dict,
                           lambd):
        nk, nf = all_samples_dict["noisy_latents"], deletion_samples_dict["noisy_latents"]
        keep = (self._keep_uniforms(nk.shape[0]) > lambd).to(nk.device)
        x = torch.where(_b(keep, nk), nk, nf)
        loss = _MseTerm.apply(unet(x, timesteps, **conditioning, return_dict=False)[0], noise)
        loss_x = (1 / (1 - lambd)) * loss[keep]
        loss_a = loss[~keep]
        if len(loss_x) == 0:
            print("no nondeletion samples")
            loss_x = torch.zeros(1, 1, 1, 1, requires_grad=True)
            loss_a = torch.zeros(1, 1, 1, 1, requires_grad=True)
        if len(loss_a) == 0:
            print("no deletion samples")
            loss_a = torch.zeros(1, 1, 1, 1, requires_grad=True)
        return None, loss_x, loss_a, None, None, loss_x, loss_a
