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

and, as DIFFERENTIABLE ops (autograd through the HIP backward kernels; layout conversion at the op boundary):

    siss::conv2d_3x3       x [N,Ci,H,W], w [Co,Ci,3,3], bias -> y           (MFMA fprop / dgrad / wgrad)
    siss::groupnorm_silu   x [N,C,H,W], gamma, beta, groups, eps, silu -> y
    siss::attention        q [B,Sq,C], k, v [B,Sk,C], heads, scale -> softmax(q k^T scale) v

Import this module to register them (``import siss_amd.torch_ops``).
"""
import os
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


# --------------------------------------------------------------------------------------------------------------
# The two workhorse layers as differentiable ops on ordinary NCHW tensors.  The engine keeps activations in its
# padded-NHWC bf16 layout between kernels; these ops convert at their boundary (so they are for composing /
# validating against torch modules, not the fast path) and run the same fprop / dgrad / wgrad and GroupNorm kernels.
# --------------------------------------------------------------------------------------------------------------
from . import ops as _ops                     # noqa: E402
from .layout import Act as _Act              # noqa: E402


@torch.library.custom_op("siss::conv2d_3x3", mutates_args=())
def conv2d_3x3(x: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    """3x3 'same' convolution (stride 1, zero padding) on the MFMA NT GEMM; bf16 operands, f32 accumulate, f32 out."""
    n, ci, h, w = x.shape
    co = weight.shape[0]
    xa = _Act.from_nchw(x, x.device)
    out = _Act(n, h, w, co, x.device)
    _ops.conv_fprop(xa, _ops.conv_w_to_native(weight.float()).to(torch.bfloat16), out, bias=bias.float().contiguous())
    return out.to_nchw()


@conv2d_3x3.register_fake
def _(x, weight, bias):
    return x.new_empty((x.shape[0], weight.shape[0], x.shape[2], x.shape[3]), dtype=torch.float32)


@torch.library.custom_op("siss::conv2d_3x3_backward", mutates_args=())
def conv2d_3x3_backward(dy: Tensor, x: Tensor, weight: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """(dx, dW, dbias): dgrad on the NT GEMM with the transposed taps, wgrad (+ bias gradient) on the TN GEMM."""
    n, ci, h, w = x.shape
    co = weight.shape[0]
    dev = x.device
    dya, xa = _Act.from_nchw(dy, dev), _Act.from_nchw(x, dev)
    wn = _ops.conv_w_to_native(weight.float())
    dx = _Act(n, h, w, ci, dev)
    _ops.conv_dgrad(dya, _ops.dgrad_weight(wn), dx)
    dW = torch.zeros(1, 9, co, ci, device=dev)
    db = torch.zeros(co, device=dev)
    _ops.conv_wgrad(dya, xa, dW, nsets=1, dbias=db)
    return dx.to_nchw(), _ops.conv_w_from_native(dW[0]), db


@conv2d_3x3_backward.register_fake
def _(dy, x, weight):
    return (x.new_empty(x.shape, dtype=torch.float32), weight.new_empty(weight.shape, dtype=torch.float32),
            weight.new_empty((weight.shape[0],), dtype=torch.float32))


def _conv_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])


def _conv_bwd(ctx, g):
    x, w = ctx.saved_tensors
    dx, dW, db = torch.ops.siss.conv2d_3x3_backward(g.contiguous(), x, w)
    return dx, dW, db


conv2d_3x3.register_autograd(_conv_bwd, setup_context=_conv_setup)


@torch.library.custom_op("siss::groupnorm_silu", mutates_args=())
def groupnorm_silu(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, silu: bool) -> Tensor:
    """y = act(GroupNorm(x)) (act = SiLU or identity) with the fused HBM-bound kernels; f32 statistics."""
    n, c, h, w = x.shape
    dev = x.device
    xa, ya = _Act.from_nchw(x, dev), _Act(n, h, w, c, dev)
    mean, rstd = torch.empty(n, groups, device=dev), torch.empty(n, groups, device=dev)
    part = torch.zeros(lib.query("siss_gn_partial_words", n, h, w, c, groups), device=dev)
    lib.call("siss_groupnorm_fwd", xa.data, gamma.float().contiguous(), beta.float().contiguous(), ya.data, mean, rstd,
             part, n, h, w, c, groups, float(eps), int(silu), 0)
    return ya.to_nchw()


@groupnorm_silu.register_fake
def _(x, gamma, beta, groups, eps, silu):
    return x.new_empty(x.shape, dtype=torch.float32)


@torch.library.custom_op("siss::groupnorm_silu_backward", mutates_args=())
def groupnorm_silu_backward(dy: Tensor, x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float,
                            silu: bool) -> Tuple[Tensor, Tensor, Tensor]:
    n, c, h, w = x.shape
    dev = x.device
    xa, ya = _Act.from_nchw(x, dev), _Act(n, h, w, c, dev)
    mean, rstd = torch.empty(n, groups, device=dev), torch.empty(n, groups, device=dev)
    part = torch.zeros(lib.query("siss_gn_partial_words", n, h, w, c, groups), device=dev)
    g32, b32 = gamma.float().contiguous(), beta.float().contiguous()
    lib.call("siss_groupnorm_fwd", xa.data, g32, b32, ya.data, mean, rstd, part, n, h, w, c, groups, float(eps),
             int(silu), 0)                                        # statistics of the saved input
    dya, dxa = _Act.from_nchw(dy, dev), _Act(n, h, w, c, dev)
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    lib.call("siss_groupnorm_bwd", dya.data, xa.data, g32, b32, mean, rstd, dxa.data, None, None, None, 0, 0, dg, db,
             None, 0, part, n, n, n, c, h, w, c, groups, int(silu), 0)
    return dxa.to_nchw(), dg, db


@groupnorm_silu_backward.register_fake
def _(dy, x, gamma, beta, groups, eps, silu):
    return (x.new_empty(x.shape, dtype=torch.float32), gamma.new_empty(gamma.shape, dtype=torch.float32),
            gamma.new_empty(gamma.shape, dtype=torch.float32))


def _gn_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], inputs[2])
    ctx.groups, ctx.eps, ctx.silu = inputs[3], inputs[4], inputs[5]


def _gn_bwd(ctx, g):
    x, gamma, beta = ctx.saved_tensors
    dx, dg, db = torch.ops.siss.groupnorm_silu_backward(g.contiguous(), x, gamma, beta, ctx.groups, ctx.eps, ctx.silu)
    return dx, dg, db, None, None, None


groupnorm_silu.register_autograd(_gn_bwd, setup_context=_gn_setup)


# --------------------------------------------------------------------------------------------------------------
# Multi-head attention  softmax(q k^T * scale) v  on [B, S, heads * D] tokens: batched MFMA GEMMs over head-split,
# zero-padded operands + the whole-row softmax kernel; the backward uses the fused dS epilogue
# (dS = scale * P o (dO V^T - rowsum(dO o O))), so neither dP nor a separate softmax-backward pass exists.
# --------------------------------------------------------------------------------------------------------------
def _up(n, m):
    return -(-n // m) * m


_SLACK = int(os.environ.get("SISS_OPS_SLACK", "1024"))


def _alloc(shape, dtype, dev):
    """GEMM operand / result buffer with the engine's slack (unet.py::_buf).  Defensive only: the wgrad kernel used to
    READ (never use) the 128-column tile's remainder past the last row of a 64-column operand; its staging now takes
    the zero page for those lanes, and tests/test_hip_torch_ops.py passes with SISS_OPS_SLACK=0."""
    n = 1
    for s in shape:
        n *= s
    return torch.empty(n + _SLACK, dtype=dtype, device=dev)[:n].view(shape)


def _split(t, B, S, H, D, Sp, Dp):
    out = _alloc((B * H, Sp, Dp), torch.bfloat16, t.device)
    lib.call("siss_head_split", t, out, B, S, H, D, Sp, Dp)
    return out


def _attn_fwd(q, k, v, heads, scale):
    B, Sq, C = q.shape
    Sk = k.shape[1]
    D = C // heads
    assert D * heads == C and D % 8 == 0 and k.shape == v.shape and k.shape[0] == B and k.shape[2] == C
    Dp, Sqp, Skp, BH = _up(D, 64), _up(Sq, 64), _up(Sk, 64), B * heads
    dev = q.device
    qb, kb, vb = (t.to(torch.bfloat16).contiguous() for t in (q, k, v))
    qh, kh, vh = _split(qb, B, Sq, heads, D, Sqp, Dp), _split(kb, B, Sk, heads, D, Skp, Dp), _split(vb, B, Sk, heads, D, Skp, Dp)
    vT = _alloc((BH, Dp, Skp), torch.bfloat16, dev)
    lib.call("siss_transpose_bf16", vh, vT, BH, Skp, Dp)
    sc = _alloc((BH, Sqp, Skp), torch.bfloat16, dev)
    p = _alloc((BH, Sqp, Skp), torch.bfloat16, dev)
    _ops.gemm_nt(lib.ptr(qh), Dp, kh, lib.ptr(sc), Skp, Sqp, Skp, Dp, [0], [0], alpha=scale, batch=BH,
                 stride_a=Sqp * Dp, stride_w=Skp * Dp, stride_c=Sqp * Skp)
    lib.call("siss_softmax_rows_fwd", sc, p, BH * Sqp, Sk, Skp, 0)
    oh = _alloc((BH, Sqp, Dp), torch.bfloat16, dev)
    _ops.gemm_nt(lib.ptr(p), Skp, vT, lib.ptr(oh), Dp, Sqp, Dp, Skp, [0], [0], batch=BH,
                 stride_a=Sqp * Skp, stride_w=Dp * Skp, stride_c=Sqp * Dp)
    o = torch.empty(B, Sq, C, dtype=torch.bfloat16, device=dev)
    lib.call("siss_head_merge", oh, o, B, Sq, heads, D, Sqp, Dp)
    return o, (qh, kh, vh, p, oh)


@torch.library.custom_op("siss::attention", mutates_args=())
def attention(q: Tensor, k: Tensor, v: Tensor, heads: int, scale: float) -> Tensor:
    return _attn_fwd(q, k, v, heads, scale)[0].float()


@attention.register_fake
def _(q, k, v, heads, scale):
    return q.new_empty(q.shape, dtype=torch.float32)


@torch.library.custom_op("siss::attention_backward", mutates_args=())
def attention_backward(do: Tensor, q: Tensor, k: Tensor, v: Tensor, heads: int, scale: float) -> Tuple[Tensor, Tensor, Tensor]:
    B, Sq, C = q.shape
    Sk = k.shape[1]
    D = C // heads
    Dp, Sqp, Skp, BH = _up(D, 64), _up(Sq, 64), _up(Sk, 64), B * heads
    dev = q.device
    _, (qh, kh, vh, p, oh) = _attn_fwd(q, k, v, heads, scale)               # recompute the forward's saved tensors
    doh = _split(do.to(torch.bfloat16).contiguous(), B, Sq, heads, D, Sqp, Dp)
    delta = _alloc((BH * Sqp,), torch.float32, dev)
    lib.call("siss_rowdot", doh, oh, delta, BH * Sqp, BH * Sqp, Dp)
    ds = _alloc((BH, Sqp, Skp), torch.bfloat16, dev)
    lib.call("siss_gemm_nt_mulsub", doh, Dp, vh, ds, Skp, p, Skp, delta, Sqp, Skp, Dp, float(scale), BH, Sqp * Dp,
             Skp * Dp, Sqp * Skp)
    zp, i0 = _ops.zero_page(dev), lib.int_array([0])
    dvf = _alloc((BH, Skp, Dp), torch.float32, dev)
    dkf = _alloc((BH, Skp, Dp), torch.float32, dev)
    lib.call("siss_gemm_tn", p, Skp, doh, Dp, dvf, Skp * Dp, Skp, Dp, 1, i0, i0, BH, Sqp, Sqp, 0, Sqp, -1, zp, None, None)
    khT = _alloc((BH, Dp, Skp), torch.bfloat16, dev)
    lib.call("siss_transpose_bf16", kh, khT, BH, Skp, Dp)
    dqh = _alloc((BH, Sqp, Dp), torch.bfloat16, dev)
    _ops.gemm_nt(lib.ptr(ds), Skp, khT, lib.ptr(dqh), Dp, Sqp, Dp, Skp, [0], [0], batch=BH,
                 stride_a=Sqp * Skp, stride_w=Dp * Skp, stride_c=Sqp * Dp)
    lib.call("siss_gemm_tn", ds, Skp, qh, Dp, dkf, Skp * Dp, Skp, Dp, 1, i0, i0, BH, Sqp, Sqp, 0, Sqp, -1, zp, None, None)

    def merge(th, S, Sp):
        out = torch.empty(B, S, C, dtype=torch.bfloat16, device=dev)
        lib.call("siss_head_merge", th.to(torch.bfloat16), out, B, S, heads, D, Sp, Dp)
        return out.float()
    return merge(dqh, Sq, Sqp), merge(dkf, Sk, Skp), merge(dvf, Sk, Skp)


@attention_backward.register_fake
def _(do, q, k, v, heads, scale):
    f = lambda t: t.new_empty(t.shape, dtype=torch.float32)
    return f(q), f(k), f(v)


def _attn_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], inputs[2])
    ctx.heads, ctx.scale = inputs[3], inputs[4]


def _attn_bwd(ctx, g):
    q, k, v = ctx.saved_tensors
    dq, dk, dv = torch.ops.siss.attention_backward(g.contiguous(), q, k, v, ctx.heads, ctx.scale)
    return dq, dk, dv, None, None


attention.register_autograd(_attn_bwd, setup_context=_attn_setup)
