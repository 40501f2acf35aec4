"""Flat-buffer norm-fix + recombine + clip + AdamW (HIP).  Mirrors the semantics of
delete_celeb.py:714-773 (reference) in two streaming passes; see csrc/optimizer.hip."""
import torch

from . import lib

MODE_NORM_FIX, MODE_ERASEDIFF, MODE_NORM_FIX_INF_GUARD = 0, 1, 2


class FlatAdamW:
    """torch.optim.AdamW semantics over ONE flat f32 parameter buffer, fed with the flat
    gradient pair [g_x ; g_a].  All step scalars stay on the device (graph-replay safe)."""

    def __init__(self, flat_params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 max_grad_norm=1.0, shadow=None):
        assert flat_params.dtype == torch.float32 and flat_params.is_cuda and flat_params.dim() == 1
        self.p = flat_params
        self.m = torch.zeros_like(flat_params)
        self.v = torch.zeros_like(flat_params)
        self.lr, self.betas, self.eps, self.wd = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.max_grad_norm = float(max_grad_norm)
        self.shadow = shadow                       # optional bf16 copy of p, refreshed every step
        dev = flat_params.device
        self.partials = torch.zeros(lib.query("siss_opt_partials_words"), dtype=torch.float64, device=dev)
        self.scalars = torch.zeros(lib.query("siss_opt_scalars_words"), dtype=torch.float32, device=dev)
        self.last_grad = None

    def launch(self, grads, *, scaling_norm=None, eta=None, inf_guard=False, want_grad=False):
        """Enqueue both passes (no host sync).  grads: [2, P] f32 = (g_x, g_a)."""
        n = self.p.numel()
        assert grads.dtype == torch.float32 and grads.shape == (2, n) and grads.is_contiguous()
        if eta is not None:
            mode, knob = MODE_ERASEDIFF, float(eta)
        else:
            mode, knob = (MODE_NORM_FIX_INF_GUARD if inf_guard else MODE_NORM_FIX), float(scaling_norm)
        lib.call("siss_grad_norms_scale", grads[0], grads[1], n, mode, knob, self.max_grad_norm,
                 self.betas[0], self.betas[1], self.partials, self.scalars)
        if want_grad and (self.last_grad is None):
            self.last_grad = torch.empty_like(self.p)
        lib.call("siss_recombine_clip_adamw", grads[0], grads[1], self.p, self.m, self.v, self.shadow,
                 self.last_grad if want_grad else None, n, self.lr, self.betas[0], self.betas[1],
                 self.eps, self.wd, self.scalars)

    def launch_sharded(self, gx_shard, ga_shard, lo, hi, group, *, scaling_norm=None, eta=None, inf_guard=False):
        """The same update on THIS rank's parameter shard [lo, hi) (sharded data-parallel exchange):
        gx_shard / ga_shard are the rank-summed gradients of the shard.  The three norm sums are all-reduced (24 bytes),
        so every rank forms the same scaling factor and clip coefficient as the replicated update; p / m / v / shadow are
        touched on [lo, hi) only -- the caller all-gathers the parameters afterwards."""
        import ctypes
        import torch.distributed as dist
        n = hi - lo
        assert gx_shard.numel() == n and ga_shard.numel() == n and gx_shard.dtype == torch.float32
        if eta is not None:
            mode, knob = MODE_ERASEDIFF, float(eta)
        else:
            mode, knob = (MODE_NORM_FIX_INF_GUARD if inf_guard else MODE_NORM_FIX), float(scaling_norm)
        nblk = ctypes.c_int(0)
        lib.call("siss_grad_norm_partials", gx_shard, ga_shard, n, self.partials, ctypes.byref(nblk))
        sums = self.partials.view(-1, 3)[:nblk.value].sum(dim=0, keepdim=True)      # [1, 3] f64 on the device
        dist.all_reduce(sums, group=group)
        lib.call("siss_grad_scalars", sums, 1, mode, knob, self.max_grad_norm, self.betas[0], self.betas[1], self.scalars)
        lib.call("siss_recombine_clip_adamw", gx_shard, ga_shard, self.p[lo:hi], self.m[lo:hi], self.v[lo:hi],
                 self.shadow[lo:hi] if self.shadow is not None else None, None, n, self.lr, self.betas[0],
                 self.betas[1], self.eps, self.wd, self.scalars)

    def stats(self):
        """One small D2H copy: the logged gradient scalars (delete_celeb.py:748)."""
        return self.stats_from(self.scalars.cpu())

    @staticmethod
    def stats_from(s):
        """The same from a host copy of the scalar block (SISSStepper.stats fetches it together with the loss rows)."""
        return {"norm_loss_x": float(s[0]), "norm_loss_a": float(s[1]), "dot": float(s[2]),
                "scaling_factor": float(s[3]), "pre_clip_norm": float(s[4]), "clip_coef": float(s[5]),
                "step": int(s[6])}

    def step(self, grads, **kw):
        self.launch(grads, **kw)
        return self.stats()


class AdamWSpec:
    """What ``_target_: torch.optim.AdamW`` resolves to (hydra_lite.TARGET_REMAP): the hyper-parameters of
    config/delete_celeb.yaml:127-133, consumed by the fused flat-buffer optimizer."""

    def __init__(self, params=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, **unused):
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(float(b) for b in betas), float(eps), float(weight_decay)
