"""One optimizer step of the unlearning loop, on CPU (oracle; test infrastructure only).

Restates the loop body of /root/reference/delete_celeb.py:557-773 (the tshirt
and SD loops differ only in input prep): per micro-batch input prep (:560-615),
loss call (:622), the two-backward gradient split (:686-711), and on the sync
micro-step the norm-fix + recombine (:714-753), global clip (:767) and AdamW
(:769-773).  accelerate semantics: ``backward`` divides by
gradient_accumulation_steps; optimizer/zero_grad only act on the sync micro-step;
``clip_grad_norm_`` is torch.nn.utils.clip_grad_norm_.

``loss_obj`` is any object with the DDPMDeletionLoss surface -- the oracle's own
restatement, or (in oracle/make_golden.py, build container only) the reference's
class imported from /root/reference, which is how the fixtures pin this file.
"""
import math
from dataclasses import dataclass

import torch

from . import schedule as S


@dataclass
class StepStats:
    norm_loss_x: float
    norm_loss_a: float
    scaling_factor: float
    pre_clip_norm: float
    weighted_loss_x: float
    weighted_loss_a: float
    batch_stats: list = None      # one dict per micro-batch: the block the reference hands to wandb.log (:626-663)


def batch_stats(items, loss_params=None, superfactor_decay=None):
    """The per-micro-step statistics block, delete_celeb.py:626-663 (identical in delete_tshirt.py:568-605), line by
    line: mean over ALL elements, max / min / (unbiased) std over the per-sample means; importance weights over the
    batch; `superfactor` logged and then decayed IN the config (:658-662) -- loss_params is mutated like cfg is."""
    loss, loss_x, loss_a, iw_x, iw_a, _, _ = items
    out = {}
    for name, v in (("loss", loss), ("loss_x", loss_x), ("loss_a", loss_a)):                 # :627-645
        if v is not None:
            v = v.detach()
            per = v.mean(dim=[1, 2, 3])
            out[name + "/mean"] = v.mean().item()
            out[name + "/max"] = per.max().item()
            out[name + "/min"] = per.min().item()
            out[name + "/std"] = per.std().item()
    for name, v in (("importance_weight_x", iw_x), ("importance_weight_a", iw_a)):          # :647-656
        if v is not None:
            v = v.detach()
            out[name + "/mean"] = v.mean().item()
            out[name + "/max"] = v.max().item()
            out[name + "/min"] = v.min().item()
            out[name + "/std"] = v.std().item()
    if loss_params is not None and "superfactor" in loss_params:                              # :658-662
        out["superfactor"] = loss_params["superfactor"]
        if superfactor_decay is not None:
            loss_params["superfactor"] *= superfactor_decay
    return out


def prep_inputs(ac, x0, a0, noise, t):
    """delete_celeb.py:602-615: the SAME noise noises both batches."""
    keep = {"og_latents": x0, "noisy_latents": S.add_noise(ac, x0, noise, t)}
    forget = {"og_latents": a0, "noisy_latents": S.add_noise(ac, a0, noise, t)}
    return keep, forget


def unlearning_step(unet, optimizer, loss_obj, loss_fn, ac, micro_batches, *,
                    train_batch_size, scaling_norm, loss_params=None, max_grad_norm=1.0,
                    eta=None, inf_guard=False, conditioning=None, pass_u=True, superfactor_decay=None):
    """Run ONE optimizer step over ``micro_batches`` (len = gradient accumulation).

    Each micro-batch is a dict with x0, a0, noise, t and optionally u (mask uniforms).
    Returns (StepStats, g_x, g_a, g) with grads as {name: tensor}.
    """
    loss_params = dict(loss_params or {})
    ga = len(micro_batches)
    blocks = []
    fn = getattr(loss_obj, loss_fn)
    names = [n for n, _ in unet.named_parameters()]
    params = [p for _, p in unet.named_parameters()]
    accum_a = None
    wlx_tot = wla_tot = 0.0
    optimizer.zero_grad()
    for mb in micro_batches:
        keep, forget = prep_inputs(ac, mb["x0"], mb["a0"], mb["noise"], mb["t"])
        kw = dict(loss_params)
        if pass_u and mb.get("u") is not None and loss_fn in (
                "importance_sampling_with_mixture", "subscore_bernoulli"):
            kw["u"] = mb["u"]
        items = fn(unet, mb["t"], mb["noise"], conditioning or {}, keep, forget, **kw)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")               # std of a single per-sample mean is nan, as in the reference
            blocks.append(batch_stats(items, loss_params, superfactor_decay))                  # :626-663
        loss, _, _, _, _, wlx, wla = items
        if loss is not None:                                   # :682-684
            (loss.sum() / train_batch_size / ga).backward()
            continue
        wlx = wlx.sum() / train_batch_size                     # :686-687
        wla = wla.sum() / train_batch_size
        wlx_tot += float(wlx.detach()) / ga
        wla_tot += float(wla.detach()) / ga
        retain = loss_fn in ("importance_sampling_with_mixture", "subscore_bernoulli")
        (wlx / ga).backward(retain_graph=retain)               # :691
        snap = [p.grad.clone() for p in params]                # :694-696
        (wla / ga).backward()                                  # :702
        delta = [p.grad.clone() - s for p, s in zip(params, snap)]   # :705-711
        accum_a = delta if accum_a is None else [a + d for a, d in zip(accum_a, delta)]

    stats = None
    gx = ga_ = None
    if accum_a is not None:
        gx = [p.grad.clone() - a for p, a in zip(params, accum_a)]              # :717-718
        nx = math.sqrt(sum(float(torch.norm(g, p=2) ** 2) for g in gx))         # :725-734
        na = math.sqrt(sum(float(torch.norm(g, p=2) ** 2) for g in accum_a))
        if loss_fn == "erasediff":                                              # :740-742
            dot = sum(float((x * a).sum()) for x, a in zip(gx, accum_a))
            s = -max(eta - dot / (na ** 2), 0)
        else:                                                                   # :746
            s = scaling_norm / na if na > 0 else float("inf")
            if inf_guard and math.isinf(s):                                     # delete_tshirt.py:688-690
                s = 0.0
        for p, x, a in zip(params, gx, accum_a):                                # :749-750
            p.grad = x - s * a
        ga_ = accum_a
    pre = float(torch.nn.utils.clip_grad_norm_(params, max_grad_norm))          # :767
    g = {n: p.grad.clone() for n, p in zip(names, params)}
    optimizer.step()                                                            # :769
    optimizer.zero_grad()
    if gx is not None:
        stats = StepStats(nx, na, s, pre, wlx_tot, wla_tot, blocks)
        gx = dict(zip(names, gx))
        ga_ = dict(zip(names, ga_))
    else:
        stats = StepStats(float("nan"), float("nan"), float("nan"), pre, wlx_tot, wla_tot, blocks)
    return stats, gx, ga_, g
