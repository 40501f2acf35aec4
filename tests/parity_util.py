"""Shared parity metrics for the GPU tests (SURVEY.md §8c tolerances, stated once).

  * step scalars (||g_x||, ||g_a||, s, pre-clip ||g||): rel 5e-2, bf16 compute vs the fp32 oracle;
  * parameter update: cosine >= 0.99 between the two updates over the elements with a significant gradient,
    |g| > 1e-6 * ||g||_inf.  AdamW's first steps are sign-like (dtheta ~ -lr g / (|g| + eps)), so an element whose
    gradient is numerically zero moves by +-lr at random on either side; those -- and only those -- are masked out.
"""
import torch

SCALAR_RTOL = 5e-2
UPDATE_COS = 0.99
MASK_REL = 1e-6


def check_scalars(ref, got, keys=("norm_loss_x", "norm_loss_a", "scaling_factor", "pre_clip_norm"), tol=SCALAR_RTOL):
    for k in keys:
        r = getattr(ref, k) if not isinstance(ref, dict) else ref[k]
        v = got[k] if isinstance(got, dict) else getattr(got, k)
        assert abs(v - r) <= tol * abs(r), (k, v, r)


def masked_update_cosine(before, after_ref, after_got, grads_ref):
    """before / after_*: {name: tensor} in reference layout; grads_ref: the oracle's final (recombined, clipped)
    gradient of that step.  Returns (cosine over the masked elements, fraction of elements kept)."""
    ginf = max(float(g.abs().max()) for g in grads_ref.values())
    num = nr = ng = 0.0
    kept = total = 0
    for n, g in grads_ref.items():
        m = g.abs().flatten() > MASK_REL * ginf
        dr = (after_ref[n].detach().double() - before[n].double()).flatten()[m]
        dg = (after_got[n].detach().double().cpu() - before[n].double()).flatten()[m]
        num += float((dr * dg).sum()); nr += float(dr.square().sum()); ng += float(dg.square().sum())
        kept += int(m.sum()); total += m.numel()
    return num / ((nr * ng) ** 0.5 + 1e-300), kept / max(total, 1)


def assert_update_direction(before, after_ref, after_got, grads_ref, what="", min_cos=UPDATE_COS):
    cos, frac = masked_update_cosine(before, after_ref, after_got, grads_ref)
    assert frac > 0.5, (what, "mask kept only", frac)
    assert cos >= min_cos, (what, "masked update cosine", cos, "kept", frac)
    return cos


def final_gradient_cosine(eng, grads_ref, scaling_factor):
    """Cosine between the oracle's final gradient of a step (g_x - s g_a, clipped: a scale) and the HIP step's, rebuilt from the two
    gradient sets it leaves in the flat buffer -- over ALL elements, no mask.  Unlike the update direction (sign-like under AdamW's
    first steps: every element counts the same, so the 0.5 % of elements whose gradient is smaller than the bf16 noise weigh as much
    as the rest) this weighs elements by their gradient."""
    gx, ga = eng.ps.grads_ref(0), eng.ps.grads_ref(1)
    num = nr = ng = 0.0
    for n, r in grads_ref.items():
        g = (gx[n].double() - scaling_factor * ga[n].double()).flatten().cpu()
        r = r.double().flatten().cpu()
        num += float((g * r).sum()); nr += float(r.square().sum()); ng += float(g.square().sum())
    return num / ((nr * ng) ** 0.5 + 1e-300)


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


def assert_grads_match(eng, names, grads_by_set, dev, zero_ok=("to_k.bias",), zero_tol=1e-6):
    """Per-tensor gradient cosine >= 0.99 and set norms within 5e-2 for every set; returns the worst cosine."""
    bad, worst = [], (1.0, None)
    for s, grads in enumerate(grads_by_set):
        got = {n: v.to(dev) for n, v in eng.ps.grads_ref(s).items()}
        tot_r = torch.sqrt(sum(v.double().square().sum() for v in grads))
        tot_g = torch.sqrt(sum(v.double().square().sum() for v in got.values()))
        assert abs(float(tot_g / tot_r) - 1) < 5e-2, (s, float(tot_g), float(tot_r))
        for n, r in zip(names, grads):
            if float(r.norm()) < 1e-8 * float(tot_r):
                # mathematically zero (softmax is invariant to a constant added to every key's logit): f32 noise in the oracle
                assert n.endswith(zero_ok), (n, float(r.norm()))
                assert float(got[n].norm()) < zero_tol * float(tot_r), (n, float(got[n].norm()), float(tot_r))
                continue
            c_ = cos(got[n], r)
            if c_ < worst[0]:
                worst = (c_, (s, n))
            if c_ < 0.99:
                bad.append((s, n, round(c_, 4), float(r.norm() / tot_r)))
    assert not bad, (len(bad), bad[:12])
    return worst
