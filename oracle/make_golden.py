"""Generate tests/golden/*.npz by running the REFERENCE's own loss code (build container only).

Run from the repo root:  ``python -m oracle.make_golden``

/root/reference exists only in the build container; its Python never travels to
the GPU box.  This script imports /root/reference/losses/ddpm_deletion_loss.py
(torch-only), drives it with seeded inputs and a small deterministic stand-in
network for ``unet``, and stores inputs + outputs as small .npz fixtures.  The
fixtures pin oracle/loss.py and oracle/step.py (tests/test_oracle_golden.py) and,
through them, the HIP path.

The keep/forget draw (`torch.rand(B)`, ddpm_deletion_loss.py:18,:101) is made
reproducible by patching ``torch.rand`` to hand back the recorded uniforms ``u``
for the duration of the reference call -- the reference source is untouched.
"""
import contextlib
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _ref_loss_class():
    sys.path.insert(0, REF)
    from losses.ddpm_deletion_loss import DDPMDeletionLoss  # noqa: E402
    return DDPMDeletionLoss


@contextlib.contextmanager
def inject_uniforms(u):
    real = torch.rand

    def fake(*size, **kw):
        n = size[0] if isinstance(size[0], int) else size[0][0]
        assert n == u.shape[0], (size, u.shape)
        return u.clone()
    torch.rand = fake
    try:
        yield
    finally:
        torch.rand = real


@contextlib.contextmanager
def inject_rand_like(target):
    """ErasEDiff draws its uniform target with torch.rand_like (ddpm_deletion_loss.py:75): hand back a recorded one."""
    real = torch.rand_like

    def fake(t, **kw):
        assert tuple(t.shape) == tuple(target.shape), (t.shape, target.shape)
        return target.clone().to(t.dtype)
    torch.rand_like = fake
    try:
        yield
    finally:
        torch.rand_like = real


@contextlib.contextmanager
def seeded_rand_like(seed):
    """torch.rand_like drawn from a private seeded generator: the same target sequence for the reference (when the
    fixture is made) and for the oracle (when it is checked), across the micro-batches of a step."""
    real = torch.rand_like
    gen = torch.Generator().manual_seed(seed)

    def fake(t, **kw):
        return torch.rand(t.shape, generator=gen).to(t.dtype)
    torch.rand_like = fake
    try:
        yield
    finally:
        torch.rand_like = real


class RefLossWithU:
    """Adapter: same surface, but accepts ``u=`` and injects it into the reference."""

    def __init__(self, ref):
        self.ref = ref

    def __getattr__(self, name):
        fn = getattr(self.ref, name)

        def call(*a, u=None, **kw):
            if u is None:
                return fn(*a, **kw)
            with inject_uniforms(u):
                return fn(*a, **kw)
        return call


from oracle.toy import ToyEps  # noqa: E402


def _np(d):
    out = {}
    for k, v in d.items():
        if v is None:
            continue
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


def loss_cases():
    from oracle import schedule as S
    Ref = _ref_loss_class()
    ac = S.alphas_cumprod()
    gamma, sigma = S.gamma_sigma(ac)
    ref = RefLossWithU(Ref(gamma, sigma))
    cases = {}
    specs = [  # name, C, HW, t-mode, lambd
        ("t999_c3_h32_l05", 3, 32, "t999", 0.5),
        ("t999_c1_h8_l03", 1, 8, "t999", 0.3),
        ("tuniform_c3_h8_l05", 3, 8, "uniform", 0.5),
        ("tsmall_saturating_c3_h32_l05", 3, 32, "small", 0.5),
        ("t999_c3_h8_l00", 3, 8, "t999", 0.0),
        ("t999_c3_h8_l10", 3, 8, "t999", 1.0),
        ("tmid_c1_h32_l03", 1, 32, "mid", 0.3),
    ]
    for i, (name, c, hw, tmode, lambd) in enumerate(specs):
        g = torch.Generator().manual_seed(1000 + i)
        B = 4
        x0 = torch.rand(B, c, hw, hw, generator=g) * 2 - 1
        a0 = (torch.rand(1, c, hw, hw, generator=g) * 2 - 1).repeat(B, 1, 1, 1)
        noise = torch.randn(B, c, hw, hw, generator=g)
        if tmode == "t999":
            t = torch.full((B,), 999, dtype=torch.long)
        elif tmode == "uniform":
            t = torch.randint(0, 1000, (B,), generator=g)
        elif tmode == "small":
            t = torch.tensor([10, 3, 50, 0])
        else:
            t = torch.tensor([500, 400, 650, 300])
        u = torch.rand(B, generator=g)
        net = ToyEps(c, seed=7 + i)
        keep = {"og_latents": x0, "noisy_latents": S.add_noise(ac, x0, noise, t)}
        forget = {"og_latents": a0, "noisy_latents": S.add_noise(ac, a0, noise, t)}
        out = ref.importance_sampling_with_mixture(net, t, noise, {}, keep, forget, lambd, u=u)
        _, lx, la, iwx, iwa, wlx, wla = out
        x_mix = torch.where((u > lambd)[:, None, None, None], keep["noisy_latents"], forget["noisy_latents"])
        pred = net(x_mix, t)[0]
        cases[name] = _np(dict(
            x0=x0, a0=a0, noise=noise, t=t, u=u, lambd=lambd, net_seed=7 + i,
            noisy_keep=keep["noisy_latents"], noisy_forget=forget["noisy_latents"],
            x_mix=x_mix, pred=pred, loss_x=lx, loss_a=la, iw_x=iwx, iw_a=iwa,
            weighted_loss_x=wlx, weighted_loss_a=wla))
        # the other objectives on the same inputs (class-surface pins, a-2 / a-4)
        o2 = ref.double_forward_with_neg_del(net, t, noise, {}, keep, forget)
        cases[name].update(_np(dict(no_is_loss_x=o2[1], no_is_loss_a=o2[2])))
        o3 = ref.simple_neg_del(net, t, noise, {}, keep, forget, superfactor=3.0)
        cases[name].update(_np(dict(neg_loss=o3[0])))
        o4 = ref.naive_del(net, t, noise, {}, keep, forget)
        cases[name].update(_np(dict(naive_loss=o4[0])))
        et = torch.rand(pred.shape, generator=g)           # ErasEDiff's uniform target, recorded (a-4)
        with inject_rand_like(et):
            o6 = ref.erasediff(net, t, noise, {}, keep, forget)
        cases[name].update(_np(dict(erase_target=et, erase_loss_x=o6[1], erase_loss_a=o6[2])))
        if lambd < 1.0:   # the reference raises ZeroDivisionError at lambd == 1 (:110)
            o5 = ref.subscore_bernoulli(net, t, noise, {}, keep, forget, lambd, u=u)
            cases[name].update(_np(dict(bern_loss_x=o5[1], bern_loss_a=o5[2])))
    return cases


STEP_ETA = 1.5        # large enough that the ErasEDiff branch s = -max(eta - <g_x, g_a> / |g_a|^2, 0) is non-zero in the fixture


def step_loss_params(loss_fn):
    return {"importance_sampling_with_mixture": {"lambd": 0.5}, "subscore_bernoulli": {"lambd": 0.5},
            "simple_neg_del": {"superfactor": 2.0}}.get(loss_fn, {})


def step_cases():
    from oracle import schedule as S
    from oracle.step import unlearning_step
    Ref = _ref_loss_class()
    ac = S.alphas_cumprod()
    gamma, sigma = S.gamma_sigma(ac)
    ref = RefLossWithU(Ref(gamma, sigma))
    cases = {}
    for name, loss_fn, ga, tmode in [
            ("siss_step_ga1", "importance_sampling_with_mixture", 1, "t999"),
            ("siss_step_ga2", "importance_sampling_with_mixture", 2, "t999"),
            ("siss_step_tuniform", "importance_sampling_with_mixture", 1, "uniform"),
            ("no_is_step", "double_forward_with_neg_del", 1, "t999"),
            # the baselines of delete_celeb.yaml's `deletion.loss_fn` choices (a-4), GA = 2 where the loop differs
            ("erasediff_step", "erasediff", 2, "t999"),
            ("neg_grad_step", "simple_neg_del", 1, "t999"),
            ("naive_step", "naive_del", 2, "uniform"),
            ("bernoulli_step", "subscore_bernoulli", 1, "t999")]:
        g = torch.Generator().manual_seed(4242 + ga + len(name))
        B, c, hw = 4, 3, 8
        net = ToyEps(c, seed=11)
        opt = torch.optim.AdamW(net.parameters(), lr=5e-3, betas=(0.95, 0.999),
                                weight_decay=1e-6, eps=1e-8)
        rec = {"lambd": 0.5, "scaling_norm": 5.0, "ga": ga, "lr": 5e-3, "net_seed": 11}
        for step in range(2):
            mbs = []
            for k in range(ga):
                x0 = torch.rand(B, c, hw, hw, generator=g) * 2 - 1
                a0 = (torch.rand(1, c, hw, hw, generator=g) * 2 - 1).repeat(B, 1, 1, 1)
                noise = torch.randn(B, c, hw, hw, generator=g)
                t = (torch.full((B,), 999, dtype=torch.long) if tmode == "t999"
                     else torch.randint(0, 1000, (B,), generator=g))
                u = torch.rand(B, generator=g)
                mbs.append(dict(x0=x0, a0=a0, noise=noise, t=t, u=u))
                for kk, vv in mbs[-1].items():
                    rec[f"s{step}_m{k}_{kk}"] = vv
            lp = step_loss_params(loss_fn)
            with seeded_rand_like(977 + step):
                st, gx, ga_, gfin = unlearning_step(
                    net, opt, ref, loss_fn, ac, mbs, train_batch_size=B, scaling_norm=5.0,
                    loss_params=lp, eta=STEP_ETA if loss_fn == "erasediff" else None)
            rec[f"s{step}_stats"] = np.array([st.norm_loss_x, st.norm_loss_a, st.scaling_factor,
                                              st.pre_clip_norm, st.weighted_loss_x, st.weighted_loss_a])
            for n in gfin:
                if gx is not None:
                    rec[f"s{step}_gx/{n}"] = gx[n]
                    rec[f"s{step}_ga/{n}"] = ga_[n]
                rec[f"s{step}_g/{n}"] = gfin[n]
            for n, p in net.named_parameters():
                rec[f"s{step}_param/{n}"] = p.detach().clone()
        cases[name] = _np(rec)
    return cases


def _ref_module(rel):
    """Import ONE reference file by path (torch + numpy only; nothing else of the reference package is touched)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_" + os.path.basename(rel)[:-3], os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


SAMPLER_CASES = [  # (dataset length, rank, num_replicas, shuffle, seed, window_size, indices to record)
    (1000, 0, 1, True, 0, 0.5, 10000), (1000, 0, 8, True, 0, 0.5, 10000), (1000, 3, 8, True, 0, 0.5, 10000),
    (37, 0, 1, True, 0, 0.5, 2000), (37, 1, 2, True, 7, 0.5, 2000), (64, 0, 1, True, 0, 0.0, 500),
    (64, 2, 4, False, 0, 0.5, 500), (3, 0, 1, True, 0, 0.5, 100), (1, 0, 1, True, 0, 0.5, 10),
]


def sampler_cases():
    """The data-input step in front of the path: index sequences of the reference's
    data/utils/infinite_sampler.py:4-35 (rank / num_replicas as the data-parallel shards use them) and
    data/utils/repeat_sampler.py:4-21 (the celeb forget set), recorded from the reference's own classes."""
    import itertools
    inf = _ref_module("data/utils/infinite_sampler.py").InfiniteSampler
    rep = _ref_module("data/utils/repeat_sampler.py").RepeatedSampler
    rec = {}
    for (n, rank, world, shuffle, seed, win, count) in SAMPLER_CASES:
        ds = list(range(n))
        # torch >= 2.4 removed Sampler.__init__(data_source), which the reference's __init__ (written for torch 2.2.1,
        # environment.yml:283) still calls: build the object without it and store the six attributes __init__ stores
        # (infinite_sampler.py:11-16); the index ALGORITHM under test is the reference's own __iter__ (:18-35).
        smp = inf.__new__(inf)
        smp.dataset, smp.rank, smp.num_replicas, smp.shuffle, smp.seed, smp.window_size = ds, rank, world, shuffle, seed, win
        idx = np.fromiter((int(i) for i in itertools.islice(iter(smp), count)), dtype=np.int32, count=count)
        rec[f"inf_n{n}_r{rank}of{world}_sh{int(shuffle)}_s{seed}_w{win}"] = idx
    for (n, repeats) in ((1, 16), (3, 4), (5, 1)):
        smp = rep(list(range(n)), repeats)
        rec[f"rep_n{n}_x{repeats}"] = np.array(list(iter(smp)), dtype=np.int32)
        assert len(smp) == n * repeats
    return rec


def main():
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "samplers.npz"), **sampler_cases())
    for name, rec in loss_cases().items():
        np.savez_compressed(os.path.join(OUT, f"siss_loss_{name}.npz"), **rec)
    for name, rec in step_cases().items():
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **rec)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
