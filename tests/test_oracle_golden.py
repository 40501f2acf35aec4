"""Pin the CPU oracle (oracle/loss.py, oracle/step.py) against the golden vectors that
oracle/make_golden.py produced by running the reference's own
losses/ddpm_deletion_loss.py in the build container."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import schedule as S
from oracle.loss import OracleDeletionLoss
from oracle.step import unlearning_step
from oracle.toy import ToyEps

HERE = os.path.join(os.path.dirname(__file__), "golden")
LOSS_FILES = sorted(glob.glob(os.path.join(HERE, "siss_loss_*.npz")))
STEP_FILES = sorted(glob.glob(os.path.join(HERE, "*_step*.npz")))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _loss_obj():
    ac = S.alphas_cumprod()
    return ac, OracleDeletionLoss(*S.gamma_sigma(ac))


def test_fixture_inventory():
    assert len(LOSS_FILES) == 7 and len(STEP_FILES) == 8


def test_schedule_known_answers():
    ac = S.alphas_cumprod()
    g, s = S.gamma_sigma(ac)
    # SURVEY Appendix C probe values
    assert abs(float(g[999]) - 0.0063528) < 1e-6 and abs(float(s[999]) - 0.9999798) < 1e-6
    assert abs(float(g[0]) - 0.99995) < 1e-5 and abs(float(s[0]) - 0.0100008) < 1e-6


@pytest.mark.parametrize("path", LOSS_FILES, ids=[os.path.basename(p)[10:-4] for p in LOSS_FILES])
def test_loss_matches_reference(path):
    z = np.load(path)
    ac, L = _loss_obj()
    x0, a0, noise, t, u = (_t(z[k]) for k in ("x0", "a0", "noise", "t", "u"))
    lambd = float(z["lambd"])
    net = ToyEps(x0.shape[1], seed=int(z["net_seed"]))
    keep = {"og_latents": x0, "noisy_latents": S.add_noise(ac, x0, noise, t)}
    forget = {"og_latents": a0, "noisy_latents": S.add_noise(ac, a0, noise, t)}
    torch.testing.assert_close(keep["noisy_latents"], _t(z["noisy_keep"]), rtol=0, atol=0)
    out = L.importance_sampling_with_mixture(net, t, noise, {}, keep, forget, lambd, u=u)
    assert out[0] is None
    for got, key in zip(out[1:], ("loss_x", "loss_a", "iw_x", "iw_a", "weighted_loss_x", "weighted_loss_a")):
        torch.testing.assert_close(got, _t(z[key]), rtol=1e-6, atol=0, equal_nan=True)
    # free known-answer invariant: (1-l) iw_x + l iw_a == 1
    inv = (1 - lambd) * out[3] + lambd * out[4]
    torch.testing.assert_close(inv, torch.ones_like(inv), rtol=1e-5, atol=1e-6)
    o2 = L.double_forward_with_neg_del(net, t, noise, {}, keep, forget)
    torch.testing.assert_close(o2[1], _t(z["no_is_loss_x"]), rtol=1e-6, atol=0)
    torch.testing.assert_close(o2[2], _t(z["no_is_loss_a"]), rtol=1e-6, atol=0)
    assert o2[0] is None and o2[3] is None and o2[4] is None
    o3 = L.simple_neg_del(net, t, noise, {}, keep, forget, superfactor=3.0)
    torch.testing.assert_close(o3[0], _t(z["neg_loss"]), rtol=1e-6, atol=0)
    o4 = L.naive_del(net, t, noise, {}, keep, forget)
    torch.testing.assert_close(o4[0], _t(z["naive_loss"]), rtol=1e-6, atol=0)
    # ErasEDiff (:70-78): loss_a against the recorded uniform target the reference drew with torch.rand_like
    from oracle.make_golden import inject_rand_like
    with inject_rand_like(_t(z["erase_target"])):
        o6 = L.erasediff(net, t, noise, {}, keep, forget)
    torch.testing.assert_close(o6[1], _t(z["erase_loss_x"]), rtol=1e-6, atol=0)
    torch.testing.assert_close(o6[2], _t(z["erase_loss_a"]), rtol=1e-6, atol=0)
    assert o6[0] is None and o6[5] is o6[1] and o6[6] is o6[2]
    if lambd < 1.0:
        o5 = L.subscore_bernoulli(net, t, noise, {}, keep, forget, lambd, u=u)
        torch.testing.assert_close(o5[1], _t(z["bern_loss_x"]), rtol=1e-6, atol=0)
        torch.testing.assert_close(o5[2], _t(z["bern_loss_a"]), rtol=1e-6, atol=0)
    else:
        with pytest.raises(ZeroDivisionError):
            L.subscore_bernoulli(net, t, noise, {}, keep, forget, lambd, u=u)


def test_saturating_case_has_exact_weights():
    z = np.load(os.path.join(HERE, "siss_loss_tsmall_saturating_c3_h32_l05.npz"))
    iwx, iwa = z["iw_x"], z["iw_a"]
    assert set(np.unique(iwx)) <= {0.0, 2.0} and set(np.unique(iwa)) <= {0.0, 2.0}
    assert np.all(iwx + iwa == 2.0)


@pytest.mark.parametrize("path", STEP_FILES, ids=[os.path.basename(p)[:-4] for p in STEP_FILES])
def test_step_matches_reference(path):
    z = np.load(path)
    ac, L = _loss_obj()
    ga = int(z["ga"])
    from oracle.make_golden import STEP_ETA, seeded_rand_like, step_loss_params
    loss_fn = {"no_is_step": "double_forward_with_neg_del", "erasediff_step": "erasediff", "neg_grad_step": "simple_neg_del",
               "naive_step": "naive_del", "bernoulli_step": "subscore_bernoulli"}.get(
        os.path.basename(path)[:-4], "importance_sampling_with_mixture")
    net = ToyEps(3, seed=int(z["net_seed"]))
    opt = torch.optim.AdamW(net.parameters(), lr=float(z["lr"]), betas=(0.95, 0.999),
                            weight_decay=1e-6, eps=1e-8)
    for step in range(2):
        mbs = [{k: _t(z[f"s{step}_m{m}_{k}"]) for k in ("x0", "a0", "noise", "t", "u")}
               for m in range(ga)]
        lp = step_loss_params(loss_fn)
        with seeded_rand_like(977 + step):           # ErasEDiff's uniform targets: the sequence the fixture was made with
            st, gx, ga_, g = unlearning_step(net, opt, L, loss_fn, ac, mbs, train_batch_size=4,
                                             scaling_norm=float(z["scaling_norm"]), loss_params=lp,
                                             eta=STEP_ETA if loss_fn == "erasediff" else None)
        ref = z[f"s{step}_stats"]
        got = np.array([st.norm_loss_x, st.norm_loss_a, st.scaling_factor, st.pre_clip_norm,
                        st.weighted_loss_x, st.weighted_loss_a])
        np.testing.assert_allclose(got, ref, rtol=2e-5, equal_nan=True)      # single-loss objectives have no g_x / g_a norms
        for n, p in net.named_parameters():
            torch.testing.assert_close(g[n], _t(z[f"s{step}_g/{n}"]), rtol=1e-4, atol=1e-7)
            if gx is not None:
                torch.testing.assert_close(gx[n], _t(z[f"s{step}_gx/{n}"]), rtol=1e-4, atol=1e-7)
            else:
                assert f"s{step}_gx/{n}" not in z.files
            torch.testing.assert_close(p.detach(), _t(z[f"s{step}_param/{n}"]), rtol=1e-5, atol=1e-7)
