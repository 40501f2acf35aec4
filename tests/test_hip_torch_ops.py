"""The tensor kernels of the path as PyTorch custom ops (``torch.ops.siss.*``, siss_amd/torch_ops.py): same results
as the direct wrappers / golden vectors, valid schemas and fake implementations (``torch.library.opcheck``)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import siss_amd.torch_ops  # noqa: F401  (registers torch.ops.siss.*)
    return torch.device("cuda:0")


def test_ops_are_registered_and_pass_opcheck(dev):
    g = torch.Generator(device=dev).manual_seed(0)
    B = 4
    x0 = torch.rand(B, 3, 16, 16, device=dev, generator=g) * 2 - 1
    a0 = torch.rand(B, 3, 16, 16, device=dev, generator=g) * 2 - 1
    noise = torch.randn(B, 3, 16, 16, device=dev, generator=g)
    t = torch.full((B,), 999, dtype=torch.long, device=dev)
    u = torch.rand(B, device=dev, generator=g)
    ac = torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, 1000), 0).to(dev)
    args = (x0, a0, noise, t, u, ac, 0.5)
    torch.library.opcheck(torch.ops.siss.mixture_fwd.default, args, test_utils=("test_schema", "test_faketensor"))
    x_mix, iw_x, iw_a, gam, sig = torch.ops.siss.mixture_fwd(*args)
    assert torch.allclose(0.5 * iw_x + 0.5 * iw_a, torch.ones(B, device=dev), atol=1e-5)
    pred = torch.randn(B, 3, 16, 16, device=dev, generator=g)
    args2 = (pred, x_mix, x0, a0, gam, sig, iw_x, iw_a, 0.25)
    torch.library.opcheck(torch.ops.siss.loss_bwd_seed.default, args2, test_utils=("test_schema", "test_faketensor"))
    cx, ca, sx, sa = torch.ops.siss.loss_bwd_seed(*args2)
    eps_x = (x_mix - gam.view(B, 1, 1, 1) * x0) / sig.view(B, 1, 1, 1)
    torch.testing.assert_close(cx, 2 * 0.25 * iw_x.view(B, 1, 1, 1) * (pred - eps_x), rtol=1e-4, atol=1e-5)
    c, s = torch.ops.siss.mse_bwd_seed(pred, noise, 0.5)
    torch.testing.assert_close(c, pred - noise, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(s, (pred - noise).square().sum((1, 2, 3)), rtol=1e-4, atol=1e-4)
    out = torch.ops.siss.ddpm_step(x0, pred, noise, 0.9, 0.4, 0.3, 0.6, 0.1, True)
    ref = 0.3 * ((x0 - 0.4 * pred) / 0.9).clamp(-1, 1) + 0.6 * x0 + 0.1 * noise
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-5)


def test_mixture_op_matches_reference_golden_vectors(dev):
    """The same golden vectors (generated from the reference's own loss code) through the registered op."""
    files = sorted(glob.glob(os.path.join(GOLD, "siss_loss_*.npz")))
    assert files
    for path in files:
        z = np.load(path)
        f = lambda k: torch.from_numpy(z[k]).to(dev)
        ac = f("alphas_cumprod") if "alphas_cumprod" in z else torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, 1000), 0).to(dev)
        if not all(k in z for k in ("x0", "a0", "noise", "t", "u", "iw_x", "iw_a")):
            continue
        lambd = float(z["lambd"]) if "lambd" in z else 0.5
        x_mix, iw_x, iw_a, _, _ = torch.ops.siss.mixture_fwd(f("x0"), f("a0"), f("noise"), f("t"), f("u"), ac, lambd)
        torch.testing.assert_close(iw_x.cpu(), torch.from_numpy(z["iw_x"]).float(), rtol=2e-4, atol=1e-30)
        torch.testing.assert_close(iw_a.cpu(), torch.from_numpy(z["iw_a"]).float(), rtol=2e-4, atol=1e-30)


def test_flat_optimizer_op_matches_torch(dev):
    g = torch.Generator(device=dev).manual_seed(1)
    n = 10_000
    gx, ga = torch.randn(n, device=dev, generator=g), torch.randn(n, device=dev, generator=g)
    p = torch.randn(n, device=dev, generator=g)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    from siss_amd import lib
    scalars = torch.zeros(lib.query("siss_opt_scalars_words"), device=dev)
    partials = torch.zeros(lib.query("siss_opt_partials_words"), dtype=torch.float64, device=dev)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.95, 0.999), eps=1e-8, weight_decay=1e-2)
    s = 5.0 / float(ga.norm())
    ref.grad = gx - s * ga
    torch.nn.utils.clip_grad_norm_([ref], 1.0)
    opt.step()
    torch.ops.siss.recombine_clip_adamw_(gx, ga, p, m, v, scalars, partials, 5.0, 1.0, 1e-3, 0.95, 0.999, 1e-8, 1e-2)
    torch.testing.assert_close(p, ref.detach(), rtol=1e-5, atol=1e-6)
