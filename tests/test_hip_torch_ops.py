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


def test_conv_and_groupnorm_ops_are_differentiable_and_match_torch(dev):
    """siss::conv2d_3x3 / siss::groupnorm_silu with autograd through the HIP dgrad / wgrad / GroupNorm-backward
    kernels, composed like a resnet half (GN -> SiLU -> conv), against the torch modules on bf16-rounded operands."""
    import torch.nn.functional as F
    bf = lambda t: t.to(torch.bfloat16).float()
    g = torch.Generator().manual_seed(3)
    n, c, co, hw = 2, 64, 128, 16
    x = bf(torch.randn(n, c, hw, hw, generator=g))
    gamma, beta = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    w = bf(torch.randn(co, c, 3, 3, generator=g) / (3 * c ** 0.5))
    b = 0.1 * torch.randn(co, generator=g)
    dy = bf(torch.randn(n, co, hw, hw, generator=g))
    leaves = [t.clone().requires_grad_(True) for t in (x, gamma, beta, w, b)]
    ref = F.conv2d(F.silu(F.group_norm(leaves[0], 32, leaves[1], leaves[2], 1e-6)), leaves[3], leaves[4], padding=1)
    ref.backward(dy)
    d = [t.to(dev).clone().requires_grad_(True) for t in (x, gamma, beta, w, b)]
    a = torch.ops.siss.groupnorm_silu(d[0], d[1], d[2], 32, 1e-6, True)
    y = torch.ops.siss.conv2d_3x3(a, d[3], d[4])
    y.backward(dy.to(dev))

    def close(got, want, rel, what):
        err, scale = (got.cpu() - want).abs().max().item(), want.abs().max().item()
        assert err <= rel * scale, (what, err, scale)
    close(y.detach(), ref.detach(), 2e-2, "y")
    for i, (name, rel) in enumerate((("dx", 3e-2), ("dgamma", 1e-2), ("dbeta", 1e-2), ("dW", 1e-2), ("dbias", 5e-3))):
        close(d[i].grad, leaves[i].grad, rel, name)
    torch.library.opcheck(torch.ops.siss.conv2d_3x3.default, (a.detach(), d[3].detach(), d[4].detach()),
                          test_utils=("test_schema", "test_faketensor"))
    torch.library.opcheck(torch.ops.siss.groupnorm_silu.default, (d[0].detach(), d[1].detach(), d[2].detach(), 32, 1e-6, True),
                          test_utils=("test_schema", "test_faketensor"))


@pytest.mark.parametrize("B,Sq,Sk,heads,D", [(2, 256, 256, 8, 40), (2, 64, 77, 4, 80), (1, 1024, 1024, 2, 64)])
def test_attention_op_matches_sdpa_forward_and_backward(dev, B, Sq, Sk, heads, D):
    """siss::attention (batched MFMA GEMMs + whole-row softmax; backward with the fused dS epilogue) against
    torch's scaled_dot_product_attention on bf16-rounded q / k / v: self- and cross-attention shapes of the SD UNet."""
    import torch.nn.functional as F
    bf = lambda t: t.to(torch.bfloat16).float()
    g = torch.Generator().manual_seed(B * 100 + Sq + D)
    C = heads * D
    q, k, v = (bf(torch.randn(B, s, C, generator=g)) for s in (Sq, Sk, Sk))
    do = bf(torch.randn(B, Sq, C, generator=g))
    scale = D ** -0.5
    leaves = [t.clone().requires_grad_(True) for t in (q, k, v)]
    split = lambda t: t.view(B, t.shape[1], heads, D).transpose(1, 2)
    ref = F.scaled_dot_product_attention(split(leaves[0]), split(leaves[1]), split(leaves[2]), scale=scale)
    ref = ref.transpose(1, 2).reshape(B, Sq, C)
    ref.backward(do)
    d = [t.to(dev).clone().requires_grad_(True) for t in (q, k, v)]
    out = torch.ops.siss.attention(d[0], d[1], d[2], heads, scale)
    out.backward(do.to(dev))

    def close(got, want, rel, what):
        err, sc = (got.cpu() - want).abs().max().item(), want.abs().max().item()
        assert err <= rel * sc, (what, err, sc)
    close(out.detach(), ref.detach(), 2e-2, "out")
    for i, name in enumerate(("dq", "dk", "dv")):
        close(d[i].grad, leaves[i].grad, 3e-2, name)
