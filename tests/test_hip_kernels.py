"""Per-kernel parity: HIP kernels (through the C-ABI) vs plain torch fp32 on the same inputs.

Tolerances: bf16 operands + f32 accumulate, outputs rounded to bf16 -> rel 1e-2 of the
tensor's scale for activations; f32 outputs (wgrad, optimizer) rel 2e-3 / 1e-5 as stated.
"""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

HERE = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from siss_amd import lib
    lib.load()
    return torch.device("cuda:0")


def _bf(x):
    return x.to(torch.bfloat16).float()


def _close(got, ref, rel, what=""):
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g} (rel {err / scale:.3g} > {rel})"


# ------------------------------------------------------------------ conv (NT / TN GEMM)
@pytest.mark.parametrize("n,h,w,ci,co", [(2, 16, 16, 128, 128), (1, 8, 8, 256, 128), (2, 12, 20, 64, 192),
                                         (1, 32, 32, 128, 256)])
def test_conv3x3_fprop_dgrad_wgrad(dev, n, h, w, ci, co):
    from siss_amd import ops
    from siss_amd.layout import Act
    g = torch.Generator().manual_seed(n * 1000 + h + ci)
    x = _bf(torch.randn(n, ci, h, w, generator=g))
    wt = _bf(torch.randn(co, ci, 3, 3, generator=g) * (1.0 / (3 * ci ** 0.5)))
    bias = torch.randn(co, generator=g)
    temb = torch.randn(n, co, generator=g)
    res = _bf(torch.randn(n, co, h, w, generator=g))
    # reference (fp32 math on bf16-rounded operands)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, bias, padding=1) + temb[:, :, None, None] + res
    dy = _bf(torch.randn(n, co, h, w, generator=g))
    y_ref.backward(dy)

    xa = Act.from_nchw(x, dev)
    ra = Act.from_nchw(res, dev)
    wn = ops.conv_w_to_native(wt).to(dev)
    out = Act(n, h, w, co, dev)
    out.buf.fill_(7.0)   # poison: the kernel must write zeros into the halo itself
    out.buf[: out.guard * co] = 0
    out.buf[-out.guard * co:] = 0
    ops.conv_fprop(xa, wn.to(torch.bfloat16), out, bias=bias.to(dev), rowbias=temb.to(dev), residual=ra)
    torch.cuda.synchronize()
    assert out.halo_is_zero()
    _close(out.to_nchw().cpu(), y_ref.detach(), 1e-2, "fprop")

    dya = Act.from_nchw(dy, dev)
    dx = Act(n, h, w, ci, dev)
    ops.conv_dgrad(dya, ops.dgrad_weight(wn), dx)
    torch.cuda.synchronize()
    assert dx.halo_is_zero()
    _close(dx.to_nchw().cpu(), xr.grad, 1e-2, "dgrad")

    dW = torch.zeros(1, 9, co, ci, device=dev)
    ops.conv_wgrad(dya, xa, dW, nsets=1)
    torch.cuda.synchronize()
    _close(ops.conv_w_from_native(dW[0]).cpu(), wr.grad, 2e-3, "wgrad")


def test_wgrad_two_sets_shared_activation(dev):
    """Dual-cotangent backward: two cotangent sets against ONE saved activation (SISS)."""
    from siss_amd import ops
    from siss_amd.layout import Act
    g = torch.Generator().manual_seed(5)
    b, h, w, ci, co = 2, 16, 16, 128, 128
    x = _bf(torch.randn(b, ci, h, w, generator=g))
    dy = _bf(torch.randn(2 * b, co, h, w, generator=g))
    ref = []
    for s in range(2):
        wr = torch.zeros(co, ci, 3, 3, requires_grad=True)
        F.conv2d(x, wr, padding=1).backward(dy[s * b:(s + 1) * b])
        ref.append(wr.grad)
    dW = torch.zeros(2, 9, co, ci, device=dev)
    ops.conv_wgrad(Act.from_nchw(dy, dev), Act.from_nchw(x, dev), dW, nsets=2)
    # and the unshared form (No-IS): activation has 2b images
    x2 = torch.cat([x, x.flip(0)], 0)
    dW2 = torch.zeros(2, 9, co, ci, device=dev)
    ops.conv_wgrad(Act.from_nchw(dy, dev), Act.from_nchw(x2, dev), dW2, nsets=2)
    torch.cuda.synchronize()
    for s in range(2):
        _close(ops.conv_w_from_native(dW[s]).cpu(), ref[s], 2e-3, f"set{s}")
    wr = torch.zeros(co, ci, 3, 3, requires_grad=True)
    F.conv2d(x.flip(0), wr, padding=1).backward(dy[b:])
    _close(ops.conv_w_from_native(dW2[1]).cpu(), wr.grad, 2e-3, "unshared set1")


def test_conv1x1(dev):
    from siss_amd import ops
    from siss_amd.layout import Act
    g = torch.Generator().manual_seed(9)
    n, h, w, ci, co = 2, 8, 8, 256, 128
    x = _bf(torch.randn(n, ci, h, w, generator=g))
    wt = _bf(torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5)
    b = torch.randn(co, generator=g)
    ref = F.conv2d(x, wt, b)
    out = Act(n, h, w, co, dev)
    ops.conv_fprop(Act.from_nchw(x, dev), ops.conv_w_to_native(wt).to(dev).to(torch.bfloat16), out,
                   bias=b.to(dev), ksize=1)
    torch.cuda.synchronize()
    _close(out.to_nchw().cpu(), ref, 1e-2, "1x1")
    assert out.halo_is_zero()


# ------------------------------------------------------------------ fused SISS pre/post kernels
LOSS_FILES = sorted(glob.glob(os.path.join(HERE, "siss_loss_*.npz")))


@pytest.mark.parametrize("accumulate", [False, True])
def test_gemm_with_depth_to_space_epilogue_equals_gemm_then_scatter(dev, accumulate):
    """siss_gemm_nt_d2s (the downsample dgrad's plane products writing straight into the full-resolution cotangent, adding the
    one x already carries) against the same product into a plane buffer followed by siss_depth_to_space: the same roundings in
    the same order, so BITWISE equal; the halo of the full-resolution tensor is untouched (Downsample2D backward:
    delete_celeb.py:691,:702 differentiate through it)."""
    from siss_amd import lib, ops
    from siss_amd.layout import Act
    n, ho, wo, c = 3, 24, 20, 128                            # plane tensors 24 x 20, full resolution 48 x 40; 3 images per tile row run
    g = torch.Generator().manual_seed(11 + int(accumulate))
    dy = Act.from_nchw(torch.randn(n, c, ho, wo, generator=g).bfloat16().float(), dev)
    prior = torch.randn(n, c, 2 * ho, 2 * wo, generator=g).bfloat16().float()
    ref_dx, fused_dx = Act.from_nchw(prior, dev), Act.from_nchw(prior, dev)
    if not accumulate:                                       # fresh buffers: interior garbage that must be overwritten, zero halo
        for a in (ref_dx, fused_dx):
            a.interior().fill_(5.0)
    dz = Act(n, ho, wo, 4 * c, dev)
    wp = wo + 2
    taps = {0: [(0, 0), (1, 0), (0, 1), (1, 1)], 1: [(0, 0), (1, 0)], 2: [(0, 0), (0, 1)], 3: [(0, 0)]}   # (dy, dx) row shifts per plane
    for plane, tl in taps.items():
        w = (torch.randn(len(tl), c, c, generator=g) * 0.05).to(dev).to(torch.bfloat16)
        shifts = [-(a * wp + b) for a, b in tl]
        ops.gemm_nt(lib.ptr(dy.data), c, w, lib.ptr(dz.data[:, plane * c:]), 4 * c, dy.rows, c, c, shifts, [0] * len(tl),
                    rows_per_image=dy.rows_per_image, hp=dy.hp, wp=dy.wp)
        lib.call("siss_gemm_nt_d2s", dy.data, c, w, fused_dx.data, c, fused_dx.data if accumulate else None, c, dy.rows, c, c,
                 len(tl), lib.int_array(shifts), lib.int_array([0] * len(tl)), dy.rows_per_image, dy.hp, dy.wp, plane)
    lib.call("siss_depth_to_space", dz.data, ref_dx.data, int(accumulate), n, 2 * ho, 2 * wo, c)
    torch.cuda.synchronize()
    assert fused_dx.halo_is_zero()
    assert torch.equal(fused_dx.buf, ref_dx.buf)


@pytest.mark.parametrize("kind", ["downsample dgrad (1/2/2/4 taps, adds R in place)", "sub-pixel upsample forward (4 x 4 taps, bias)"])
@pytest.mark.parametrize("shape", [(3, 24, 20, 128), (2, 8, 8, 256), (9, 64, 64, 128)], ids=["24x20x128", "8x8x256", "64x64x128 large grid"])
def test_four_planes_in_one_launch_equal_four_launches(dev, kind, shape):
    """siss_gemm_nt_d2s_phases against the four single-plane launches it replaces (siss_gemm_nt_d2s / siss_gemm_nt_d2s_bias): every
    block runs the same K order and the same epilogue, so the full-resolution tensor is BITWISE the same (Downsample2D backward /
    Upsample2D forward behind losses/ddpm_deletion_loss.py:24 and delete_celeb.py:691,:702).  (Where the single-plane launches
    split K -- small grids with a long K loop -- the sums are formed in another order: one bf16 rounding apart.)"""
    from siss_amd import lib
    from siss_amd.layout import Act
    lib.ensure_workspace(dev)                              # (split-K needs it: the case below must not depend on test order)
    n, ho, wo, c = shape
    g = torch.Generator().manual_seed(5 + c + ho)
    a = Act.from_nchw(torch.randn(n, c, ho, wo, generator=g).bfloat16().float(), dev)
    wp = wo + 2
    down = kind.startswith("down")
    if down:
        taps = [[(0, 0)], [(0, 1), (0, 0)], [(1, 0), (0, 0)], [(1, 1), (1, 0), (0, 1), (0, 0)]]
    else:
        taps = [[(y + (pl >> 1) - 1, x + (pl & 1) - 1) for y in range(2) for x in range(2)] for pl in range(4)]
    shifts = [[dy_ * wp + dx_ for dy_, dx_ in tl] for tl in taps]
    npan = sum(len(tl) for tl in taps)
    w = (torch.randn(npan, c, c, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    bias = None if down else torch.randn(c, generator=g).to(dev)
    prior = torch.randn(n, c, 2 * ho, 2 * wo, generator=g).bfloat16().float()
    one, four = Act.from_nchw(prior, dev), Act.from_nchw(prior, dev)
    p0, pos = [0], 0
    for plane, sh in enumerate(shifts):
        if down:
            lib.call("siss_gemm_nt_d2s", a.data, c, w[pos:], four.data, c, four.data, c, a.rows, c, c, len(sh), lib.int_array(sh),
                     lib.int_array([0] * len(sh)), a.rows_per_image, a.hp, a.wp, plane)
        else:
            lib.call("siss_gemm_nt_d2s_bias", a.data, c, w[pos:], four.data, c, bias, a.rows, c, c, len(sh), lib.int_array(sh),
                     lib.int_array([0] * len(sh)), a.rows_per_image, a.hp, a.wp, plane)
        pos += len(sh)
        p0.append(pos)
    flat = [s_ for sh in shifts for s_ in sh]
    lib.call("siss_gemm_nt_d2s_phases", a.data, c, w, one.data, c, bias, one.data if down else None, c if down else 0, a.rows, c, c,
             lib.int_array(p0), lib.int_array(flat), lib.int_array([0] * npan), a.rows_per_image, a.hp, a.wp)
    torch.cuda.synchronize()
    assert one.halo_is_zero()
    if c * max(len(sh) for sh in shifts) // 64 >= 12 and a.rows <= 128 * 128:
        # the single-plane launches of this small grid split K (>= 12 K-steps on <= 128 tiles: f32 partial tiles summed by a second
        # kernel), the one launch has 4x the blocks and does not: the same products summed in another order, one bf16 rounding apart
        d = (one.buf.float() - four.buf.float()).abs()
        assert float(d.max()) <= 2.0 ** -7 * float(four.buf.float().abs().max())
        assert float((d > 0).float().mean()) < 0.05
    else:
        assert torch.equal(one.buf, four.buf)
    assert float((one.interior().float() - prior.to(dev).permute(0, 2, 3, 1)).abs().max()) > 0.1      # (it did write)


@pytest.mark.parametrize("path", LOSS_FILES, ids=[os.path.basename(p)[10:-4] for p in LOSS_FILES])
def test_mixture_and_loss_seed_vs_golden(dev, path):
    """fp32 mode against the golden vectors made by the reference's own loss code."""
    from siss_amd.loss import mixture_fwd, loss_bwd_seed
    from oracle import schedule as S
    z = np.load(path)
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    ac = S.alphas_cumprod()
    gam, sig = S.gamma_sigma(ac)
    lambd = float(z["lambd"])
    m = mixture_fwd(t("x0"), t("a0"), t("noise"), t("t"), t("u"), ac.to(dev), gam.to(dev), sig.to(dev), lambd)
    torch.testing.assert_close(m.x_mix.cpu(), torch.from_numpy(z["x_mix"]), rtol=0, atol=0)  # bit-exact
    torch.testing.assert_close(m.iw_x.cpu(), torch.from_numpy(z["iw_x"]), rtol=2e-4, atol=1e-30)
    torch.testing.assert_close(m.iw_a.cpu(), torch.from_numpy(z["iw_a"]), rtol=2e-4, atol=1e-30)
    B = z["x0"].shape[0]
    s = loss_bwd_seed(t("pred"), m, t("x0"), t("a0"), scale=1.0 / B, want_losses=True)
    torch.testing.assert_close(s.loss_x.cpu(), torch.from_numpy(z["loss_x"]), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(s.loss_a.cpu(), torch.from_numpy(z["loss_a"]), rtol=1e-5, atol=1e-6)
    # cotangents = d/dpred of sum(weighted_loss)/B
    wx = torch.from_numpy(z["iw_x"])[:, None, None, None]
    pred = torch.from_numpy(z["pred"]).requires_grad_(True)
    from oracle.loss import siss_terms
    ex, ea, *_ = siss_terms(torch.from_numpy(z["x_mix"]), torch.from_numpy(z["x0"]), torch.from_numpy(z["a0"]),
                            gam[torch.from_numpy(z["t"])], sig[torch.from_numpy(z["t"])], lambd)
    (wx * (pred - ex) ** 2).sum().div(B).backward()
    torch.testing.assert_close(s.c_x.cpu(), pred.grad, rtol=3e-4, atol=1e-7)
    torch.testing.assert_close(s.sum_loss_x.cpu(), torch.from_numpy(z["loss_x"]).sum(dim=[1, 2, 3]), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(s.sum_loss_a.cpu(), torch.from_numpy(z["loss_a"]).sum(dim=[1, 2, 3]), rtol=1e-5, atol=1e-6)


def test_mixture_bf16_mode_matches_torch_bf16(dev):
    """bf16 mode reproduces DDPMScheduler.add_noise on bf16 tensors bit-for-bit."""
    from siss_amd.loss import mixture_fwd
    from oracle import schedule as S
    g = torch.Generator().manual_seed(3)
    B, c, hw = 4, 3, 32
    x0 = (torch.rand(B, c, hw, hw, generator=g) * 2 - 1).to(torch.bfloat16)
    a0 = (torch.rand(1, c, hw, hw, generator=g) * 2 - 1).repeat(B, 1, 1, 1).to(torch.bfloat16)
    noise = torch.randn(B, c, hw, hw, generator=g).to(torch.bfloat16)
    t = torch.tensor([999, 500, 3, 999])
    u = torch.tensor([0.9, 0.1, 0.7, 0.2])
    ac = S.alphas_cumprod()
    gam, sig = S.gamma_sigma(ac)
    keep = S.add_noise(ac, x0, noise, t)
    forget = S.add_noise(ac, a0, noise, t)
    ref = torch.where((u > 0.5)[:, None, None, None], keep, forget)
    m = mixture_fwd(x0.to(dev), a0.to(dev), noise.to(dev), t.to(dev), u.to(dev), ac.to(dev), gam.to(dev),
                    sig.to(dev), 0.5)
    assert m.x_mix.dtype == torch.bfloat16
    torch.testing.assert_close(m.x_mix.cpu().float(), ref.float(), rtol=0, atol=0)


# ------------------------------------------------------------------ flat-buffer optimizer
def test_norms_recombine_clip_adamw_vs_torch(dev):
    from siss_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(11)
    n = 100_004   # flat buffers are padded to a multiple of 4 floats (16-B rows)
    p0 = torch.randn(n, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref_p], lr=5e-3, betas=(0.95, 0.999), weight_decay=1e-2, eps=1e-8)
    fo = FlatAdamW(p0.to(dev).clone(), lr=5e-3, betas=(0.95, 0.999), weight_decay=1e-2, eps=1e-8)
    for step in range(3):
        gx = torch.randn(n, generator=g) * 0.01
        ga = torch.randn(n, generator=g) * 0.02
        s = 5.0 / ga.norm().item()
        gref = gx - s * ga
        ref_p.grad = gref.clone()
        pre = torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        grads = torch.stack([gx, ga]).to(dev)
        st = fo.step(grads, scaling_norm=5.0, want_grad=True)
        assert abs(st["norm_loss_a"] - ga.norm().item()) < 1e-4 * ga.norm().item()
        assert abs(st["scaling_factor"] - s) < 1e-4 * s
        assert abs(st["pre_clip_norm"] - pre.item()) < 1e-4 * pre.item()
        torch.testing.assert_close(fo.last_grad.cpu(), ref_p.grad, rtol=1e-4, atol=1e-8)
        # AdamW's first steps are sign-like (dp ~ -lr * g/(|g|+eps)): an element whose |g| ~ eps at ANY step
        # amplifies the 1e-4 relative gradient difference and keeps that offset; so: >= 99.99 % of the
        # elements bit-close, and no element further than 1 % of one lr-sized update.
        err = (fo.p.cpu() - ref_p.detach()).abs()
        assert (err > 1e-6 + 1e-5 * ref_p.detach().abs()).float().mean() < 1e-4
        assert err.max() < 1e-2 * 5e-3


def test_gemm_nt_alpha_on_the_first_columns_only(dev):
    """siss_gemm_nt_alpha_cols: C[:, :alpha_cols] = alpha * A W^T (+ bias), the other columns unscaled -- the fused q / k / v projection
    whose query part leaves pre-scaled for the attention kernels.  Against an f32 matmul of the same bf16 operands; a split-K shape too."""
    from siss_amd import lib, ops
    lib.ensure_workspace("cuda:0")
    g = torch.Generator().manual_seed(3)
    for (M, N, K, ac) in ((1000, 960, 320, 320), (300, 384, 1280, 128), (4096, 320, 320, 320)):
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        c = torch.full((M, N), 5.0, dtype=torch.bfloat16, device=dev)
        ops.gemm_nt(lib.ptr(a), K, w, lib.ptr(c), N, M, N, K, [0], [0], bias=bias, alpha=0.37, alpha_cols=ac)
        torch.cuda.synchronize()
        ref = a.float() @ w.float().T
        ref[:, :ac] *= 0.37
        ref += bias
        err = (c.float() - ref).abs().max().item()
        assert err <= 1e-2 * ref.abs().max().item(), (M, N, K, ac, err)
