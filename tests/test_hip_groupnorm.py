"""GroupNorm(+SiLU) forward / backward kernels against torch fp32 on the same bf16-rounded inputs, for BOTH
implementations: the two-pass kernels (groupnorm.hip: statistics launch + apply launch) and the one-launch slab kernels
of the small sites (groupnorm_slab.hip: a block holds its (sample, channel slice) on chip).  Reference provider: torch's
GroupNorm + SiLU inside diffusers' ResnetBlock2D.norm1/norm2 and Attention.group_norm, reached from
losses/ddpm_deletion_loss.py:24 and differentiated twice at delete_celeb.py:691,:702 (here: two cotangent sets against
one saved activation).

Covered: channel counts with 4..32 channels per group incl. the non-power-of-two 384 / 768 (lanes that straddle two
groups, idle lanes), large sites (two-pass kernels in both modes), row-strided inputs (column views of a concat buffer), compact outputs / cotangents (attention), the residual
inputs (accum, accum2), the channel-split output with accumulation (concat backward), per-sample column sums
(time-embedding gradient), one and two cotangent sets, the No-IS layout (2B saved samples, sets by sample index).
Tolerances: y, dx rel 1.5e-2 of scale (bf16 outputs); dgamma / dbeta / colsum rel 5e-3; mean / rstd rel 1e-5.
The forward and dx must be bitwise deterministic.
"slab" = the default configuration: slab kernels at the small sites (<= 32 x 32), two-pass kernels elsewhere.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
G = 32


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


def _run_fwd(lib, dev, x, gamma, beta, eps, silu, compact, ld_extra, part):
    from siss_amd.layout import Act, ActView
    B, C, H, W = x.shape
    if ld_extra:
        base = Act(B, H, W, C + ld_extra, dev)
        base.buf.normal_()                                   # neighbours' columns hold junk
        xa = ActView(base, ld_extra, C)
        base.interior()[..., ld_extra:] = x.permute(0, 2, 3, 1).to(torch.bfloat16).to(dev)
        base.padded()[:, 0] = 0; base.padded()[:, -1] = 0; base.padded()[:, :, 0] = 0; base.padded()[:, :, -1] = 0
        ldx = base.c
    else:
        xa = Act.from_nchw(x, dev)
        ldx = 0
    mean, rstd = torch.empty(B, G, device=dev), torch.empty(B, G, device=dev)
    if compact:
        y = torch.full((B * H * W, C), 7.0, dtype=torch.bfloat16, device=dev)
        yp = y
    else:
        y = Act(B, H, W, C, dev)
        yp = y.data
    lib.call("siss_groupnorm_fwd_ld", xa.data, gamma.to(dev), beta.to(dev), yp, mean, rstd, part, B, H, W, C, G, eps,
             int(silu), int(compact), ldx)
    torch.cuda.synchronize()
    out = y.view(B, H, W, C).permute(0, 3, 1, 2).float() if compact else y.to_nchw()
    if not compact:
        assert y.halo_is_zero()
    return xa, ldx, out.cpu(), mean.cpu(), rstd.cpu(), (mean, rstd)


FWD_CASES = [  # B, C, H, W, silu, compact, ld_extra
    (2, 128, 16, 16, True, False, 0), (3, 256, 12, 20, True, False, 0), (2, 384, 9, 9, True, False, 0),
    (2, 512, 8, 8, False, True, 0), (1, 768, 5, 7, True, False, 0), (2, 1024, 4, 4, True, False, 0),
    (2, 128, 24, 24, True, False, 128), (5, 128, 40, 40, True, False, 0),
    (2, 128, 160, 160, True, False, 0),          # 25.6 k pixels per sample: 100 blocks per sample, two samples per round
    (1, 256, 272, 272, True, False, 0),          # 74 k pixels x 256 channels: longer than the on-chip capacity (tail re-read)
]


@pytest.mark.parametrize("mode", [0, 1], ids=["two_pass", "slab"])
@pytest.mark.parametrize("B,C,H,W,silu,compact,ldx", FWD_CASES)
def test_groupnorm_forward(dev, mode, B, C, H, W, silu, compact, ldx):
    from siss_amd import lib
    g = torch.Generator().manual_seed(C + H + B)
    x = _bf(torch.randn(B, C, H, W, generator=g) * 1.7 + 0.4)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    eps = 1e-6
    ref = F.group_norm(x, G, gamma, beta, eps)
    ref = F.silu(ref) if silu else ref
    xg = x.view(B, G, -1)
    m_ref, v_ref = xg.mean(-1), xg.var(-1, unbiased=False)
    part = torch.zeros(lib.query("siss_gn_partial_words", B, H, W, C, G), device=dev)
    assert lib.query("siss_groupnorm_set_slab", mode) == mode
    lib.dispatch_counts(reset=True)
    try:
        _, _, out, mean, rstd, _ = _run_fwd(lib, dev, x, gamma, beta, eps, silu, compact, ldx, part)
        _close(out, ref, 1.5e-2, "y")
        torch.testing.assert_close(mean, m_ref, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(rstd, (v_ref + eps).rsqrt(), rtol=1e-4, atol=1e-6)
        if mode:
            if H * W <= 1024 and C // G >= 4:
                assert lib.dispatch_counts()["gn_slab"] > 0, "a small site must run on the slab kernel"
            _, _, out2, *_ = _run_fwd(lib, dev, x, gamma, beta, eps, silu, compact, ldx, part)
            assert torch.equal(out, out2), "forward must be bitwise deterministic"
    finally:
        lib.query("siss_groupnorm_set_slab", -1)


BWD_CASES = [  # B(saved), sets, C, H, W, silu, compact_dy, accum, accum2, split, colsum, ld_extra
    (2, 2, 128, 16, 16, True, False, False, False, 0, True, 0),
    (2, 2, 256, 12, 20, True, False, True, False, 0, False, 0),
    (2, 2, 384, 9, 9, True, False, True, True, 256, False, 0),
    (2, 2, 512, 8, 8, False, True, True, False, 0, False, 0),
    (1, 2, 768, 5, 7, True, False, False, False, 512, True, 0),
    (2, 1, 1024, 4, 4, True, False, False, False, 0, False, 0),
    (4, 1, 128, 24, 24, True, False, True, False, 0, True, 128),       # No-IS layout: 4 saved samples, sets by sample index
    (3, 2, 128, 40, 40, True, False, False, False, 0, False, 0),
    (2, 2, 128, 160, 160, True, False, True, False, 0, True, 0),
    (1, 2, 256, 272, 272, True, False, False, False, 128, False, 0),   # tail re-read + split with accumulation
]


@pytest.mark.parametrize("mode", [0, 1], ids=["two_pass", "slab"])
@pytest.mark.parametrize("B,sets,C,H,W,silu,cdy,acc,acc2,split,colsum,ldx", BWD_CASES)
def test_groupnorm_backward(dev, mode, B, sets, C, H, W, silu, cdy, acc, acc2, split, colsum, ldx):
    from siss_amd import lib
    from siss_amd.layout import Act
    g = torch.Generator().manual_seed(C + H + B + sets)
    x = _bf(torch.randn(B, C, H, W, generator=g) * 1.7 + 0.4)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    eps = 1e-6
    n2 = sets * B
    dy = _bf(torch.randn(n2, C, H, W, generator=g))
    r1 = _bf(torch.randn(n2, C, H, W, generator=g)) if acc else None
    r2 = _bf(torch.randn(n2, C, H, W, generator=g)) if acc2 else None
    r3 = _bf(torch.randn(n2, C - split, H, W, generator=g)) if split else None     # running cotangent of the concat's tail part
    nsets = 2
    set_images = n2 // nsets
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.group_norm(xr, G, gr, br, eps)
    y = F.silu(y) if silu else y
    dx_ref = torch.cat([torch.autograd.grad(y, xr, dy[k * B:(k + 1) * B], retain_graph=True)[0] for k in range(sets)])
    # parameter gradients per SET (set = cotangent sample index // set_images)
    dg_ref, db_ref = torch.zeros(nsets, C), torch.zeros(nsets, C)
    for i in range(n2):
        mask = torch.zeros(B, C, H, W)
        mask[i % B] = dy[i]
        gg, gb = torch.autograd.grad(y, (gr, br), mask, retain_graph=True)
        dg_ref[i // set_images] += gg; db_ref[i // set_images] += gb
    col_ref = dx_ref.sum(dim=(2, 3))
    tot_ref = dx_ref + (r1 if acc else 0) + (r2 if acc2 else 0)
    if split:
        tot_ref = torch.cat([tot_ref[:, :split], tot_ref[:, split:] + r3], 1)

    part = torch.zeros(lib.query("siss_gn_partial_words", n2, H, W, C, G), device=dev)
    assert lib.query("siss_groupnorm_set_slab", mode) == mode
    try:
        xa, ldxv, _, _, _, (mean, rstd) = _run_fwd(lib, dev, x, gamma, beta, eps, silu, False, ldx, part)
        if cdy:
            dyp = dy.permute(0, 2, 3, 1).reshape(n2 * H * W, C).to(torch.bfloat16).to(dev).contiguous()
        else:
            dya = Act.from_nchw(dy, dev); dyp = dya.data
        a1 = Act.from_nchw(r1, dev) if acc else None
        a2 = Act.from_nchw(r2, dev) if acc2 else None
        P = 4096
        grads = torch.zeros(nsets, P, device=dev)
        cs = torch.zeros(n2, C + 8, device=dev) if colsum else None
        runs = []
        for rep in range(2 if mode else 1):
            grads.zero_()
            if colsum:
                cs.zero_()
            if split:
                da, db = Act(n2, H, W, split, dev), Act.from_nchw(r3, dev)
                dxp, dx2p = da.data, db.data
            else:
                dxa = Act(n2, H, W, C, dev); dxa.buf.fill_(3.0); dxa.buf[:dxa.guard * C] = 0; dxa.buf[-dxa.guard * C:] = 0
                dxa.padded()[:, 0] = 0; dxa.padded()[:, -1] = 0; dxa.padded()[:, :, 0] = 0; dxa.padded()[:, :, -1] = 0
                dxp, dx2p = dxa.data, None
            lib.call("siss_groupnorm_bwd_ld", dyp, xa.data, gamma.to(dev), beta.to(dev), mean, rstd, dxp,
                     a1.data if acc else None, a2.data if acc2 else None, dx2p, split, int(bool(split)),
                     grads[0, 64:], grads[0, 2048:], cs, C + 8, part, n2, B, set_images, P, H, W, C, G, int(silu), int(cdy), ldxv)
            torch.cuda.synchronize()
            got = torch.cat([da.to_nchw(), db.to_nchw()], 1).cpu() if split else dxa.to_nchw().cpu()
            runs.append(got)
        _close(got, tot_ref, 1.5e-2, "dx")
        for k in range(nsets):
            _close(grads[k, 64:64 + C].cpu(), dg_ref[k], 5e-3, f"dgamma set {k}")
            _close(grads[k, 2048:2048 + C].cpu(), db_ref[k], 5e-3, f"dbeta set {k}")
        if colsum:
            _close(cs[:, :C].cpu(), col_ref, 5e-3, "colsum")
        if mode:
            assert torch.equal(runs[0], runs[1]), "dx must be bitwise deterministic (fixed-order folds)"
    finally:
        lib.query("siss_groupnorm_set_slab", -1)


@pytest.mark.parametrize("B,C,H,W,split,acc", [(2, 256, 64, 64, 128, True), (1, 384, 40, 48, 256, False), (2, 96, 24, 36, 64, True)])
def test_groupnorm_backward_space_to_depth_target_equals_the_layout_pass(dev, B, C, H, W, split, acc):
    """siss_groupnorm_bwd_ld_s2d writes the FIRST part of a split target in space-to-depth layout (what a sub-pixel upsample
    convolution's backward consumes): bit for bit the plain split output pushed through siss_space_to_depth; the second part, the
    parameter gradients and the column sums are those of the plain launch."""
    from siss_amd import lib
    from siss_amd.layout import Act
    g = torch.Generator().manual_seed(C + H + W)
    x = _bf(torch.randn(B, C, H, W, generator=g) * 1.3 + 0.2)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    n2, nsets, eps, silu = 2 * B, 2, 1e-6, True
    dy = _bf(torch.randn(n2, C, H, W, generator=g))
    r1 = _bf(torch.randn(n2, C, H, W, generator=g)) if acc else None
    r3 = _bf(torch.randn(n2, C - split, H, W, generator=g))
    part = torch.zeros(lib.query("siss_gn_partial_words", n2, H, W, C, G), device=dev)
    xa, ldxv, _, _, _, (mean, rstd) = _run_fwd(lib, dev, x, gamma, beta, eps, silu, False, 0, part)
    dya = Act.from_nchw(dy, dev)
    a1 = Act.from_nchw(r1, dev) if acc else None
    P = 4096
    outs = []
    for entry in ("siss_groupnorm_bwd_ld", "siss_groupnorm_bwd_ld_s2d"):
        grads = torch.zeros(nsets, P, device=dev)
        cs = torch.zeros(n2, C + 8, device=dev)
        s2d = entry.endswith("_s2d")
        da = Act(n2, H // 2, W // 2, 4 * split, dev) if s2d else Act(n2, H, W, split, dev)
        db = Act.from_nchw(r3, dev)
        lib.call(entry, dya.data, xa.data, gamma.to(dev), beta.to(dev), mean, rstd, da.data, a1.data if acc else None, None, db.data,
                 split, 1, grads[0, 64:], grads[0, 2048:], cs, C + 8, part, n2, B, n2 // nsets, P, H, W, C, G, int(silu), 0, ldxv)
        torch.cuda.synchronize()
        if not s2d:
            z = Act(n2, H // 2, W // 2, 4 * split, dev)
            lib.call("siss_space_to_depth", da.data, z.data, n2, H, W, split)
            torch.cuda.synchronize()
            da = z
        assert da.halo_is_zero()
        outs.append((da.buf.clone(), db.buf.clone(), cs.clone(), grads.clone()))
    (z0, b0, c0, g0), (z1, b1, c1, g1) = outs
    assert torch.equal(z0, z1), "space-to-depth first part"
    assert torch.equal(b0, b1), "second part"
    assert torch.allclose(c0, c1, rtol=1e-5, atol=1e-5) and torch.allclose(g0, g1, rtol=1e-4, atol=1e-4)   # (float atomics: order-dependent)


@pytest.mark.parametrize("mode", [0, 1], ids=["two_pass", "slab"])
def test_groupnorm_backward_extreme_negative_preactivation_stays_finite(dev, mode):
    """A pre-activation below about -88.7 makes exp(-z) overflow in f32: SiLU'(z) must come out as 0 there (torch's value), not as
    inf * 0 = NaN -- one such element would poison the statistics of its (sample, group), hence dx of the whole group, dgamma / dbeta
    and the gradient norm.  Channel 3 sits at beta = -150, channel 5 at gamma = 90 (both signs of a huge |z|)."""
    from siss_amd import lib
    from siss_amd.layout import Act
    B, C, H, W, sets = 2, 128, 40, 40, 2
    g = torch.Generator().manual_seed(11)
    x = _bf(torch.randn(B, C, H, W, generator=g) * 1.7 + 0.4)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    beta[3], gamma[5] = -150.0, 90.0
    eps, n2 = 1e-6, sets * B
    dy = _bf(torch.randn(n2, C, H, W, generator=g))
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.silu(F.group_norm(xr, G, gr, br, eps))
    assert (F.group_norm(x, G, gamma, beta, eps) < -90).any(), "the case must reach the overflow range"
    dx_ref = torch.cat([torch.autograd.grad(y, xr, dy[k * B:(k + 1) * B], retain_graph=True)[0] for k in range(sets)])
    dg_ref = torch.stack([torch.autograd.grad(y, gr, dy[k * B:(k + 1) * B], retain_graph=True)[0] for k in range(sets)])
    db_ref = torch.stack([torch.autograd.grad(y, br, dy[k * B:(k + 1) * B], retain_graph=True)[0] for k in range(sets)])
    part = torch.zeros(lib.query("siss_gn_partial_words", n2, H, W, C, G), device=dev)
    assert lib.query("siss_groupnorm_set_slab", mode) == mode
    try:
        xa, ldxv, _, _, _, (mean, rstd) = _run_fwd(lib, dev, x, gamma, beta, eps, True, False, 0, part)
        dya, dxa = Act.from_nchw(dy, dev), Act(n2, H, W, C, dev)
        P = 4096
        grads = torch.zeros(2, P, device=dev)
        lib.call("siss_groupnorm_bwd_ld", dya.data, xa.data, gamma.to(dev), beta.to(dev), mean, rstd, dxa.data, None, None, None, 0, 0,
                 grads[0, 64:], grads[0, 2048:], None, 0, part, n2, B, B, P, H, W, C, G, 1, 0, ldxv)
        torch.cuda.synchronize()
        got = dxa.to_nchw().cpu()
        assert torch.isfinite(got).all() and torch.isfinite(grads).all(), "NaN / inf from an overflowing exp(-z)"
        _close(got, dx_ref, 1.5e-2, "dx")
        for k in range(2):
            _close(grads[k, 64:64 + C].cpu(), dg_ref[k], 5e-3, f"dgamma set {k}")
            _close(grads[k, 2048:2048 + C].cpu(), db_ref[k], 5e-3, f"dbeta set {k}")
    finally:
        lib.query("siss_groupnorm_set_slab", -1)
