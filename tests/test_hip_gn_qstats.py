"""GroupNorm statistics from the producing convolution's epilogue (NTParams::qstats / siss_groupnorm_fwd_qs).

The persistent 3x3 kernel's store waves can leave, per (128-row half of a 254-row tile, image slot, 4-channel quad), the
sum and sum of squares of the bf16 values they store; the GroupNorm that consumes the tensor then folds those entries and
skips its own statistics pass (one full read of the tensor).  Checked here:

* the entries themselves, folded on the host with the documented geometry, against sums over the stored tensor -- tiles
  that straddle two images (slot 1), the partial last tile, residual + row-bias epilogues, several column tiles;
* siss_groupnorm_fwd_qs against siss_groupnorm_fwd_ld on the same tensor: mean / rstd rel 1e-5 (both are double sums of
  f32 partials, grouped differently), y identical up to one bf16 rounding on a handful of elements;
* a channel concat of two producers whose group boundary falls INSIDE neither part (128 + 128, 32 groups of 8) and one
  whose groups straddle the seam (256 + 128, 32 groups of 12);
* the UNet engine with and without the hand-over (SISS_GN_EPI_STATS): same prediction, same gradients.

Reference provider: GroupNorm in diffusers' ResnetBlock2D (norm1 / norm2), reached from losses/ddpm_deletion_loss.py:24.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

TILE, HALF = 254, 128          # common.h kQsTileRows / kQsHalfRows


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from siss_amd import lib
    lib.load()
    lib.ensure_workspace("cuda:0")
    return torch.device("cuda:0")


def _bf(x):
    return x.to(torch.bfloat16).float()


def _conv_with_stats(dev, n, h, w, ci, co, seed, residual=True, rowbias=True):
    """Run one 3x3 conv through siss_gemm_nt_qstats; returns (out Act, qstats tensor)."""
    from siss_amd import lib, ops
    from siss_amd.layout import Act
    g = torch.Generator().manual_seed(seed)
    x = _bf(torch.randn(n, ci, h, w, generator=g))
    wt = _bf(torch.randn(co, ci, 3, 3, generator=g) * (1.0 / (3 * ci ** 0.5)))
    bias = torch.randn(co, generator=g) * 0.5 + 0.3        # a non-zero mean makes E[x^2] - E[x]^2 do some work
    temb = torch.randn(n, co, generator=g)
    res = _bf(torch.randn(n, co, h, w, generator=g))
    xa = Act.from_nchw(x, dev)
    out = Act(n, h, w, co, dev)
    words = lib.query("siss_conv_qstats_words", xa.rows, co)
    assert words == -(-xa.rows // TILE) * 2 * 2 * (co // 4) * 2
    qs = torch.full((words,), float("nan"), device=dev)    # every entry must be written by the launch
    lib.dispatch_counts(reset=True)
    ok = ops.conv_fprop_qstats(xa, ops.conv_w_to_native(wt).to(dev).to(torch.bfloat16), out, qs, bias=bias.to(dev),
                               rowbias=temb.to(dev) if rowbias else None,
                               residual=Act.from_nchw(res, dev) if residual else None)
    torch.cuda.synchronize()
    assert ok and lib.dispatch_counts(reset=True)["gemm_nt_c3p_kernel"] == 1
    return out, qs


def _fold_on_host(qs, n, h, w, co):
    """[n][co/4][2] from the entry layout documented in common.h / nt_common.h."""
    rpi = (h + 2) * (w + 2)
    q = qs.double().cpu().view(-1, 2, co // 4, 2)           # [half tile][slot][quad][sum, sumsq]
    assert not torch.isnan(q).any()
    out = torch.zeros(n, co // 4, 2, dtype=torch.float64)
    for i in range(n):
        t0, t1 = i * rpi // TILE, ((i + 1) * rpi - 1) // TILE
        for t in range(t0, t1 + 1):
            slot = 0 if t * TILE // rpi == i else 1
            out[i] += q[2 * t, slot] + q[2 * t + 1, slot]
    return out


@pytest.mark.parametrize("n,h,w,ci,co,residual", [
    (2, 128, 128, 128, 128, True),       # tile 66 straddles the two images
    (4, 64, 64, 256, 256, True),         # two column tiles, three image seams, partial last tile
    (5, 96, 96, 128, 128, False),        # no residual: the counted-vmcnt path of the store waves (10 stores in flight)
])
def test_entries_match_the_stored_tensor(dev, n, h, w, ci, co, residual):
    out, qs = _conv_with_stats(dev, n, h, w, ci, co, seed=n + h + co, residual=residual)
    got = _fold_on_host(qs, n, h, w, co)
    y = out.to_nchw().double().cpu()                        # interior pixels; the halo is zero and adds nothing
    ref_s = y.view(n, co // 4, 4, -1).sum(dim=(2, 3))
    ref_q = (y * y).view(n, co // 4, 4, -1).sum(dim=(2, 3))
    scale = ref_q.abs().max().item()
    assert (got[..., 0] - ref_s).abs().max().item() <= 2e-5 * max(scale, ref_s.abs().max().item())
    assert (got[..., 1] - ref_q).abs().max().item() <= 2e-5 * scale
    assert out.halo_is_zero()


def _gn_both_ways(dev, x_act, c, G, qa, ca, qb, silu=True):
    from siss_amd import lib
    from siss_amd.layout import Act
    n, h, w = x_act.n, x_act.h, x_act.w
    g = torch.Generator().manual_seed(c + G)
    gamma = (torch.rand(c, generator=g) + 0.5).to(dev)
    beta = torch.randn(c, generator=g).to(dev)
    part = torch.zeros(lib.query("siss_gn_partial_words", n, h, w, c, G), device=dev)
    res = []
    for use_q in (False, True):
        y = Act(n, h, w, c, dev)
        mean, rstd = torch.zeros(n, G, device=dev), torch.zeros(n, G, device=dev)
        lib.dispatch_counts(reset=True)
        if use_q:
            lib.call("siss_groupnorm_fwd_qs", x_act.data, gamma, beta, y.data, mean, rstd, part, qa, ca, qb,
                     n, h, w, c, G, 1e-6, int(silu), 0, 0)
        else:
            lib.call("siss_groupnorm_fwd_ld", x_act.data, gamma, beta, y.data, mean, rstd, part, n, h, w, c, G, 1e-6, int(silu), 0, 0)
        torch.cuda.synchronize()
        assert lib.dispatch_counts(reset=True)["gn_qstats"] == int(use_q)
        res.append((y.to_nchw().cpu(), mean.cpu(), rstd.cpu()))
    (y0, m0, r0), (y1, m1, r1) = res
    assert (m0 - m1).abs().max().item() <= 1e-5 * (m0.abs().max().item() + 1.0)
    assert ((r0 - r1).abs() / r0).max().item() <= 1e-5
    d = (y0 - y1).abs()
    assert d.max().item() <= 2.0 ** -7 * (y0.abs().max().item() + 1e-6)     # at most one bf16 step on the largest value
    assert (d > 0).float().mean().item() < 1e-3                                # ... and only where a rounding tie flipped


def test_groupnorm_on_the_conv_statistics_equals_the_two_pass_form(dev):
    out, qs = _conv_with_stats(dev, 2, 128, 128, 128, 128, seed=7)
    _gn_both_ways(dev, out, 128, 32, qs, 128, None)
    out, qs = _conv_with_stats(dev, 4, 64, 64, 256, 256, seed=8)
    _gn_both_ways(dev, out, 256, 32, qs, 256, None, silu=False)


@pytest.mark.parametrize("ca,cb", [(128, 128), (256, 128)])
def test_groupnorm_over_a_concat_of_two_producers(dev, ca, cb):
    """Both parts written straight into the concat buffer (epilogue ldc = ca + cb), each with its own statistics."""
    from siss_amd import lib, ops
    from siss_amd.layout import Act, ActView
    n, h, w = 8, 64, 64                                     # 34,848 padded rows: >= 256 128-row tiles even at 128 channels
    cat = Act(n, h, w, ca + cb, dev)
    g = torch.Generator().manual_seed(ca + cb)
    qss = []
    for c0, c in ((0, ca), (ca, cb)):
        x = Act.from_nchw(_bf(torch.randn(n, 128, h, w, generator=g)), dev)
        wt = _bf(torch.randn(c, 128, 3, 3, generator=g) * 0.03)
        qs = torch.full((lib.query("siss_conv_qstats_words", x.rows, c),), float("nan"), device=dev)
        view = ActView(cat, c0, c)
        ok = ops.conv_fprop_qstats(x, ops.conv_w_to_native(wt).to(dev).to(torch.bfloat16), view, qs,
                                   bias=(torch.randn(c, generator=g) + 0.2).to(dev))
        assert ok
        qss.append(qs)
    torch.cuda.synchronize()
    assert cat.halo_is_zero()
    _gn_both_ways(dev, cat, ca + cb, 32, qss[0], ca, qss[1])


def test_engine_with_and_without_the_hand_over(dev):
    """Full-width CelebA-HQ UNet at 64 x 64 (the 64 x 64 level's convs run on the persistent kernel): forward + dual backward
    with the statistics hand-over equal the two-pass GroupNorm run."""
    from siss_amd import lib
    from siss_amd.config import UNet2DConfig
    from siss_amd.unet import UNetEngine
    cfg = UNet2DConfig(sample_size=64, block_out_channels=(128, 256, 256), down_block_types=("DownBlock2D",) * 3,
                       up_block_types=("UpBlock2D",) * 3)
    x = torch.randn(8, 3, 64, 64, generator=torch.Generator().manual_seed(0)).to(dev)
    t = torch.full((8,), 999, dtype=torch.int64, device=dev)
    cot = torch.randn(16, 3, 64, 64, generator=torch.Generator().manual_seed(1)).to(dev)
    runs = []
    for on in (False, True):
        eng = UNetEngine(cfg, "cuda:0")
        eng.init_random(seed=3)
        eng.epi_stats = on
        lib.dispatch_counts(reset=True)
        pred = eng.forward(x, t).clone()
        torch.cuda.synchronize()
        cnt = lib.dispatch_counts(reset=True)
        assert (cnt["gn_qstats"] > 0) == on, cnt
        eng.zero_grad()
        eng.backward(cot, nsets=2)
        torch.cuda.synchronize()
        runs.append((pred.cpu(), eng.ps.grads.clone().cpu()))
    (p0, g0), (p1, g1) = runs
    assert (p0 - p1).abs().max().item() <= 2e-2 * p0.abs().max().item()
    for s in range(2):
        cos = torch.nn.functional.cosine_similarity(g0[s], g1[s], dim=0).item()
        assert cos > 0.9999, cos
        assert abs(g0[s].norm().item() / g1[s].norm().item() - 1) < 1e-3


# ---------------------------------------------------------------------------------------------------------------------
# siss_quad_stats: the same entries for a tensor that no persistent-conv epilogue produced (conv_in, downsample, sub-pixel upsample)
@pytest.mark.parametrize("n,h,w,c,ld", [
    (2, 128, 128, 128, 128),         # a tile straddles the two images
    (5, 40, 40, 256, 384),           # a column view of a concat buffer (ld > c), four image seams, partial last tile
    (3, 64, 64, 384, 384),           # 48 lanes per row: 5 rows in flight, 16 idle lanes
    (16, 34, 34, 512, 512),          # the widest part the engine hands over
])
def test_quad_stats_entries_match_the_tensor(dev, n, h, w, c, ld):
    from siss_amd import lib
    from siss_amd.layout import Act, ActView
    g = torch.Generator().manual_seed(n + h + c)
    x = _bf(torch.randn(n, c, h, w, generator=g) + 0.3)
    if ld == c:
        a = Act.from_nchw(x, dev)
    else:
        base = Act.from_nchw(_bf(torch.randn(n, ld, h, w, generator=g)), dev)
        a = ActView(base, ld - c, c)
        a.data.copy_(Act.from_nchw(x, dev).data)
    qs = torch.full((lib.query("siss_conv_qstats_words", a.rows, c),), float("nan"), device=dev)
    lib.call("siss_quad_stats", a.data, ld, a.rows, c, a.rows_per_image, qs)
    torch.cuda.synchronize()
    got = _fold_on_host(qs, n, h, w, c)
    y = x.double()
    ref_s = y.view(n, c // 4, 4, -1).sum(dim=(2, 3))
    ref_q = (y * y).view(n, c // 4, 4, -1).sum(dim=(2, 3))
    scale = ref_q.abs().max().item()
    assert (got[..., 0] - ref_s).abs().max().item() <= 2e-5 * max(scale, ref_s.abs().max().item())
    assert (got[..., 1] - ref_q).abs().max().item() <= 2e-5 * scale


def test_groupnorm_on_formed_statistics_of_one_half_and_conv_statistics_of_the_other(dev):
    """concat(a sub-pixel upsample's output: siss_quad_stats, a resnet's output: the conv's epilogue) -> the two-pass form's result."""
    from siss_amd import lib, ops
    from siss_amd.layout import Act, ActView
    n, h, w, ca, cb = 8, 64, 64, 128, 128
    cat = Act(n, h, w, ca + cb, dev)
    g = torch.Generator().manual_seed(77)
    va, vb = ActView(cat, 0, ca), ActView(cat, ca, cb)
    va.data.copy_(Act.from_nchw(_bf(torch.randn(n, ca, h, w, generator=g) * 1.5 - 0.2), dev).data)
    qa = torch.full((lib.query("siss_conv_qstats_words", va.rows, ca),), float("nan"), device=dev)
    lib.call("siss_quad_stats", va.data, ca + cb, va.rows, ca, va.rows_per_image, qa)
    x = Act.from_nchw(_bf(torch.randn(n, 128, h, w, generator=g)), dev)
    wt = _bf(torch.randn(cb, 128, 3, 3, generator=g) * 0.03)
    qb = torch.full((lib.query("siss_conv_qstats_words", x.rows, cb),), float("nan"), device=dev)
    assert ops.conv_fprop_qstats(x, ops.conv_w_to_native(wt).to(dev).to(torch.bfloat16), vb, qb, bias=(torch.randn(cb, generator=g) + 0.2).to(dev))
    torch.cuda.synchronize()
    assert cat.halo_is_zero()
    _gn_both_ways(dev, cat, ca + cb, 32, qa, ca, qb)


def test_engine_forms_statistics_once_per_skip_tensor(dev):
    """CelebA-HQ-shaped UNet at 128 x 128: with UNetEngine.quad_stats the two-pass GroupNorm sites whose input (or one half of it) no
    persistent-conv epilogue produced run on formed statistics -- no statistics pass of their own --, a skip tensor's entries are
    formed once and used twice, and the prediction / gradients equal the run without."""
    from siss_amd import lib
    from siss_amd.config import UNet2DConfig
    from siss_amd.unet import UNetEngine
    cfg = UNet2DConfig(sample_size=128, block_out_channels=(128, 128, 256), down_block_types=("DownBlock2D",) * 3,
                       up_block_types=("UpBlock2D",) * 3)
    x = torch.randn(4, 3, 128, 128, generator=torch.Generator().manual_seed(0)).to(dev)
    t = torch.full((4,), 999, dtype=torch.int64, device=dev)
    cot = torch.randn(8, 3, 128, 128, generator=torch.Generator().manual_seed(1)).to(dev)
    runs = []
    for on in (False, True):
        eng = UNetEngine(cfg, "cuda:0")
        eng.init_random(seed=3)
        eng.quad_stats = on
        calls = []
        orig = lib.call
        lib.call = lambda name, *a, _o=orig, _c=calls, **k: (_c.append(name), _o(name, *a, **k))[1]
        try:
            for _ in range(2):                              # twice: the entries of the first pass must not serve the second
                pred = eng.forward(x if _ == 1 else x * 0.5, t).clone()
            eng.zero_grad()
            eng.backward(cot, nsets=2)
            torch.cuda.synchronize()
        finally:
            lib.call = orig
        runs.append((pred.cpu(), eng.ps.grads.clone().cpu(), calls))
    (p0, g0, c0), (p1, g1, c1) = runs
    assert c0.count("siss_quad_stats") == 0 and c1.count("siss_quad_stats") > 0
    assert c1.count("siss_groupnorm_fwd_qs") > c0.count("siss_groupnorm_fwd_qs")        # sites that left the statistics pass
    two_pass = lambda c: c.count("siss_groupnorm_fwd_ld")
    assert two_pass(c1) < two_pass(c0)
    assert (p0 - p1).abs().max().item() <= 2e-2 * p0.abs().max().item()
    for s in range(2):
        assert torch.nn.functional.cosine_similarity(g0[s], g1[s], dim=0).item() > 0.9999
