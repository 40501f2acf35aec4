"""Element-wise parity of the kernels that CARRY the step, at shapes that really dispatch to them.

The small cases of test_hip_kernels.py all land on the generic `gemm_nt_kernel` / `gemm_tn_kernel<1>`.  The CelebA-HQ step
spends 31 % of its time in `gemm_nt_c3p_kernel` (persistent 3x3 kernel: grids of >= 256 128-row tiles, >= 256 padded rows
per image, N % 128 == 0) and 22 % in `gemm_tn_kernel<3>` (fused three-tap wgrad, >= 8192 reduction rows).  Every case
below is sized to be eligible for them, and the library's own dispatch counters (siss_dispatch_count) are asserted, so a
change of the dispatch thresholds cannot silently turn these back into small-kernel tests.

What the shapes stress (gemm_nt_c3p.hip): 254-row tiles whose seams fall anywhere in an image row; tiles that span TWO
images (per-image row bias select + halo mask of both, `park(two_images)` / `store_tile`); several tile rounds per
block with an uneven last round (balanced persistent grid); a narrowed grid (siss_gemm_nt_set_c3p_blocks) as the
data-parallel autotune uses it.

Reference = torch fp32 conv2d on the CPU over the SAME bf16-rounded operands (reference provider: the cuDNN convs behind
diffusers' ResnetBlock2D, reached from losses/ddpm_deletion_loss.py:24 and differentiated at delete_celeb.py:691,:702).
Tolerances: bf16 outputs rel 1e-2 of the tensor's scale (a wrong / dropped / mis-masked row is an O(0.2) error);
f32 wgrad rel 2e-3 of scale (a dropped 254-row tile of a 65 k-row reduction is > 1e-2).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from siss_amd import lib
    lib.load()
    lib.ensure_workspace("cuda:0")
    return torch.device("cuda:0")


def _bf(x):
    return x.to(torch.bfloat16).float()


def _close(got, ref, rel, what=""):
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g} (rel {err / scale:.3g} > {rel})"


# (n, h, w, cin, cout, nsets, c3p_blocks): padded rows M = n (h+2)(w+2); 254-row c3p tiles = ceil(M / 254) * cout / 128
CASES = [
    (2, 128, 128, 128, 128, 1, 0),      # 134 c3p tiles in one round; tile 66/67 crosses the image boundary
    (1, 256, 256, 256, 128, 2, 0),      # 263 tiles: two rounds on a balanced 136-block grid; residual + row bias; K = 256
    (4, 64, 64, 256, 256, 1, 0),        # 69 x 2 tiles (two column tiles share an A tile), three image seams
    (5, 160, 160, 128, 128, 1, 0),      # 517 tiles: three rounds, uneven last round, 5 images with row bias
    (2, 128, 128, 128, 128, 1, 248),    # the same product on a grid held to 248 CUs (data-parallel overlap knob)
    (8, 64, 64, 64, 128, 2, 0),         # K = 64: ONE k-chunk per filter row, three groups per tile (the store waves' two halves and the
                                        # next tile's pull meet); dgrad 128 -> 64 takes the generic kernel (N = 64); the wgrad tile
                                        # has 64 of its 128 columns (zero-page lanes: the general staging form, not the fast one)
]


@pytest.mark.parametrize("n,h,w,ci,co,nsets,blocks", CASES)
def test_conv3x3_on_the_persistent_and_fused_wgrad_kernels(dev, n, h, w, ci, co, nsets, blocks):
    from siss_amd import lib, ops
    from siss_amd.layout import Act
    g = torch.Generator().manual_seed(n * 1000 + h + ci + blocks)
    x = _bf(torch.randn(n, ci, h, w, generator=g))
    wt = _bf(torch.randn(co, ci, 3, 3, generator=g) * (1.0 / (3 * ci ** 0.5)))
    bias = torch.randn(co, generator=g)
    temb = torch.randn(n, co, generator=g)                 # per-image row bias: DIFFERENT per image on purpose
    res = _bf(torch.randn(n, co, h, w, generator=g))
    dy = _bf(torch.randn(nsets * n, co, h, w, generator=g))
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wt, bias, padding=1) + temb[:, :, None, None] + res
    # dgrad runs at batch nsets * n (the dual-cotangent backward), wgrad per set against the SAME saved activation
    dx_ref = torch.cat([torch.autograd.grad(F.conv2d(xr, wt, padding=1), xr, dy[s * n:(s + 1) * n])[0] for s in range(nsets)])
    dw_ref = []
    for s in range(nsets):
        wr = wt.clone().requires_grad_(True)
        dw_ref.append(torch.autograd.grad(F.conv2d(x, wr, padding=1), wr, dy[s * n:(s + 1) * n])[0])

    assert lib.query("siss_gemm_nt_set_c3p_blocks", blocks) == (blocks or 256)
    try:
        lib.dispatch_counts(reset=True)
        xa, ra = Act.from_nchw(x, dev), Act.from_nchw(res, dev)
        wn = ops.conv_w_to_native(wt).to(dev)
        out = Act(n, h, w, co, dev)
        out.buf.fill_(7.0)                                  # poison: the kernel must write the zero halo itself
        out.buf[: out.guard * co] = 0
        out.buf[-out.guard * co:] = 0
        ops.conv_fprop(xa, wn.to(torch.bfloat16), out, bias=bias.to(dev), rowbias=temb.to(dev), residual=ra)
        torch.cuda.synchronize()
        cnt = lib.dispatch_counts(reset=True)
        assert cnt["gemm_nt_c3p_kernel"] == 1 and cnt["gemm_nt_kernel"] == 0, cnt
        assert out.halo_is_zero()
        _close(out.to_nchw().cpu(), y_ref.detach(), 1e-2, "fprop")

        dya = Act.from_nchw(dy, dev)
        dx = Act(nsets * n, h, w, ci, dev)
        dx.buf.fill_(-3.0)
        dx.buf[: dx.guard * ci] = 0
        dx.buf[-dx.guard * ci:] = 0
        ops.conv_dgrad(dya, ops.dgrad_weight(wn), dx)
        torch.cuda.synchronize()
        cnt = lib.dispatch_counts(reset=True)
        if ci % 128 == 0:
            assert cnt["gemm_nt_c3p_kernel"] == 1 and cnt["gemm_nt_kernel"] == 0, cnt
        else:                                               # the persistent kernel needs whole 128-column tiles
            assert cnt["gemm_nt_c3p_kernel"] == 0 and cnt["gemm_nt_kernel"] == 1, cnt
        assert dx.halo_is_zero()
        _close(dx.to_nchw().cpu(), dx_ref, 1e-2, "dgrad")
    finally:
        lib.query("siss_gemm_nt_set_c3p_blocks", 0)

    dW = torch.zeros(nsets, 9, co, ci, device=dev)
    dB = torch.zeros(nsets, co, device=dev)
    lib.dispatch_counts(reset=True)
    ops.conv_wgrad(dya, xa, dW, nsets=nsets, dbias=dB) if nsets == 1 else _wgrad_sets(ops, lib, dya, xa, dW, dB, nsets)
    torch.cuda.synchronize()
    cnt = lib.dispatch_counts(reset=True)
    assert cnt["gemm_tn_kernel<3>"] == 1 and cnt["gemm_tn_kernel<1>"] == 0, cnt
    for s in range(nsets):
        _close(ops.conv_w_from_native(dW[s]).cpu(), dw_ref[s], 2e-3, f"wgrad set {s}")
        _close(dB[s].cpu(), dy[s * n:(s + 1) * n].sum(dim=(0, 2, 3)), 2e-3, f"bias grad set {s}")


# 128 x 160 tiles of the generic NT kernel (round 6): widths that are multiples of 160 but not of 128 on grids of more than 512 tiles
@pytest.mark.parametrize("n,h,w,ci,co", [(8, 64, 64, 320, 320), (2, 96, 96, 64, 960), (6, 72, 72, 320, 640)])
def test_conv3x3_and_linear_on_the_160_wide_tiles(dev, n, h, w, ci, co):
    """SD v1.5's 320-wide level: a 3x3 convolution (bias, per-image row bias, residual, halo mask; its dgrad) and a linear layer (bias +
    residual, ragged last row tile) whose output width is 320 / 960 land on gemm_nt_kernel<128, 4, 1, 160> -- the dispatch counter says
    so -- against torch fp32 on the same bf16 operands; a 640-wide product of the same grid keeps the 128-wide tile."""
    from siss_amd import lib, ops
    from siss_amd.layout import Act
    g = torch.Generator().manual_seed(n * 100 + h + ci + co)
    wide = co % 160 == 0 and co % 128 != 0
    x = _bf(torch.randn(n, ci, h, w, generator=g))
    wt = _bf(torch.randn(co, ci, 3, 3, generator=g) * (1.0 / (3 * ci ** 0.5)))
    bias, temb = torch.randn(co, generator=g), torch.randn(n, co, generator=g)
    res = _bf(torch.randn(n, co, h, w, generator=g))
    y_ref = F.conv2d(x, wt, bias, padding=1) + temb[:, :, None, None] + res
    xa, ra = Act.from_nchw(x, dev), Act.from_nchw(res, dev)
    wn = ops.conv_w_to_native(wt).to(dev)
    out = Act(n, h, w, co, dev)
    out.buf.fill_(7.0)
    out.buf[: out.guard * co] = 0
    out.buf[-out.guard * co:] = 0
    lib.dispatch_counts(reset=True)
    ops.conv_fprop(xa, wn.to(torch.bfloat16), out, bias=bias.to(dev), rowbias=temb.to(dev), residual=ra)
    torch.cuda.synchronize()
    cnt = lib.dispatch_counts(reset=True)
    # (a width of whole 128-column tiles goes to the persistent 3x3 kernel)
    assert cnt["gemm_nt_kernel"] + cnt["gemm_nt_c3p_kernel"] == 1 and cnt["gemm_nt_kernel/wide"] == (1 if wide else 0), cnt
    assert out.halo_is_zero()
    _close(out.to_nchw().cpu(), y_ref, 1e-2, "fprop")
    if ci % 160 == 0 and ci % 128 != 0:                       # the dgrad's width is the input's
        dy = _bf(torch.randn(n, co, h, w, generator=g))
        xr = x.clone().requires_grad_(True)
        (dx_ref,) = torch.autograd.grad(F.conv2d(xr, wt, padding=1), xr, dy)
        dx = Act(n, h, w, ci, dev)
        dx.buf.fill_(-3.0)
        dx.buf[: dx.guard * ci] = 0
        dx.buf[-dx.guard * ci:] = 0
        ops.conv_dgrad(Act.from_nchw(dy, dev), ops.dgrad_weight(wn), dx)
        torch.cuda.synchronize()
        assert lib.dispatch_counts(reset=True)["gemm_nt_kernel/wide"] == 1
        assert dx.halo_is_zero()
        _close(dx.to_nchw().cpu(), dx_ref, 1e-2, "dgrad")
    # a linear layer on token rows (no pixel structure: the running-pointer store path), ragged last row tile
    rows = n * h * w - 37
    xl = _bf(torch.randn(rows, ci, generator=g))
    wl = _bf(torch.randn(co, ci, generator=g) / ci ** 0.5)
    rl = _bf(torch.randn(rows, co, generator=g))
    yl = torch.full((rows, co), 5.0, dtype=torch.bfloat16, device=dev)
    xd, wd, rd, bd = xl.to(torch.bfloat16).to(dev), wl.to(torch.bfloat16).to(dev), rl.to(torch.bfloat16).to(dev), bias.to(dev)
    ops.gemm_nt(lib.ptr(xd), ci, wd, lib.ptr(yl), co, rows, co, ci, [0], [0], bias=bd, res_ptr=lib.ptr(rd), ldr=co)
    torch.cuda.synchronize()
    assert lib.dispatch_counts(reset=True)["gemm_nt_kernel/wide"] == (1 if wide else 0)
    _close(yl.float().cpu(), _bf(xl @ wl.t() + bias) + rl, 1e-2, "linear")


# (n, h, w, cin of the 3x3, cout, channels of the shortcut's input, row stride of that input [0 = its channel count], statistics)
SC_CASES = [
    (2, 128, 128, 128, 128, 256, 0, False),      # up-block shape: 256-channel concat -> 128; one tile crosses the image seam
    (1, 256, 256, 128, 128, 256, 0, True),       # two tile rounds; GroupNorm statistics of the sum from the store path
    (4, 64, 64, 256, 256, 128, 0, False),        # down-block shape 128 -> 256: two column tiles, K2 < Kp
    (4, 96, 96, 128, 128, 192, 320, True),       # K2 = 192 (three shortcut groups), the input is a column view of a wider buffer
    (8, 64, 64, 64, 128, 64, 0, False),          # ONE k-chunk per filter row and ONE shortcut group: four groups per tile
]


@pytest.mark.parametrize("n,h,w,ci,co,c2,ld2,stats", SC_CASES)
def test_conv3x3_with_the_1x1_shortcut_folded_in(dev, n, h, w, ci, co, c2, ld2, stats):
    """siss_conv3x3_sc: conv3x3(a; W) + conv1x1(x; W_sc) + both biases + the per-image row bias as ONE product on the
    persistent kernel (ResnetBlock2D: conv2 + conv_shortcut, diffusers resnet.py -> losses/ddpm_deletion_loss.py:24)
    against torch fp32 on the same bf16-rounded operands; the statistics variant against the stored tensor."""
    from siss_amd import lib, ops
    from siss_amd.layout import Act, ActView
    g = torch.Generator().manual_seed(n * 100 + h + c2)
    a = _bf(torch.randn(n, ci, h, w, generator=g))
    x = _bf(torch.randn(n, c2, h, w, generator=g))
    wt = _bf(torch.randn(co, ci, 3, 3, generator=g) * (1.0 / (3 * ci ** 0.5)))
    ws = _bf(torch.randn(co, c2, 1, 1, generator=g) * (1.0 / c2 ** 0.5))
    b1, b2 = torch.randn(co, generator=g), torch.randn(co, generator=g)
    temb = torch.randn(n, co, generator=g)
    ref = F.conv2d(a, wt, b1, padding=1) + F.conv2d(x, ws, b2) + temb[:, :, None, None]

    aa = Act.from_nchw(a, dev)
    if ld2:
        wide = Act(n, h, w, ld2, dev)
        xa = ActView(wide, ld2 - c2, c2)                    # the shortcut input lives in the TAIL columns of a wider buffer
        wide.buf.normal_()                                  # ... whose other columns hold garbage
        xa.data.copy_(Act.from_nchw(x, dev).data)
    else:
        xa = Act.from_nchw(x, dev)
    out = Act(n, h, w, co, dev)
    out.buf.fill_(7.0)
    out.buf[: out.guard * co] = 0
    out.buf[-out.guard * co:] = 0
    assert ops.conv3x3_sc_takes(aa, co, out, xa)
    qs = torch.full((lib.query("siss_conv_qstats_words", aa.rows, co),), float("nan"), device=dev) if stats else None
    lib.dispatch_counts(reset=True)
    wrote = ops.conv_fprop_sc(aa, ops.conv_w_to_native(wt).to(dev).to(torch.bfloat16), out, xa, ws.view(co, c2).to(dev).to(torch.bfloat16),
                              bias=b1.to(dev), bias2=b2.to(dev), rowbias=temb.to(dev), qstats=qs)
    torch.cuda.synchronize()
    cnt = lib.dispatch_counts(reset=True)
    assert cnt["gemm_nt_c3p_kernel"] == 1 and cnt["gemm_nt_kernel"] == 0, cnt
    assert out.halo_is_zero()
    _close(out.to_nchw().cpu(), ref, 1e-2, "conv3x3 + 1x1 shortcut")
    if stats:
        assert wrote and bool(torch.isfinite(qs).all())
        # fold the entries the way the consumer does: per (sample, 4-channel quad) sum / sum of squares of the STORED values
        G = 32
        part = torch.zeros(lib.query("siss_gn_partial_words", n, h, w, co, G), device=dev)
        y2, mean, rstd = Act(n, h, w, co, dev), torch.zeros(n, G, device=dev), torch.zeros(n, G, device=dev)
        gamma, beta = torch.ones(co, device=dev), torch.zeros(co, device=dev)
        lib.call("siss_groupnorm_fwd_qs", out.data, gamma, beta, y2.data, mean, rstd, part, qs, co, None, n, h, w, co, G, 1e-6, 0, 0, 0)
        torch.cuda.synchronize()
        st = out.to_nchw().view(n, G, -1)
        torch.testing.assert_close(mean.cpu(), st.mean(-1).cpu(), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(rstd.cpu(), (st.var(-1, unbiased=False) + 1e-6).rsqrt().cpu(), rtol=1e-3, atol=1e-5)


# (n [cotangent images], h, w, cout of both convs = channels of the cotangent, cin of the 3x3, cin of the 1x1, residual on the 3x3 output)
SC_DGRAD_CASES = [
    (2, 128, 128, 128, 128, 256, False),     # up-block shape: conv2 128 -> 128, shortcut 256 -> 128; a tile crosses the image seam
    (2, 256, 256, 128, 128, 256, True),      # batch-2 cotangent at 256 x 256: several rounds of mixed tiles; running cotangent added to dx
    (4, 64, 64, 256, 256, 128, False),       # down-block shape: K = 256 (four centre-tap groups per x tile), two + one column tiles
    (8, 64, 64, 128, 128, 384, True),        # Nx = 384: three x tiles per row tile
]


@pytest.mark.parametrize("n,h,w,co,ci,cx,res", SC_DGRAD_CASES)
def test_conv3x3_dgrad_with_the_shortcut_dgrad_folded_in(dev, n, h, w, co, ci, cx, res):
    """siss_conv3x3_dgrad_sc: dx = conv3x3^T(dy; W) (+ r) and dx_sc = conv1x1^T(dy; W_sc) from ONE launch of the persistent kernel
    (the backward of ResnetBlock2D's conv2 + conv_shortcut tail, differentiated at delete_celeb.py:691,:702) against torch
    autograd on the same bf16-rounded operands; both outputs' halo must be written as zero."""
    from siss_amd import lib, ops
    from siss_amd.layout import Act
    g = torch.Generator().manual_seed(n * 10 + h + cx)
    dy = _bf(torch.randn(n, co, h, w, generator=g))
    wt = _bf(torch.randn(co, ci, 3, 3, generator=g) * (1.0 / (3 * ci ** 0.5)))
    ws = _bf(torch.randn(co, cx, 1, 1, generator=g) * (1.0 / cx ** 0.5))
    r = _bf(torch.randn(n, ci, h, w, generator=g))
    x3 = torch.zeros(n, ci, h, w, requires_grad=True)
    x1 = torch.zeros(n, cx, h, w, requires_grad=True)
    dx_ref = torch.autograd.grad(F.conv2d(x3, wt, padding=1), x3, dy)[0] + (r if res else 0)
    dxs_ref = torch.autograd.grad(F.conv2d(x1, ws), x1, dy)[0]

    dya = Act.from_nchw(dy, dev)
    wn = ops.conv_w_to_native(wt).to(dev)
    wsn = ws.view(1, co, cx).to(dev)
    wT, wsT = ops.dgrad_weight(wn), ops.dgrad_weight(wsn)[0]                # [9][ci][co], [cx][co]
    dx = Act.from_nchw(r, dev) if res else Act(n, h, w, ci, dev)
    dxs = Act(n, h, w, cx, dev)
    for a in ((dxs,) if res else (dx, dxs)):
        a.buf.fill_(5.0); a.buf[: a.guard * a.c] = 0; a.buf[-a.guard * a.c:] = 0      # poison: the kernel writes the zero halo itself
    assert ops.conv3x3_dgrad_sc_takes(dya, ci, dx, dxs)
    lib.dispatch_counts(reset=True)
    ops.conv_dgrad_sc(dya, wT, dx, wsT, dxs, residual=dx if res else None)
    torch.cuda.synchronize()
    cnt = lib.dispatch_counts(reset=True)
    assert cnt["gemm_nt_c3p_kernel"] == 1 and cnt["gemm_nt_kernel"] == 0, cnt
    assert dx.halo_is_zero() and dxs.halo_is_zero()
    _close(dx.to_nchw().cpu(), dx_ref, 1e-2, "3x3 dgrad")
    _close(dxs.to_nchw().cpu(), dxs_ref, 1e-2, "1x1 shortcut dgrad")


def test_conv3x3_sc_is_refused_where_the_persistent_kernel_is_not_taken(dev):
    from siss_amd import lib, ops
    from siss_amd.layout import Act
    a, x, out = Act(2, 16, 16, 128, dev), Act(2, 16, 16, 256, dev), Act(2, 16, 16, 128, dev)
    assert not ops.conv3x3_sc_takes(a, 128, out, x)         # 16 x 16: the split-K kernels' territory
    with pytest.raises(RuntimeError, match="bad argument"):
        ops.conv_fprop_sc(a, torch.zeros(9, 128, 128, dtype=torch.bfloat16, device=dev), out, x,
                          torch.zeros(128, 256, dtype=torch.bfloat16, device=dev))


def _wgrad_sets(ops, lib, dya, xa, dW, dB, nsets):
    """conv_wgrad with a bias-gradient buffer per set (set stride = co floats)."""
    from siss_amd.layout import conv3x3_panels
    co, ci = dya.c, xa.c
    b = dya.n // nsets
    shifts, coffs = conv3x3_panels(dya.wp, ci)
    rows_per_set = b * dya.rows_per_image
    # dbias[set * set_stride + n] uses dW's set stride (9*co*ci floats), so give it a buffer laid out that way
    big = torch.zeros(nsets, 9 * co * ci, device=dW.device)
    lib.call("siss_gemm_tn", dya.data, dya.c, xa.data, xa.c, dW, 9 * co * ci, co, ci, 9, lib.int_array(shifts),
             lib.int_array(coffs), nsets, rows_per_set, 0, dya.wp + 1, rows_per_set - (dya.wp + 1), 0,
             ops.zero_page(dW.device), big, None)
    dB.copy_(big[:, :co])


def test_dispatch_counters_tell_small_from_large(dev):
    """The counters themselves: a 16x16 conv is NOT on the persistent kernel, a 128x128 one is."""
    from siss_amd import lib, ops
    from siss_amd.layout import Act
    for hw, want in ((16, "gemm_nt_kernel"), (128, "gemm_nt_c3p_kernel")):
        x = Act(2, hw, hw, 128, dev)
        y = Act(2, hw, hw, 128, dev)
        wn = torch.zeros(9, 128, 128, dtype=torch.bfloat16, device=dev)
        lib.dispatch_counts(reset=True)
        ops.conv_fprop(x, wn, y)
        torch.cuda.synchronize()
        cnt = lib.dispatch_counts(reset=True)
        assert cnt[want] + (cnt["gemm_nt_kernel/splitk"] if hw == 16 else 0) == 1, (hw, cnt)


# ---------------------------------------------------------------------------------------------------------------------
# Full-size network: BASELINE configs[1] (CelebA-HQ 256x256 UNet, 113.7 M parameters, 450 tensors), forward + ONE
# dual-cotangent backward on HIP against the oracle network (oracle/unet.py, fp32, torch autograd: two backward calls
# with retain_graph like delete_celeb.py:686-711) evaluated on the same GPU through PyTorch-ROCm's own fp32 kernels.
# Tolerances: pred max-err <= 3e-2 * max|pred|; per-tensor gradient cosine >= 0.99 for all 450 tensors
# of both sets (the six attention key biases have an identically zero gradient: |g| < 1e-6 of the total instead);
# set norms within 5e-2.
# ---------------------------------------------------------------------------------------------------------------------
def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


@pytest.mark.parametrize("B", [4, 16])
def test_full_size_forward_and_dual_backward_match_the_fp32_oracle(dev, B):
    from siss_amd import lib
    from siss_amd.config import UNet2DConfig
    from siss_amd.unet import UNetEngine
    from oracle.unet import OracleUNet2D, UNetConfig
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    HW = 256
    eng = UNetEngine(UNet2DConfig.celebahq256(), dev)
    sd = eng.init_random(seed=42)
    assert len(sd) == 450 and sum(v.numel() for v in sd.values()) == 113_673_219
    net = OracleUNet2D(UNetConfig.celebahq256())
    net.load_state_dict(sd)
    net = net.to(dev).float()
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(B, 3, HW, HW, generator=g, device=dev).to(torch.bfloat16)      # bf16 I/O mode (configs[1])
    t = torch.tensor(([999, 999, 250, 3] + [999] * B)[:B], device=dev)
    cx = torch.randn(B, 3, HW, HW, generator=g, device=dev) * 1e-3
    ca = torch.randn(B, 3, HW, HW, generator=g, device=dev) * 1e-3

    lib.dispatch_counts(reset=True)
    pred = eng.forward(x, t).clone()
    eng.zero_grad()
    eng.backward(torch.cat([cx, ca]).contiguous(), nsets=2)
    torch.cuda.synchronize()
    cnt = lib.dispatch_counts(reset=True)
    # the kernels under test really ran: fprop + dgrad of the >= 64x64 levels on c3p, their wgrads on <3>
    assert cnt["gemm_nt_c3p_kernel"] >= 40 and cnt["gemm_tn_kernel<3>"] >= 30, cnt

    ref = net(x.float(), t)[0]
    err = (pred - ref.detach()).abs().max().item()
    scale = ref.detach().abs().max().item()
    assert err <= 3e-2 * scale, (err, scale)

    names = [n for n, _ in net.named_parameters()]
    params = [p for _, p in net.named_parameters()]
    bad, worst = [], (1.0, None)
    for s, c in enumerate((cx, ca)):
        grads = torch.autograd.grad(ref, params, c, retain_graph=(s == 0))
        got = {n: v.to(dev) for n, v in eng.ps.grads_ref(s).items()}
        tot_r = torch.sqrt(sum(v.double().square().sum() for v in grads))
        tot_g = torch.sqrt(sum(v.double().square().sum() for v in got.values()))
        assert abs(float(tot_g / tot_r) - 1) < 5e-2, (s, float(tot_g), float(tot_r))
        for n, r in zip(names, grads):
            if float(r.norm()) < 1e-8 * float(tot_r):
                # a mathematically ZERO gradient (to_k.bias: softmax is invariant to a constant added to every key's
                # logit): the oracle's value is f32 rounding noise, so the direction means nothing -- the HIP value
                # must be just as negligible
                assert n.endswith("to_k.bias"), (n, float(r.norm()))
                assert float(got[n].norm()) < 1e-6 * float(tot_r), (n, float(got[n].norm()), float(tot_r))
                continue
            c_ = _cos(got[n], r)
            if c_ < worst[0]:
                worst = (c_, (s, n))
            if c_ < 0.99:
                bad.append((s, n, round(c_, 4), float(r.norm() / tot_r)))
        del grads
    print(f"\nfull-size parity: pred rel err {err / scale:.3g}; worst per-tensor gradient cosine {worst[0]:.5f} at {worst[1]}")
    assert not bad, (len(bad), bad[:12])


def test_full_size_no_is_double_forward_matches_the_fp32_oracle(dev):
    """BASELINE configs[3] (SISS-No-IS, losses/ddpm_deletion_loss.py:60-67) at full size: ONE 32-image forward (the keep
    batch q_sample(x0) stacked on the forget batch q_sample(a0), B = 16 each) and the dual backward with
    x_set_rows = rows_per_set -- set 0 differentiates images [0, 16), set 1 images [16, 32): other grids and tile rounds
    than the shared-forward SISS step.  Oracle: the fp32 torch network on the same GPU, one 16-image half at a time (samples
    do not interact: GroupNorm and attention are per sample), which is exactly g_x / g_a of delete_celeb.py:691,:702."""
    from siss_amd import lib
    from siss_amd.config import UNet2DConfig
    from siss_amd.unet import UNetEngine
    from oracle.unet import OracleUNet2D, UNetConfig
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    B, HW = 16, 256
    eng = UNetEngine(UNet2DConfig.celebahq256(), dev)
    sd = eng.init_random(seed=43)
    net = OracleUNet2D(UNetConfig.celebahq256())
    net.load_state_dict(sd)
    net = net.to(dev).float()
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(2 * B, 3, HW, HW, generator=g, device=dev).to(torch.bfloat16)
    t = torch.full((2 * B,), 999, device=dev)
    t[1], t[B + 1] = 250, 250
    cot = torch.randn(2 * B, 3, HW, HW, generator=g, device=dev) * 1e-3

    lib.dispatch_counts(reset=True)
    pred = eng.forward(x, t).clone()
    eng.zero_grad()
    eng.backward(cot.contiguous(), nsets=2)          # nb = nf = 32: every set reads its OWN saved rows
    torch.cuda.synchronize()
    cnt = lib.dispatch_counts(reset=True)
    assert cnt["gemm_nt_c3p_kernel"] >= 40 and cnt["gemm_tn_kernel<3>"] >= 30, cnt

    names = [n for n, _ in net.named_parameters()]
    params = [p for _, p in net.named_parameters()]
    grads_by_set, err, scale = [], 0.0, 0.0
    for s in range(2):
        sl = slice(s * B, (s + 1) * B)
        ref = net(x[sl].float(), t[sl])[0]
        err = max(err, (pred[sl] - ref.detach()).abs().max().item())
        scale = max(scale, ref.detach().abs().max().item())
        grads_by_set.append(torch.autograd.grad(ref, params, cot[sl]))
        del ref
    assert err <= 3e-2 * scale, (err, scale)
    from parity_util import assert_grads_match
    worst = assert_grads_match(eng, names, grads_by_set, dev)
    print(f"\nfull-size No-IS parity: pred rel err {err / scale:.3g}; worst per-tensor gradient cosine {worst[0]:.5f} at {worst[1]}")


def test_grouped_weight_gradient_launch_equals_single_launches(dev):
    """siss_gemm_tn_grouped: a mixed job table (3x3 filters on two resolutions -> the fused 3-tap variant, a linear and a
    1x1 conv -> the one-tap variant; one and two cotangent sets; bias gradients) in ONE call against the same products
    launched one by one with siss_gemm_tn.  Both accumulate through f32 atomics / plain adds: equal to f32 rounding."""
    from siss_amd import lib, ops
    from siss_amd.layout import Act, conv3x3_panels
    g = torch.Generator().manual_seed(17)
    zp = ops.zero_page(dev)
    jobs, singles, outs = [], [], []
    for (n, hw, ci, co, nsets, ksize) in ((2, 16, 128, 256, 2, 3), (4, 8, 512, 512, 2, 3), (2, 32, 256, 128, 1, 3), (2, 16, 384, 128, 2, 1)):
        x = Act.from_nchw(_bf(torch.randn(n, ci, hw, hw, generator=g)), dev)
        dy = Act.from_nchw(_bf(torch.randn(nsets * n, co, hw, hw, generator=g)), dev)
        shifts, coffs = conv3x3_panels(dy.wp, ci) if ksize == 3 else ([0], [0])
        t = len(shifts)
        rps = n * dy.rows_per_image
        rb, re = dy.wp + 1, rps - (dy.wp + 1)
        pair = []
        for which in range(2):
            dW = torch.zeros(nsets, t * co * ci + co, device=dev)          # [dW | dbias] per set, set stride = row length
            pair.append(dW)
            args = dict(Y=dy.data.data_ptr(), ldy=co, X=x.data.data_ptr(), ldx=ci, dW=dW.data_ptr(), set_stride=dW.shape[1],
                        N=co, C=ci, npanels=t, nsets=nsets, rows_per_set=rps, row_begin=rb, row_end=re, nsplits=0, x_set_rows=0,
                        zero_page=zp.data_ptr(), dbias=dW[0, t * co * ci:].data_ptr(), dbias2=None,
                        shifts=(lib.I * 9)(*shifts, *([0] * (9 - t))), coffs=(lib.I * 9)(*coffs, *([0] * (9 - t))))
            if which == 0:
                jobs.append((lib.TNJob(**args), (x, dy, dW)))
            else:
                singles.append((dy, x, dW, co, ci, t, shifts, coffs, nsets, rps, rb, re))
        outs.append(pair)
    lib.dispatch_counts(reset=True)
    arr = (lib.TNJob * len(jobs))(*[j for j, _ in jobs])
    lib.call("siss_gemm_tn_grouped", arr, len(jobs))
    torch.cuda.synchronize()
    cnt = lib.dispatch_counts(reset=True)
    assert cnt["gemm_tn_kernel<3>"] == 3 and cnt["gemm_tn_kernel<1>"] == 1, cnt
    for (dy, x, dW, co, ci, t, shifts, coffs, nsets, rps, rb, re) in singles:
        lib.call("siss_gemm_tn", dy.data, co, x.data, ci, dW, dW.shape[1], co, ci, t, lib.int_array(shifts), lib.int_array(coffs),
                 nsets, rps, 0, rb, re, 0, zp, dW[0, t * co * ci:], None)
    torch.cuda.synchronize()
    for grouped, single in outs:
        assert float(single.abs().max()) > 0
        _close(grouped.cpu(), single.cpu(), 1e-5, "grouped vs single")


@pytest.mark.parametrize("n,hw,ci,co,c2,n1,blocks", [(4, 64, 128, 128, 256, 128, 0), (4, 64, 128, 128, 256, 128, 40), (2, 48, 256, 128, 128, 27, 64), (3, 40, 128, 256, 64, 256, 40)])
def test_paired_weight_gradient_launch_equals_single_launches(dev, n, hw, ci, co, c2, n1, blocks):
    """siss_gemm_tn_pair: a 3x3 convolution's weight gradient (fused 3-tap body) and a ONE-PANEL product (a resnet's 1x1
    conv_shortcut over the same cotangent; conv_out's 27-row product; a 64-column input) in one launch of one round of blocks --
    3-tap blocks beside 512-thread blocks that each run two virtual one-tap blocks, uneven last splits, partial tiles -- against
    the two products launched one by one with siss_gemm_tn.  Two cotangent sets, bias gradients; equal to f32 rounding."""
    from siss_amd import lib, ops
    from siss_amd.layout import Act, conv3x3_panels
    g = torch.Generator().manual_seed(n + hw + c2)
    zp = ops.zero_page(dev)
    nsets = 2
    x3 = Act.from_nchw(_bf(torch.randn(n, ci, hw, hw, generator=g)), dev)
    dy3 = Act.from_nchw(_bf(torch.randn(nsets * n, co, hw, hw, generator=g)), dev)
    x1 = Act.from_nchw(_bf(torch.randn(n, c2, hw, hw, generator=g)), dev)
    # the one-panel product's cotangent: conv2's own (the shortcut case) or another tensor (conv_out: 27 columns of a 64-wide matrix)
    dy1 = dy3 if n1 == co else Act.from_nchw(_bf(torch.randn(nsets * n, 64, hw, hw, generator=g)), dev)
    rps = n * dy3.rows_per_image
    rb, re = dy3.wp + 1, rps - (dy3.wp + 1)
    s3, c3 = conv3x3_panels(dy3.wp, ci)
    outs = []
    for which in range(2):
        dW3 = torch.zeros(nsets, 9 * co * ci + co, device=dev)
        dW1 = torch.zeros(nsets, n1 * c2 + n1, device=dev)
        outs.append((dW3, dW1))
        if which == 0:
            j3 = lib.TNJob(Y=dy3.data.data_ptr(), ldy=co, X=x3.data.data_ptr(), ldx=ci, dW=dW3.data_ptr(), set_stride=dW3.shape[1], N=co, C=ci,
                           npanels=9, nsets=nsets, rows_per_set=rps, row_begin=rb, row_end=re, nsplits=0, x_set_rows=0, zero_page=zp.data_ptr(),
                           dbias=dW3[0, 9 * co * ci:].data_ptr(), dbias2=None, shifts=(lib.I * 9)(*s3), coffs=(lib.I * 9)(*c3))
            z9 = (lib.I * 9)(*([0] * 9))
            j1 = lib.TNJob(Y=dy1.data.data_ptr(), ldy=dy1.c, X=x1.data.data_ptr(), ldx=c2, dW=dW1.data_ptr(), set_stride=dW1.shape[1], N=n1, C=c2,
                           npanels=1, nsets=nsets, rows_per_set=rps, row_begin=rb, row_end=re, nsplits=0, x_set_rows=0, zero_page=zp.data_ptr(),
                           dbias=dW1[0, n1 * c2:].data_ptr(), dbias2=None, shifts=z9, coffs=z9)
            lib.dispatch_counts(reset=True)
            lib.call("siss_gemm_tn_pair", lib.C.byref(j3), lib.C.byref(j1), blocks)
            torch.cuda.synchronize()
            assert lib.dispatch_counts(reset=True)["gemm_tn_pair"] == 1
        else:
            lib.call("siss_gemm_tn", dy3.data, co, x3.data, ci, dW3, dW3.shape[1], co, ci, 9, lib.int_array(s3), lib.int_array(c3),
                     nsets, rps, 0, rb, re, 0, zp, dW3[0, 9 * co * ci:], None)
            lib.call("siss_gemm_tn", dy1.data, dy1.c, x1.data, c2, dW1, dW1.shape[1], n1, c2, 1, lib.int_array([0]), lib.int_array([0]),
                     nsets, rps, 0, rb, re, 0, zp, dW1[0, n1 * c2:], None)
            torch.cuda.synchronize()
    (p3, p1), (q3, q1) = outs
    assert float(q3.abs().max()) > 0 and float(q1.abs().max()) > 0
    _close(p3.cpu(), q3.cpu(), 1e-5, "paired 3-tap product vs single")
    _close(p1.cpu(), q1.cpu(), 1e-5, "paired one-panel product vs single")
    # and the one-panel product against torch: dW[set][n][c] = sum over the set's pixels of dy[n] x[c]
    dyf, xf = dy1.to_nchw()[:, :n1].cpu(), x1.to_nchw().cpu()
    for k in range(nsets):
        ref = torch.einsum("bnhw,bchw->nc", dyf[k * n:(k + 1) * n], xf)
        _close(p1[k, :n1 * c2].view(n1, c2).cpu(), ref, 2e-3, f"one-panel product set {k} vs torch")
