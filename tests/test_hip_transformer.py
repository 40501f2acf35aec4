"""Parity of the token-space kernels of the SD UNet (transformer.hip) against plain torch fp32 on the same
bf16-rounded inputs.  Tolerances: outputs are rounded to bf16 -> rel 1e-2 of the tensor's scale; f32 column
sums (dgamma / dbeta) rel 2e-3."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("rows,C", [(96, 320), (130, 640), (64, 1280), (24, 64)])
def test_layernorm_fwd_bwd_two_sets(dev, rows, C):
    from siss_amd import lib
    g = torch.Generator().manual_seed(rows + C)
    x = _bf(torch.randn(rows, C, generator=g) * 2 + 0.5)
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    dy = _bf(torch.randn(2 * rows, C, generator=g))            # two cotangent sets over the same saved rows
    acc = _bf(torch.randn(2 * rows, C, generator=g))
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    y_ref = F.layer_norm(xr, (C,), gr, br, 1e-5)
    refs = []
    for k in range(2):
        gx, gg, gb = torch.autograd.grad(y_ref, (xr, gr, br), dy[k * rows:(k + 1) * rows], retain_graph=True)
        refs.append((gx, gg, gb))

    xd, yd = x.to(dev).to(torch.bfloat16), torch.empty(rows, C, dtype=torch.bfloat16, device=dev)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    lib.call("siss_layernorm_fwd", xd, gamma.to(dev), beta.to(dev), yd, mean, rstd, rows, C, 1e-5)
    _close(yd.float().cpu(), y_ref.detach(), 1e-2, "ln fwd")
    dx = torch.empty(2 * rows, C, dtype=torch.bfloat16, device=dev)
    P = 4096
    grads = torch.zeros(2, P, device=dev)
    lib.call("siss_layernorm_bwd", dy.to(dev).to(torch.bfloat16), xd, gamma.to(dev), mean, rstd,
             acc.to(dev).to(torch.bfloat16), dx, grads[0, 100:], grads[0, 2000:], 2 * rows, rows, rows, P, C)
    torch.cuda.synchronize()
    for k in range(2):
        _close(dx[k * rows:(k + 1) * rows].float().cpu(), refs[k][0] + acc[k * rows:(k + 1) * rows], 1.5e-2, f"ln dx set {k}")
        _close(grads[k, 100:100 + C].cpu(), refs[k][1], 2e-3, f"ln dgamma set {k}")
        _close(grads[k, 2000:2000 + C].cpu(), refs[k][2], 2e-3, f"ln dbeta set {k}")


def test_geglu_fwd_bwd(dev):
    from siss_amd import lib
    g = torch.Generator().manual_seed(3)
    rows, Fd = 70, 256
    h = _bf(torch.randn(rows, 2 * Fd, generator=g) * 1.5)
    dout = _bf(torch.randn(2 * rows, Fd, generator=g))
    hr = h.clone().requires_grad_(True)
    a, gg = hr.chunk(2, dim=-1)
    out_ref = a * F.gelu(gg)
    hd = h.to(dev).to(torch.bfloat16)
    out = torch.empty(rows, Fd, dtype=torch.bfloat16, device=dev)
    lib.call("siss_geglu_fwd", hd, out, rows, Fd)
    _close(out.float().cpu(), out_ref.detach(), 1e-2, "geglu fwd")
    dh = torch.empty(2 * rows, 2 * Fd, dtype=torch.bfloat16, device=dev)
    lib.call("siss_geglu_bwd", dout.to(dev).to(torch.bfloat16), hd, dh, 2 * rows, rows, Fd)
    for k in range(2):
        (gh,) = torch.autograd.grad(out_ref, hr, dout[k * rows:(k + 1) * rows], retain_graph=True)
        _close(dh[k * rows:(k + 1) * rows].float().cpu(), gh, 1e-2, f"geglu bwd set {k}")


@pytest.mark.parametrize("rows,Fd,K", [(192, 256, 64), (8192, 1280, 320), (100, 640, 128)])
def test_geglu_backward_in_the_gemm_epilogue(dev, rows, Fd, K):
    """siss_gemm_nt_geglu_bwd: the output projection's dgrad with the GEGLU backward in its epilogue, against the two launches it
    replaces (siss_gemm_nt into a [rows2, F] cotangent, then siss_geglu_bwd) -- the same arithmetic on the same bf16-rounded
    cotangent: BITWISE equal (three grids: one tile round, the large-grid kernel, a ragged row tile with split K)."""
    from siss_amd import lib
    lib.ensure_workspace(dev)
    g = torch.Generator().manual_seed(rows + Fd)
    sets = 2
    h = (torch.randn(rows, 2 * Fd, generator=g) * 1.5).to(torch.bfloat16).to(dev)
    dy = torch.randn(sets * rows, K, generator=g).to(torch.bfloat16).to(dev)           # cotangent of y = out W^T
    wT = (torch.randn(Fd, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)      # [F][K]: d(out) = dy wT^T
    z = lib.int_array([0])
    dout = torch.empty(sets * rows, Fd, dtype=torch.bfloat16, device=dev)
    lib.call("siss_gemm_nt", dy, K, wT, dout, Fd, None, None, Fd, None, 0, sets * rows, Fd, K, 1, z, z, 1, 0, 0, 1.0, 1, 0, 0, 0)
    ref = torch.full((sets * rows, 2 * Fd), 3.0, dtype=torch.bfloat16, device=dev)
    lib.call("siss_geglu_bwd", dout, h, ref, sets * rows, rows, Fd)
    got = torch.full((sets * rows, 2 * Fd), 5.0, dtype=torch.bfloat16, device=dev)
    lib.call("siss_gemm_nt_geglu_bwd", dy, K, wT, got, h, rows, sets * rows, Fd, K)
    torch.cuda.synchronize()
    assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())
    # and against autograd of a * gelu(g) through the projection, in f32
    hr = h.float().cpu().requires_grad_(True)
    a, gg = hr.chunk(2, dim=-1)
    out_ref = a * F.gelu(gg)
    for k in range(sets):
        d_out = (dy[k * rows:(k + 1) * rows].float() @ wT.float().t()).cpu()
        (gh,) = torch.autograd.grad(out_ref, hr, d_out, retain_graph=True)
        _close(got[k * rows:(k + 1) * rows].float().cpu(), gh, 1.5e-2, f"fused geglu bwd set {k}")


@pytest.mark.parametrize("rows,Fd,K", [(192, 256, 64), (8192, 1280, 320), (100, 640, 128), (4096, 5120, 1280), (300, 1280, 768)])
def test_geglu_forward_in_the_gemm_epilogue(dev, rows, Fd, K):
    """siss_gemm_nt_geglu_fwd: GEGLU's input projection h = x W^T + b with y = a * gelu(g) formed in its epilogue (a tile holds 64
    value columns and the 64 gate columns that go with them), against the two launches it replaces (siss_gemm_nt into h, then
    siss_geglu_fwd): the same K loops per element and the same bf16-rounded h: BITWISE equal h and y (grids: one tile round, the
    large-grid kernel, ragged row tiles, split K), and against torch in f32."""
    from siss_amd import lib
    lib.ensure_workspace(dev)
    g = torch.Generator().manual_seed(rows + Fd + K)
    x = torch.randn(rows, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(2 * Fd, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = (0.3 * torch.randn(2 * Fd, generator=g)).to(dev)
    z = lib.int_array([0])
    h_ref = torch.full((rows, 2 * Fd), 3.0, dtype=torch.bfloat16, device=dev)
    lib.call("siss_gemm_nt", x, K, w, h_ref, 2 * Fd, b, None, 0, None, 0, rows, 2 * Fd, K, 1, z, z, 1, 0, 0, 1.0, 1, 0, 0, 0)
    y_ref = torch.full((rows, Fd), 3.0, dtype=torch.bfloat16, device=dev)
    lib.call("siss_geglu_fwd", h_ref, y_ref, rows, Fd)
    h = torch.full((rows, 2 * Fd), 5.0, dtype=torch.bfloat16, device=dev)
    y = torch.full((rows, Fd), 5.0, dtype=torch.bfloat16, device=dev)
    lib.call("siss_gemm_nt_geglu_fwd", x, K, w, b, h, y, rows, Fd, K)
    torch.cuda.synchronize()
    assert torch.equal(h, h_ref), float((h.float() - h_ref.float()).abs().max())
    assert torch.equal(y, y_ref), float((y.float() - y_ref.float()).abs().max())
    hf = x.float() @ w.float().t() + b
    a, gg = hf.chunk(2, dim=-1)
    _close(y.float().cpu(), (a * F.gelu(gg)).cpu(), 1.5e-2, "fused geglu fwd")


@pytest.mark.parametrize("B,S,H,D,Sp,Dp", [(2, 64, 8, 40, 64, 64), (3, 7, 2, 32, 64, 64), (1, 77, 8, 160, 128, 192)])
def test_head_split_merge_roundtrip(dev, B, S, H, D, Sp, Dp):
    from siss_amd import lib
    g = torch.Generator().manual_seed(B + S)
    x = torch.randn(B, S, H * D, generator=g).to(torch.bfloat16).to(dev)
    hs = torch.full((B * H, Sp, Dp), 7.0, dtype=torch.bfloat16, device=dev)
    lib.call("siss_head_split", x, hs, B, S, H, D, Sp, Dp)
    ref = x.view(B, S, H, D).permute(0, 2, 1, 3).reshape(B * H, S, D)
    assert torch.equal(hs[:, :S, :D], ref)
    assert float(hs[:, S:, :].abs().sum()) == 0 and float(hs[:, :, D:].abs().sum()) == 0
    back = torch.empty_like(x)
    lib.call("siss_head_merge", hs, back, B, S, H, D, Sp, Dp)
    assert torch.equal(back, x)


@pytest.mark.parametrize("rows,valid,ld", [(40, 77, 128), (16, 4096, 4096), (33, 64, 64)])
def test_softmax_rows_fwd_bwd(dev, rows, valid, ld):
    from siss_amd import lib
    g = torch.Generator().manual_seed(valid)
    s = _bf(torch.randn(rows, ld, generator=g) * 3)
    dp = _bf(torch.randn(2 * rows, ld, generator=g))
    sr = s[:, :valid].clone().requires_grad_(True)
    p_ref = torch.softmax(sr, dim=-1)
    sd = s.to(dev).to(torch.bfloat16)
    p = torch.full((rows, ld), 5.0, dtype=torch.bfloat16, device=dev)
    lib.call("siss_softmax_rows_fwd", sd, p, rows, valid, ld, 0)
    _close(p[:, :valid].float().cpu(), p_ref.detach(), 1e-2, "softmax fwd")
    assert float(p[:, valid:].abs().sum()) == 0
    ds = torch.full((2 * rows, ld), 5.0, dtype=torch.bfloat16, device=dev)
    scale = 0.37
    lib.call("siss_softmax_rows_bwd", p, dp.to(dev).to(torch.bfloat16), ds, 2 * rows, rows, valid, ld, scale)
    pb = p[:, :valid].float().cpu()
    for k in range(2):
        d = dp[k * rows:(k + 1) * rows, :valid]
        ref = scale * pb * (d - (pb * d).sum(-1, keepdim=True))
        _close(ds[k * rows:(k + 1) * rows, :valid].float().cpu(), ref, 2e-2, f"softmax bwd set {k}")
    assert float(ds[:, valid:].abs().sum()) == 0


@pytest.mark.parametrize("S,ld", [(77, 128), (40, 64)])
def test_softmax_rows_causal(dev, S, ld):
    """CLIP text-encoder mask: row r of every (sample, head) block of `ld` rows sees keys k <= r."""
    from siss_amd import lib
    g = torch.Generator().manual_seed(S)
    heads = 3
    s = _bf(torch.randn(heads, ld, ld, generator=g) * 2)
    mask = torch.full((ld, ld), float("-inf")).triu(1)
    ref = torch.softmax(s[:, :S, :S] + mask[:S, :S], dim=-1)
    p = torch.full((heads * ld, ld), 5.0, dtype=torch.bfloat16, device=dev)
    lib.call("siss_softmax_rows_fwd", s.reshape(-1, ld).to(dev).to(torch.bfloat16), p, heads * ld, S, ld, ld)
    p = p.view(heads, ld, ld).float().cpu()
    _close(p[:, :S, :S], ref, 1e-2, "causal softmax")
    assert float(p[:, :S, S:].abs().sum()) == 0
    assert float(p[:, :S, :S].triu(1).abs().sum()) == 0


def test_quick_gelu(dev):
    from siss_amd import lib
    x = _bf(torch.randn(37, 256) * 3)
    y = torch.empty(37, 256, dtype=torch.bfloat16, device=dev)
    lib.call("siss_quick_gelu", x.to(dev).to(torch.bfloat16), y, x.numel())
    _close(y.float().cpu(), x * torch.sigmoid(1.702 * x), 1e-2, "quick_gelu")


def test_gemm_nt_mulsub_and_rowdot(dev):
    """dS = scale * P o (dO V^T - delta) from the product's epilogue, delta = rowsum(dO o O) (attention backward)."""
    from siss_amd import lib
    g = torch.Generator().manual_seed(9)
    BH, Sq, Sk, D = 3, 192, 128, 64
    do = _bf(torch.randn(2 * BH, Sq, D, generator=g))          # two cotangent groups over the same forward heads
    o = _bf(torch.randn(BH, Sq, D, generator=g))
    v = _bf(torch.randn(BH, Sk, D, generator=g))
    p = _bf(torch.softmax(torch.randn(BH, Sq, Sk, generator=g), -1))
    scale = 0.25
    d_do, d_o, d_v, d_p = (t.to(dev).to(torch.bfloat16).contiguous() for t in (do, o, v, p))
    delta = torch.empty(2 * BH * Sq, device=dev)
    lib.call("siss_rowdot", d_do, d_o, delta, 2 * BH * Sq, BH * Sq, D)
    ref_delta = (do * o.repeat(2, 1, 1)).sum(-1)
    _close(delta.view(2 * BH, Sq).cpu(), ref_delta, 1e-5, "rowdot")
    ds = torch.empty(2 * BH, Sq, Sk, dtype=torch.bfloat16, device=dev)
    for grp in range(2):
        sl = slice(grp * BH, (grp + 1) * BH)
        lib.call("siss_gemm_nt_mulsub", d_do[sl], D, d_v, ds[sl], Sk, d_p, Sk, delta[grp * BH * Sq:], Sq, Sk, D, scale,
                 BH, Sq * D, Sk * D, Sq * Sk)
        ref = scale * p * (do[sl] @ v.transpose(1, 2) - ref_delta[sl].unsqueeze(-1))
        _close(ds[sl].float().cpu(), ref, 1.5e-2, f"mulsub group {grp}")
