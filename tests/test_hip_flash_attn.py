"""Fused attention kernels (csrc/flash_attn.hip) against torch fp32 softmax(scale Q K^T) V and its autograd gradients on the
same bf16-rounded operands -- the computation diffusers' BasicTransformerBlock attn1 / attn2 hands to
scaled_dot_product_attention (reached from delete_sd.py:977-985 through losses/ddpm_deletion_loss.py:24, differentiated
twice at delete_sd.py:1040-1060: here one dual-cotangent backward, 2 x BH cotangent entries against BH forward entries).

Shapes: the SD-v1.5 head dims 40 / 80 / 160 (padded to 64 / 128 / 192), self-attention with ragged sequence lengths
(padded queries AND padded keys), cross-attention onto 77 text tokens (masked keys in the second 64-key tile), several
key / query tiles (online-softmax rescaling across tiles), large score ranges (max subtraction).
Tolerances: O rel 1e-2 of scale (bf16 output, bf16 P operand), lse abs 2e-3, dQ / dK / dV rel 2e-2 of scale.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from siss_amd import lib
    lib.load()
    lib.ensure_workspace("cuda:0")
    return torch.device("cuda:0")


def _up(n, m):
    return -(-n // m) * m


def _bf(x):
    return x.to(torch.bfloat16).float()


def _close(got, ref, rel, what=""):
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g} (rel {err / scale:.3g} > {rel})"


def _pad(x, Sp, Dp, dev):
    BH, S, D = x.shape
    out = torch.zeros(BH, Sp, Dp, dtype=torch.bfloat16, device=dev)
    out[:, :S, :D] = x.to(torch.bfloat16).to(dev)
    return out


CASES = [  # BH, Sq, Sk, D, logit spread
    (4, 256, 256, 40, 1.0), (2, 1024, 77, 40, 1.0), (4, 200, 200, 80, 1.0), (2, 64, 64, 160, 1.0),
    (3, 320, 77, 160, 1.0), (2, 192, 448, 64, 6.0), (1, 4096, 4096, 40, 1.0),
]


@pytest.mark.parametrize("BH,Sq,Sk,D,spread", CASES)
def test_flash_attention_forward_and_dual_backward(dev, BH, Sq, Sk, D, spread):
    from siss_amd import lib
    g = torch.Generator().manual_seed(BH * 1000 + Sq + Sk + D)
    q = _bf(torch.randn(BH, Sq, D, generator=g) * spread)
    k = _bf(torch.randn(BH, Sk, D, generator=g))
    v = _bf(torch.randn(BH, Sk, D, generator=g))
    sets = 2
    do = _bf(torch.randn(sets * BH, Sq, D, generator=g))
    scale = D ** -0.5
    big = Sq * Sk > 4_000_000
    rdev = dev if big else torch.device("cpu")                  # the 4096^2 case: fp32 reference on the GPU through torch
    qr, kr, vr = (t.to(rdev).clone().requires_grad_(True) for t in (q, k, v))
    s = (qr @ kr.transpose(1, 2)) * scale
    p = torch.softmax(s, dim=-1)
    o_ref = p @ vr
    lse_ref = torch.logsumexp(s, dim=-1) / math.log(2.0)
    grads = [torch.autograd.grad(o_ref, (qr, kr, vr), do[i * BH:(i + 1) * BH].to(rdev), retain_graph=True) for i in range(sets)]

    Dp, Sqp, Skp = _up(D, 64), _up(Sq, 64), _up(Sk, 64)
    qh, kh, vh = _pad(q, Sqp, Dp, dev), _pad(k, Skp, Dp, dev), _pad(v, Skp, Dp, dev)
    oh = torch.full((BH, Sqp, Dp), 7.0, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(BH, Sqp, device=dev)
    lib.call("siss_flash_attn_fwd", qh, kh, vh, oh, lse, BH, Sqp, Skp, Dp, Sk, float(scale))
    torch.cuda.synchronize()
    _close(oh[:, :Sq, :D].float().cpu(), o_ref.detach().cpu(), 1e-2, "O")
    if Dp > D:
        assert float(oh[:, :, D:].float().abs().max()) == 0.0, "padded head columns must stay zero"
    assert (lse[:, :Sq].cpu() - lse_ref.detach().cpu()).abs().max() < 2e-3

    doh = _pad(do, Sqp, Dp, dev)
    delta = torch.zeros(sets * BH * Sqp, device=dev)
    lib.call("siss_rowdot", doh, oh, delta, sets * BH * Sqp, BH * Sqp, Dp)
    dq, dk, dv = (torch.full((sets * BH, n, Dp), 3.0, dtype=torch.bfloat16, device=dev) for n in (Sqp, Skp, Skp))
    lib.call("siss_flash_attn_bwd", qh, kh, vh, doh, lse, delta, dq, dk, dv, sets * BH, BH, Sqp, Skp, Dp, Sk, float(scale))
    torch.cuda.synchronize()
    for i in range(sets):
        sl = slice(i * BH, (i + 1) * BH)
        _close(dq[sl, :Sq, :D].float().cpu(), grads[i][0].cpu(), 2e-2, f"dQ set {i}")
        _close(dk[sl, :Sk, :D].float().cpu(), grads[i][1].cpu(), 2e-2, f"dK set {i}")
        _close(dv[sl, :Sk, :D].float().cpu(), grads[i][2].cpu(), 2e-2, f"dV set {i}")
    # padded keys take no gradient, padded head columns stay zero
    if Skp > Sk:
        assert float(dk[:, Sk:].float().abs().max()) == 0.0 and float(dv[:, Sk:].float().abs().max()) == 0.0
    if Dp > D:
        assert float(dq[:, :, D:].float().abs().max()) == 0.0


def test_flash_attention_rejects_unsupported_shapes(dev):
    from siss_amd import lib
    t = torch.zeros(1, 64, 256, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(64, device=dev)
    with pytest.raises(RuntimeError, match="bad argument"):
        lib.call("siss_flash_attn_fwd", t, t, t, t, lse, 1, 64, 64, 256, 64, 1.0)      # D_pad 256: not covered
    with pytest.raises(RuntimeError, match="bad argument"):
        lib.call("siss_flash_attn_fwd", t, t, t, t, lse, 1, 60, 64, 64, 64, 1.0)       # Sq_pad not a multiple of 64


MERGED_CASES = [  # B, H, Sq, Sk, D, extra row-stride columns
    (2, 8, 256, 256, 40, 0),      # SD 64-channel-per-... level shape: 8 heads of 40 in 320-wide rows
    (2, 8, 128, 77, 40, 0),       # cross attention: 77 keys, second key tile mostly padding that does not exist in memory
    (1, 4, 100, 100, 80, 64),     # ragged rows (100 of 128), head dim 80, rows wider than heads * D
    (2, 2, 64, 200, 160, 0),      # head dim 160 (three 64-wide k-steps), ragged keys
    # the live-tile variants (tiles and contraction steps that are all padding are not computed) and their boundaries
    (1, 2, 128, 128, 48, 0),      # 3 of 4 d tiles, the augmented columns at 48..55 in the second contraction step
    (1, 2, 128, 192, 56, 8),      # augmented, all 4 tiles
    (1, 2, 64, 64, 64, 0),        # no pad chunk: plain form
    (1, 2, 70, 130, 16, 0),       # one live tile of the 3 computed
    (1, 3, 128, 128, 96, 0),      # 128-wide operands, all tiles (80 < D <= 128)
    (1, 1, 64, 128, 192, 0),      # 192-wide operands, all tiles
    # few key tiles, many query tiles: the dK / dV kernel cuts the queries into chunks (partials in the workspace + a reduce kernel)
    (2, 8, 1024, 77, 40, 0),      # SD cross attention: 4 chunks of 4 query tiles
    (1, 8, 640, 77, 80, 0),       # 2 chunks of 5
    (1, 4, 570, 100, 160, 8),     # 9 query tiles (the last one ragged): chunks of 5 + 4
    # SD's head dims on whole 128-query blocks: forward and backward run on the 32x32x16 kernels (csrc/flash_attn32.hip)
    (1, 8, 512, 384, 40, 8),      # several key / query tiles per block, rows wider than heads * D, XCD-grouped block order
    (1, 4, 256, 128, 40, 0),      # forward entries not a multiple of 8: plain block order
    (3, 8, 128, 640, 40, 0),      # one query block, five key blocks
    (1, 8, 256, 100, 40, 0),      # ragged keys: a second key tile of 36 rows, a key block of 100
    (1, 8, 192, 320, 40, 0),      # queries not a multiple of 128: stays on the 16x16x32 kernels
    (1, 8, 256, 256, 80, 0),      # head dim 80: 256-B LDS rows, six contraction steps, five d tiles, VALU row sums
    (1, 4, 128, 200, 80, 8),      # ... ragged keys (an odd number of key tiles), wide rows
    (2, 8, 256, 256, 160, 0),     # head dim 160: 512-B rows, chunk slots past 16
    (1, 8, 128, 77, 160, 0),      # ... cross attention
]


@pytest.mark.parametrize("B,H,Sq,Sk,D,extra", MERGED_CASES)
@pytest.mark.parametrize("pre", [0, 1])
def test_flash_attention_on_the_projection_layout(dev, B, H, Sq, Sk, D, extra, pre):
    """siss_flash_attn_fwd_merged / _bwd_merged: the same kernels reading q / k / v / dO and writing o / dq / dk / dv in the
    projections' own [B * S, heads * D] layout (head h at columns h * D) -- no head-split / head-merge copies, no padding in
    memory; delta = rowsum(dO o O) formed inside the dQ kernel.  Against torch fp32 attention + autograd, two cotangent sets."""
    from siss_amd import lib
    g = torch.Generator().manual_seed(B * 100 + H + Sq + Sk + D)
    C, ld = H * D, H * D + extra
    sets = 2

    def rows(n, S):                                             # [n * S, ld] bf16 on the device, garbage in the extra columns
        t = torch.randn(n * S, ld, generator=g)
        return t.to(torch.bfloat16).to(dev)
    q, k, v, do = rows(B, Sq), rows(B, Sk), rows(B, Sk), rows(sets * B, Sq)
    heads = lambda t, n, S: t[:, :C].float().cpu().view(n, S, H, D).permute(0, 2, 1, 3)        # [n, H, S, D]
    scale = D ** -0.5
    qr, kr, vr = (heads(t, B, S).clone().requires_grad_(True) for t, S in ((q, Sq), (k, Sk), (v, Sk)))
    if pre:
        # q_prescaled: the tensor handed to the kernels is scale * log2(e) * Q (as the projection's epilogue leaves it); the query
        # the reference differentiates is that bf16 tensor divided by the factor again -- dq is d / d(unscaled q)
        cq = scale * math.log2(math.e)
        q = (q.float() * cq).to(torch.bfloat16)
        qr = (heads(q, B, Sq) / cq).clone().requires_grad_(True)
    s = (qr @ kr.transpose(2, 3)) * scale
    o_ref = torch.softmax(s, dim=-1) @ vr
    dor = heads(do, sets * B, Sq)
    grads = [torch.autograd.grad(o_ref, (qr, kr, vr), dor[i * B:(i + 1) * B], retain_graph=True) for i in range(sets)]

    Sqp = _up(Sq, 64)
    o = torch.full((B * Sq, ld), 7.0, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(B * H, Sqp, device=dev)
    lib.dispatch_counts(reset=True)
    lib.call("siss_flash_attn_fwd_merged", q, ld, k, ld, v, ld, o, ld, lse, B, H, Sq, Sk, D, float(scale), pre)
    torch.cuda.synchronize()
    _close(heads(o, B, Sq), o_ref.detach(), 1e-2, "O")
    if extra:
        assert float((o[:, C:].float() - 7.0).abs().max()) == 0.0, "columns past heads * D are not the kernel's to write"
    lse_ref = torch.logsumexp(s, dim=-1) / math.log(2.0)
    # (at D % 16 == 8 the denominator is summed over the bf16-rounded p the P V product uses -- the ones column --: up to 2^-9 relative)
    assert (lse.view(B, H, Sqp)[:, :, :Sq].cpu() - lse_ref.detach()).abs().max() < 3e-3

    dq = torch.full((sets * B * Sq, ld), 3.0, dtype=torch.bfloat16, device=dev)
    dk, dv = (torch.full((sets * B * Sk, ld), 3.0, dtype=torch.bfloat16, device=dev) for _ in range(2))
    delta = torch.zeros(sets * B * H * Sqp, device=dev)
    lib.call("siss_flash_attn_bwd_merged", q, ld, k, ld, v, ld, o, ld, do, ld, lse, delta, dq, ld, dk, ld, dv, ld,
             sets * B, B, H, Sq, Sk, D, float(scale), pre)
    torch.cuda.synchronize()
    cnt = lib.dispatch_counts(reset=True)
    assert cnt["flash_attn_fwd"] == 1 and cnt["flash_attn_bwd"] == 1
    on32 = D in (40, 80, 160) and Sq % 128 == 0
    assert cnt["flash32_bwd"] == (1 if on32 else 0) and cnt["flash32_fwd"] == (1 if on32 else 0), cnt
    # the dK / dV kernel cuts the queries into chunks when its grid (key blocks x cotangent entries) is small (both families' rule)
    nbh = sets * B * H
    base, qt = (-(-Sk // 128) * nbh, Sq // 64) if on32 else (_up(Sk, 64) // 64 * nbh, _up(Sq, 64) // 64)
    split = base < 512 and qt >= 8 and min((1024 + base - 1) // base, qt // 4) >= 2
    assert cnt["flash_dkdv_qsplit"] == (1 if split else 0), cnt
    d_ref = (dor * heads(o, B, Sq).repeat(sets, 1, 1, 1)).sum(-1)                       # <dO, O> with the bf16 O the kernel read
    _close(delta.view(sets * B, H, Sqp)[:, :, :Sq].cpu(), d_ref, 1e-2, "delta")
    for i in range(sets):
        _close(heads(dq, sets * B, Sq)[i * B:(i + 1) * B], grads[i][0], 2e-2, f"dQ set {i}")
        _close(heads(dk, sets * B, Sk)[i * B:(i + 1) * B], grads[i][1], 2e-2, f"dK set {i}")
        _close(heads(dv, sets * B, Sk)[i * B:(i + 1) * B], grads[i][2], 2e-2, f"dV set {i}")
    if extra:
        for t in (dq, dk, dv):
            assert float((t[:, C:].float() - 3.0).abs().max()) == 0.0
