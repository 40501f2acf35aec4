"""Fused single-head attention kernels (csrc/attn1h.hip) against torch fp32 softmax(scale Q K^T) V and its autograd gradients on
the same bf16-rounded inputs.  Reference provider: diffusers' Attention (attention_head_dim = None: one head of C channels) in the
AttnDownBlock2D / AttnUpBlock2D / UNetMidBlock2D blocks of `google/ddpm-celebahq-256`, reached from
losses/ddpm_deletion_loss.py:24 and differentiated twice at delete_celeb.py:691,:702 (here: two cotangent sets against one saved
forward).

Layouts as the engine uses them: q / k / v = column windows of ONE compact [B * S][3 D] projection buffer; o and dO in the padded
NHWC activation layout (halo rows must stay untouched) or compact rows; dq / dk / dv = column windows of ONE [nb * S][3 D] buffer.
Shapes: the CelebA-HQ sites (D = 512; S = 256 at 16 x 16 and 64 at 8 x 8), the toy networks' (D = 128), one, two and four key
tiles, a large score spread (max subtraction).  Tolerances: o rel 1e-2 of scale, lse abs 2e-2 (bf16 operands, f32 scores),
dq / dk / dv rel 2e-2 of scale.
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
    return torch.device("cuda:0")


def _bf(x):
    return x.to(torch.bfloat16).float()


def _close(got, ref, rel, what):
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert math.isfinite(err) and err <= rel * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g} (rel {err / scale:.3g} > {rel})"


CASES = [  # B, S, D, W (0: compact o / dO), spread
    (2, 64, 128, 8, 1.0), (3, 256, 128, 16, 1.0), (2, 128, 256, 0, 1.0), (2, 128, 256, 16, 1.0),
    (2, 64, 512, 8, 1.0), (2, 256, 512, 16, 1.0), (2, 256, 512, 16, 6.0), (1, 256, 256, 32, 1.0),
]


@pytest.mark.parametrize("B,S,D,W,spread", CASES)
def test_attn1h_forward_and_dual_backward(dev, B, S, D, W, spread):
    from siss_amd import lib
    from siss_amd.layout import Act
    g = torch.Generator().manual_seed(S + D + B)
    nb = 2 * B
    qkv = _bf(torch.randn(B, S, 3 * D, generator=g) * spread)
    do = _bf(torch.randn(nb, S, D, generator=g))
    scale = D ** -0.5
    q, k, v = (qkv[..., i * D:(i + 1) * D].clone().requires_grad_(True) for i in range(3))
    s = (q @ k.transpose(1, 2)) * scale
    p = torch.softmax(s, dim=-1)
    o_ref = p @ v
    lse_ref = torch.logsumexp(s, dim=-1) / math.log(2.0)
    grads = [torch.autograd.grad(o_ref, (q, k, v), do[z * B:(z + 1) * B], retain_graph=True) for z in range(2)]
    dq_ref, dk_ref, dv_ref = (torch.cat([grads[z][i] for z in range(2)]) for i in range(3))

    qkv_d = qkv.to(torch.bfloat16).to(dev).reshape(B * S, 3 * D).contiguous()
    lse = torch.zeros(B * S, device=dev)
    assert lib.query("siss_attn1h_takes", S, D) == 1
    if W:
        H = S // W
        o_act = Act(B, H, W, D, dev)
        o_act.buf.fill_(3.0)                               # halo / guard rows must come back untouched
        op, ldo = o_act.data, D
        do_act = Act.from_nchw(do.view(nb, H, W, D).permute(0, 3, 1, 2), dev)
        dop = do_act.data
    else:
        o_c = torch.zeros(B * S, D, dtype=torch.bfloat16, device=dev)
        op, ldo = o_c, D
        dop = do.to(torch.bfloat16).to(dev).reshape(nb * S, D).contiguous()
    lib.dispatch_counts(reset=True)
    lib.call("siss_attn1h_fwd", qkv_d, qkv_d[:, D:], qkv_d[:, 2 * D:], 3 * D, op, ldo, W, lse, B, S, D, scale)
    torch.cuda.synchronize()
    if W:
        o_got = o_act.interior().float().reshape(B, S, D).cpu()
        halo = o_act.padded().float()
        assert float((halo[:, 0] - 3).abs().max()) == 0 and float((halo[:, :, 0] - 3).abs().max()) == 0
        assert float((halo[:, -1] - 3).abs().max()) == 0 and float((halo[:, :, -1] - 3).abs().max()) == 0
        # the backward reads o at its interior rows only; give the halo its zeros back as the engine keeps them
        pad = o_act.padded(); pad[:, 0] = 0; pad[:, -1] = 0; pad[:, :, 0] = 0; pad[:, :, -1] = 0
    else:
        o_got = o_c.float().reshape(B, S, D).cpu()
    _close(o_got, o_ref.detach(), 1e-2, "o")
    assert float((lse.cpu().view(B, S) - lse_ref.detach()).abs().max()) <= 2e-2 * max(1.0, spread * spread)

    dqkv = torch.full((nb * S, 3 * D), 7.0, dtype=torch.bfloat16, device=dev)
    delta = torch.zeros(nb * S, device=dev)
    lib.call("siss_attn1h_bwd", qkv_d, qkv_d[:, D:], qkv_d[:, 2 * D:], 3 * D, op, ldo, dop, D, W, lse, delta,
             dqkv, dqkv[:, D:], dqkv[:, 2 * D:], 3 * D, nb, B, S, D, scale)
    torch.cuda.synchronize()
    got = dqkv.float().cpu().view(nb, S, 3 * D)
    delta_ref = (do * torch.cat([o_ref.detach(), o_ref.detach()])).sum(-1)
    _close(delta.cpu().view(nb, S), delta_ref, 2e-2, "delta")
    _close(got[..., :D], dq_ref, 2e-2, "dq")
    _close(got[..., D:2 * D], dk_ref, 2e-2, "dk")
    _close(got[..., 2 * D:], dv_ref, 2e-2, "dv")
    c = lib.dispatch_counts()
    assert c["attn1h_fwd"] == 1 and c["attn1h_bwd"] == 1


def test_attn1h_rejects_unsupported_shapes(dev):
    from siss_amd import lib
    assert lib.query("siss_attn1h_takes", 196, 128) == 0 and lib.query("siss_attn1h_takes", 256, 64) == 0
    assert lib.query("siss_attn1h_takes", 4096, 512) == 0
    t = torch.zeros(64 * 192, dtype=torch.bfloat16, device=dev).view(64, 192)
    lse = torch.zeros(64, device=dev)
    with pytest.raises(RuntimeError):
        lib.call("siss_attn1h_fwd", t, t, t, 192, t, 192, 0, lse, 1, 64, 64, 0.125)        # D = 64: not covered
    with pytest.raises(RuntimeError):
        lib.call("siss_attn1h_fwd", t, t, t, 192, t, 192, 6, lse, 1, 64, 128, 0.1)        # padded rows need W % 4 == 0
