"""BASELINE.json configs[1] at FULL size (CelebA-HQ 256x256 UNet, 113.7 M parameters, B = 16, bf16, SISS) on the
GPU, checked through size-independent properties (the CPU oracle needs ~10 minutes per step at this size):

  * defensive mixture: x_mix bit-exact vs torch's bf16 add_noise + row select; (1-lambd) iw_x + lambd iw_a = 1
  * forward determinism and zero halos of the padded-NHWC activations
  * the dual-cotangent backward equals two single-cotangent backwards (g_x, g_a) and is linear in the cotangent
  * norm fixing |s g_a| = scaling_norm, the clip coefficient, the AdamW step-1 bound |dtheta| <= lr (1 + wd |theta|)
  * hipGraph replay reproduces the eager step's scalars

Tolerances: wgrad / GroupNorm-parameter gradients accumulate through f32 atomics (reproducible to f32 rounding, not
bitwise): cosine >= 0.9999 and norm ratio within 2e-3 between repeated backwards; linearity additionally carries the
bf16 rounding of every intermediate cotangent: cosine >= 0.999, norm ratio within 2e-2.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

B, HW, LAMBD = 16, 256, 0.5


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from siss_amd.config import UNet2DConfig
    from siss_amd.step import SISSStepper
    from siss_amd.unet import UNetEngine
    dev = torch.device("cuda:0")
    eng = UNetEngine(UNet2DConfig.celebahq256(), dev)
    eng.init_random(seed=42)
    ac = torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, 1000, dtype=torch.float32), 0)
    st = SISSStepper(eng, ac, lr=5e-6, betas=(0.95, 0.999), eps=1e-8, weight_decay=1e-6, scaling_norm=500.0,
                     lambd=LAMBD, train_batch_size=B, mixed_precision="bf16")
    g = torch.Generator(device=dev).manual_seed(42)
    x0 = (torch.rand(B, 3, HW, HW, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    a0 = (torch.rand(1, 3, HW, HW, generator=g, device=dev) * 2 - 1).repeat(B, 1, 1, 1).to(torch.bfloat16)
    noise = torch.randn(B, 3, HW, HW, generator=g, device=dev).to(torch.bfloat16)
    u = torch.rand(B, generator=g, device=dev)
    return dict(dev=dev, eng=eng, st=st, ac=ac.to(dev), x0=x0, a0=a0, noise=noise, u=u, g=g)


def _cos(a, b):
    return float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm() + 1e-300))


@pytest.mark.parametrize("tmode", ["t999", "tuniform"])
def test_mixture_kernel_at_full_size(env, tmode):
    from siss_amd.loss import mixture_fwd
    st, dev = env["st"], env["dev"]
    t = torch.full((B,), 999, dtype=torch.long, device=dev) if tmode == "t999" else \
        torch.randint(0, 1000, (B,), device=dev, generator=env["g"])
    m = mixture_fwd(env["x0"], env["a0"], env["noise"], t, env["u"], st.ac, st.gamma_tab, st.sigma_tab, LAMBD)
    # diffusers add_noise in bf16 (alphas_cumprod cast to the sample dtype first): the oracle
    from oracle import schedule as S
    acc, tc = env["ac"].cpu(), t.cpu()
    keep = S.add_noise(acc, env["x0"].cpu(), env["noise"].cpu(), tc)
    forget = S.add_noise(acc, env["a0"].cpu(), env["noise"].cpu(), tc)
    ref = torch.where((env["u"].cpu() > LAMBD).view(B, 1, 1, 1), keep, forget).to(dev)
    assert torch.equal(m.x_mix, ref)
    inv = (1 - LAMBD) * m.iw_x + LAMBD * m.iw_a
    assert torch.allclose(inv, torch.ones_like(inv), rtol=0, atol=2e-5), inv
    assert torch.isfinite(m.iw_x).all() and torch.isfinite(m.iw_a).all()


def test_forward_is_deterministic_and_keeps_halos_zero(env):
    eng, dev = env["eng"], env["dev"]
    t = torch.full((B,), 999, dtype=torch.long, device=dev)
    x = env["noise"]
    p1 = eng.forward(x, t).clone()
    p2 = eng.forward(x, t).clone()
    assert torch.isfinite(p1).all()
    assert torch.equal(p1, p2)
    acts = list(eng._acts.values())
    assert len(acts) > 100
    for a in acts[:: max(1, len(acts) // 12)]:
        assert a.halo_is_zero(), (a.n, a.h, a.w, a.c)


def test_dual_backward_equals_single_backwards_and_is_linear(env):
    eng, dev = env["eng"], env["dev"]
    t = torch.full((B,), 999, dtype=torch.long, device=dev)
    eng.forward(env["noise"], t)
    g = env["g"]
    cx = torch.randn(B, 3, HW, HW, device=dev, generator=g) * 1e-3
    ca = torch.randn(B, 3, HW, HW, device=dev, generator=g) * 1e-3
    eng.zero_grad()
    eng.backward(torch.cat([cx, ca]).contiguous(), nsets=2)
    gx, ga = eng.ps.grads[0].clone(), eng.ps.grads[1].clone()
    assert torch.isfinite(gx).all() and torch.isfinite(ga).all() and float(gx.norm()) > 0 and float(ga.norm()) > 0
    for c, ref in ((cx, gx), (ca, ga)):
        eng.zero_grad()
        eng.backward(c.contiguous(), nsets=1)
        got = eng.ps.grads[0]
        assert _cos(got, ref) >= 0.9999, _cos(got, ref)
        assert abs(float(got.norm() / ref.norm()) - 1) < 2e-3
        assert float(eng.ps.grads[1].abs().max()) == 0.0          # the other set is untouched
    eng.zero_grad()
    eng.backward((cx + ca).contiguous(), nsets=1)
    got, ref = eng.ps.grads[0], gx + ga
    assert _cos(got, ref) >= 0.999, _cos(got, ref)
    assert abs(float(got.norm() / ref.norm()) - 1) < 2e-2


def test_step_invariants_and_graph_replay(env):
    st, eng, dev = env["st"], env["eng"], env["dev"]
    t = torch.full((B,), 999, dtype=torch.long, device=dev)
    args = (env["x0"], env["a0"], env["noise"], t, env["u"])
    before = eng.ps.flat.clone()
    st.step(*args)
    s = st.stats()
    for k in ("norm_loss_x", "norm_loss_a", "scaling_factor", "pre_clip_norm", "clip_coef"):
        assert s[k] == s[k] and 0 < s[k] < float("inf"), (k, s[k])
    assert abs(s["scaling_factor"] * s["norm_loss_a"] - 500.0) < 0.05           # norm fixing (delete_celeb.py:746)
    assert abs(s["clip_coef"] - min(1.0, 1.0 / (s["pre_clip_norm"] + 1e-6))) <= 1e-6 * s["clip_coef"] + 1e-12
    # pre-clip norm is consistent with the two norms and the triangle inequality: | |g_x| - s|g_a| | <= |g| <= |g_x| + s|g_a|
    lo, hi = abs(s["norm_loss_x"] - 500.0), s["norm_loss_x"] + 500.0
    assert lo * (1 - 1e-3) <= s["pre_clip_norm"] <= hi * (1 + 1e-3)
    # AdamW step 1: |m_hat / (sqrt(v_hat) + eps)| <= 1, so |dtheta| <= lr (1 + wd |theta|)
    d = (eng.ps.flat - before).abs()
    bound = 5e-6 * (1 + 1e-6 * before.abs()) * (1 + 1e-3) + 2.4e-7 * before.abs() + 1e-12      # + 2 ulp of theta (f32 master)
    assert bool((d <= bound).all()), float((d - bound).max())
    assert float(d.max()) > 0
    # the bf16 operand shadow follows the master
    assert torch.equal(eng.ps.shadow, eng.ps.flat.to(torch.bfloat16))
    # importance-weight invariant on the step's own weights
    inv = (1 - LAMBD) * st.last["iw_x"] + LAMBD * st.last["iw_a"]
    assert torch.allclose(inv, torch.ones_like(inv), rtol=0, atol=2e-5)

    # hipGraph: capture one step and replay it from the same parameters as an eager step
    snap = (eng.ps.flat.clone(), st.opt.m.clone(), st.opt.v.clone(), st.opt.scalars.clone())

    def restore():
        eng.ps.flat.copy_(snap[0]); st.opt.m.copy_(snap[1]); st.opt.v.copy_(snap[2]); st.opt.scalars.copy_(snap[3])
        eng.refresh_weights(cast_shadow=True)
    st.step(*args)
    eager = st.stats()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        restore()
        st.step(*args)                       # settle allocations on the capture stream
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            st.step(*args)
    torch.cuda.current_stream().wait_stream(side)
    restore()
    graph.replay()
    torch.cuda.synchronize()
    rep = st.stats()
    for k in ("norm_loss_x", "norm_loss_a", "scaling_factor", "pre_clip_norm"):
        assert abs(rep[k] - eager[k]) <= 2e-3 * abs(eager[k]), (k, rep[k], eager[k])
