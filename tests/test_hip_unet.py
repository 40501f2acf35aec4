"""End-to-end parity of the HIP UNet engine and the SISS step against the CPU oracle
(oracle/unet.py + oracle/step.py) on the same seeded inputs.

Tolerances (bf16 operands / f32 accumulate vs an fp32 oracle):
  forward pred      max-abs err <= 3e-2 * max|pred|
  g_x, g_a          cosine >= 0.995 per large tensor, ||g|| rel 5e-2
  step scalars      rel 5e-2
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfgs():
    from siss_amd.config import UNet2DConfig
    from oracle.unet import UNetConfig
    kw = dict(sample_size=16, in_channels=3, out_channels=3, block_out_channels=(64, 128),
              down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"),
              layers_per_block=2, attention_head_dim=None, norm_num_groups=32, norm_eps=1e-6,
              downsample_padding=0, flip_sin_to_cos=False, freq_shift=1)
    return UNet2DConfig(**kw), UNetConfig(**kw)


@pytest.fixture(scope="module")
def setup():
    assert torch.cuda.is_available()
    from siss_amd.unet import UNetEngine
    from oracle.unet import OracleUNet2D
    hc, oc = _cfgs()
    eng = UNetEngine(hc, "cuda:0")
    sd = eng.init_random(seed=1)
    net = OracleUNet2D(oc)
    net.load_state_dict(sd)
    return eng, net, sd


def _cos(a, b):
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))


def test_param_roundtrip(setup):
    eng, net, sd = setup
    back = eng.state_dict()
    assert set(back) == set(sd)
    for k in sd:
        torch.testing.assert_close(back[k], sd[k].float(), rtol=0, atol=0)


def test_forward_matches_oracle(setup):
    eng, net, _ = setup
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 3, 16, 16, generator=g)
    t = torch.tensor([999, 500, 3, 999])
    with torch.no_grad():
        ref = net(x, t)[0]
    got = eng.forward(x.cuda(), t.cuda()).cpu()
    err = (got - ref).abs().max().item()
    assert err <= 3e-2 * ref.abs().max().item(), (err, ref.abs().max().item())


def test_dual_backward_matches_oracle(setup):
    eng, net, _ = setup
    g = torch.Generator().manual_seed(1)
    B = 4
    x = torch.randn(B, 3, 16, 16, generator=g)
    t = torch.tensor([999, 10, 700, 999])
    cx = torch.randn(B, 3, 16, 16, generator=g)
    ca = torch.randn(B, 3, 16, 16, generator=g)
    names, params = zip(*net.named_parameters())
    out_ref = net(x, t)[0]                                              # one oracle forward, a backward per cotangent
    refs = [dict(zip(names, torch.autograd.grad(out_ref, params, c, retain_graph=(i == 0)))) for i, c in enumerate((cx, ca))]
    eng.forward(x.cuda(), t.cuda())
    eng.zero_grad()
    eng.backward(torch.cat([cx, ca]).cuda().contiguous(), nsets=2)
    torch.cuda.synchronize()
    for s in range(2):
        got = eng.ps.grads_ref(s)
        tot_r = torch.sqrt(sum(v.square().sum() for v in refs[s].values()))
        tot_g = torch.sqrt(sum(v.square().sum() for v in got.values()))
        assert abs(float(tot_g / tot_r) - 1) < 5e-2, (s, float(tot_g), float(tot_r))
        bad = []
        for n, r in refs[s].items():
            c = _cos(got[n].float(), r)
            if r.norm() > 1e-3 * tot_r and c < 0.99:
                bad.append((n, c, float(got[n].norm()), float(r.norm())))
        assert not bad, bad[:10]


def test_siss_step_matches_oracle(setup):
    """Two optimizer steps, fp32 I/O mode (the bf16 I/O mode is covered in test_hip_kernels)."""
    import copy
    from siss_amd.step import SISSStepper
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    eng, net0, sd = setup
    eng.load_state_dict(sd)
    net = copy.deepcopy(net0)
    net.load_state_dict(sd)
    ac = S.alphas_cumprod()
    L = OracleDeletionLoss(*S.gamma_sigma(ac))
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    st = SISSStepper(eng, ac, lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8, scaling_norm=5.0,
                     lambd=0.5, train_batch_size=4, mixed_precision=None)
    g = torch.Generator().manual_seed(7)
    from parity_util import assert_update_direction
    for step in range(2):
        before_ref = {n: p.detach().clone() for n, p in net.named_parameters()}
        before_got = eng.state_dict()
        x0 = torch.rand(4, 3, 16, 16, generator=g) * 2 - 1
        a0 = (torch.rand(1, 3, 16, 16, generator=g) * 2 - 1).repeat(4, 1, 1, 1)
        noise = torch.randn(4, 3, 16, 16, generator=g)
        t = torch.full((4,), 999, dtype=torch.long)
        u = torch.tensor([0.9, 0.2, 0.7, 0.4])
        ref, gx, ga, gfin = unlearning_step(net, opt, L, "importance_sampling_with_mixture", ac,
                                            [dict(x0=x0, a0=a0, noise=noise, t=t, u=u)], train_batch_size=4,
                                            scaling_norm=5.0, loss_params={"lambd": 0.5})
        st.step(x0, a0, noise, t.cuda(), u)
        got = st.stats()
        for k_ref, k_got in (("norm_loss_x", "norm_loss_x"), ("norm_loss_a", "norm_loss_a"),
                             ("scaling_factor", "scaling_factor"), ("pre_clip_norm", "pre_clip_norm")):
            r, v = getattr(ref, k_ref), got[k_got]
            assert abs(v - r) <= 5e-2 * abs(r), (step, k_ref, v, r)
        # this step's parameter update, per side from its own starting point: masked direction cosine
        new = eng.state_dict()
        d_got = {n: before_ref[n] + (new[n] - before_got[n]) for n in new}
        assert_update_direction(before_ref, dict(net.named_parameters()), d_got, gfin, f"step {step}")


def _fresh(setup):
    import copy
    eng, net0, sd = setup
    eng.load_state_dict(sd)
    net = copy.deepcopy(net0)
    net.load_state_dict(sd)
    return eng, net, sd


def _batch(g, B=4):
    x0 = torch.rand(B, 3, 16, 16, generator=g) * 2 - 1
    a0 = (torch.rand(1, 3, 16, 16, generator=g) * 2 - 1).repeat(B, 1, 1, 1)
    noise = torch.randn(B, 3, 16, 16, generator=g)
    return dict(x0=x0, a0=a0, noise=noise, t=torch.full((B,), 999, dtype=torch.long), u=torch.rand(B, generator=g))


def _check_scalars(ref, got, tol=5e-2):
    for k in ("norm_loss_x", "norm_loss_a", "scaling_factor", "pre_clip_norm"):
        r, v = getattr(ref, k), got[k]
        assert abs(v - r) <= tol * abs(r), (k, v, r)


def test_no_is_fast_path_matches_oracle(setup):
    """SISS-No-IS (double_forward_with_neg_del, ddpm_deletion_loss.py:60-67): ONE batch-2B forward + dual backward."""
    from siss_amd.step import SISSStepper, NO_IS
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    eng, net, sd = _fresh(setup)
    ac = S.alphas_cumprod()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6)
    st = SISSStepper(eng, ac, lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6, scaling_norm=5.0,
                     train_batch_size=4, loss_fn=NO_IS, mixed_precision=None)
    mb = _batch(torch.Generator().manual_seed(21))
    ref, *_ = unlearning_step(net, opt, OracleDeletionLoss(*S.gamma_sigma(ac)), NO_IS, ac, [mb],
                              train_batch_size=4, scaling_norm=5.0)
    st.step(mb["x0"], mb["a0"], mb["noise"], mb["t"].cuda(), mb["u"])
    _check_scalars(ref, st.stats())


@pytest.mark.parametrize("loss_fn", ["erasediff", "simple_neg_del", "naive_del", "subscore_bernoulli"])
def test_baseline_losses_fast_path_matches_oracle(setup, loss_fn):
    """The other objectives of the class surface (ddpm_deletion_loss.py:70-122) on the fused stepper: EraseDiff
    (two forwards, U[0,1) target, s = -max(eta - <g_x,g_a>/|g_a|^2, 0)), NegGrad / naive (one backward, no split),
    Bernoulli sub-score (row selection as per-sample weights)."""
    from siss_amd.step import SISSStepper
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    eng, net, sd = _fresh(setup)
    ac = S.alphas_cumprod()
    okw = dict(lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6)
    opt = torch.optim.AdamW(net.parameters(), **okw)
    mb = _batch(torch.Generator().manual_seed(41))
    name = loss_fn
    lp = {"lambd": 0.5} if name == "subscore_bernoulli" else ({"superfactor": 3.0} if name == "simple_neg_del" else {})
    kw = dict(scaling_norm=5.0) if name != "erasediff" else dict(eta=1e-2)
    st = SISSStepper(eng, ac, lambd=0.5, train_batch_size=4, loss_fn=name, mixed_precision=None, superfactor=3.0,
                     inf_guard=True, **okw, **kw)
    torch.manual_seed(99)                                     # erasediff: rand_like is the first draw in the oracle
    ref, _, _, gfin = unlearning_step(net, opt, OracleDeletionLoss(*S.gamma_sigma(ac)), name, ac, [mb], train_batch_size=4,
                                      scaling_norm=5.0, eta=1e-2, loss_params=lp, inf_guard=True)
    torch.manual_seed(99)
    target = torch.rand(mb["noise"].shape) if name == "erasediff" else None
    st.step(mb["x0"], mb["a0"], mb["noise"], mb["t"].cuda(), mb["u"], erase_target=target)
    got = st.stats()
    if name in ("simple_neg_del", "naive_del"):
        assert abs(got["pre_clip_norm"] - ref.pre_clip_norm) <= 5e-2 * ref.pre_clip_norm
    else:
        _check_scalars(ref, got)
    from parity_util import assert_update_direction
    assert_update_direction(sd, dict(net.named_parameters()), eng.state_dict(), gfin, name)


def test_gradient_accumulation_two_micro_batches(setup):
    """GA = 2: gradients of both micro-batches accumulate in the flat [g_x ; g_a] buffer (delete_celeb.py:705-711)."""
    from siss_amd.step import SISSStepper
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    eng, net, sd = _fresh(setup)
    ac = S.alphas_cumprod()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6)
    st = SISSStepper(eng, ac, lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6, scaling_norm=5.0, lambd=0.5,
                     train_batch_size=4, grad_accum=2, mixed_precision=None)
    g = torch.Generator().manual_seed(33)
    mbs = [_batch(g), _batch(g)]
    ref, *_ = unlearning_step(net, opt, OracleDeletionLoss(*S.gamma_sigma(ac)), "importance_sampling_with_mixture", ac,
                              mbs, train_batch_size=4, scaling_norm=5.0, loss_params={"lambd": 0.5})
    for mb in mbs:
        st.micro_step(mb["x0"], mb["a0"], mb["noise"], mb["t"].cuda(), mb["u"])
    _check_scalars(ref, st.stats())


def test_bf16_io_mode_and_graph_replay(setup):
    """mixed_precision=bf16 (images/noise cast to bf16 first, delete_celeb.py:561-581) against the oracle fed the
    same bf16-rounded inputs; then the captured hipGraph replays to the same scalars as the eager step."""
    from siss_amd.step import SISSStepper
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    eng, net, sd = _fresh(setup)
    ac = S.alphas_cumprod()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6)
    mb = _batch(torch.Generator().manual_seed(5))
    rb = {k: (v.to(torch.bfloat16).float() if v.dtype == torch.float32 and k != "u" else v) for k, v in mb.items()}
    ref, *_ = unlearning_step(net, opt, OracleDeletionLoss(*S.gamma_sigma(ac)), "importance_sampling_with_mixture", ac,
                              [rb], train_batch_size=4, scaling_norm=5.0, loss_params={"lambd": 0.5})
    st = SISSStepper(eng, ac, lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6, scaling_norm=5.0, lambd=0.5,
                     train_batch_size=4, mixed_precision="bf16")
    dev = eng.device
    x0, a0, noise = (mb[k].to(dev).to(torch.bfloat16) for k in ("x0", "a0", "noise"))
    t, u = mb["t"].to(dev), mb["u"].to(dev)
    st.step(x0, a0, noise, t, u)
    eager = st.stats()
    _check_scalars(ref, eager)               # rel 5e-2 also with bf16 noising / bf16 targets
    # graph: same inputs, parameters restored -> identical schedule replayed from a hipGraph
    eng.load_state_dict(sd)
    st2 = SISSStepper(eng, ac, lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6, scaling_norm=5.0, lambd=0.5,
                      train_batch_size=4, mixed_precision="bf16")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        st2.step(x0, a0, noise, t, u)                 # warm-up on the capture stream (allocations)
        eng.load_state_dict(sd)
        st2.opt.m.zero_(); st2.opt.v.zero_(); st2.opt.scalars.zero_()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            st2.step(x0, a0, noise, t, u)
    torch.cuda.current_stream().wait_stream(side)
    eng.load_state_dict(sd)
    st2.opt.m.zero_(); st2.opt.v.zero_(); st2.opt.scalars.zero_()
    graph.replay()
    torch.cuda.synchronize()
    rep = st2.stats()
    for k in ("norm_loss_x", "norm_loss_a", "scaling_factor", "pre_clip_norm"):
        assert abs(rep[k] - eager[k]) <= 2e-2 * abs(eager[k]), (k, rep[k], eager[k])
    assert rep["step"] == 1


def test_side_stream_weight_gradients_equal_the_main_stream_schedule(setup):
    """UNetEngine.wgrad_side: the weight gradients queued before the low-resolution middle of the backward pass run on a SIDE stream
    as capped grouped launches (siss_gemm_tn_grouped_capped: 8 workgroups walking all blocks) beside it and join at the end.  Same
    products, same operands: both gradient sets equal the one-stream schedule's to f32 rounding (float atomics) -- eagerly and
    replayed from a hipGraph whose capture forks to the side stream and joins it.  (side_max_px lowered so that the toy network,
    8 x 8 / 16 x 16, has a 'low-resolution middle' at all; group_max lowered so that several later batches FOLLOW the first one to
    the side stream: side_follow.)"""
    eng, _, sd = _fresh(setup)
    g = torch.Generator().manual_seed(21)
    B = 4
    x = torch.randn(B, 3, 16, 16, generator=g).cuda()
    t = torch.tensor([999, 10, 700, 999]).cuda()
    cot = torch.randn(2 * B, 3, 16, 16, generator=g).cuda().contiguous()

    def run():
        eng.forward(x, t)
        eng.zero_grad()
        eng.backward(cot, nsets=2)
        torch.cuda.synchronize()
        return eng.ps.grads.clone()
    saved = (eng.wgrad_side, eng.side_max_px, eng.side_blocks, eng.group_max, eng.side_follow)
    try:
        eng.wgrad_side = False
        ref = run()
        eng.wgrad_side, eng.side_max_px, eng.side_blocks, eng.group_max, eng.side_follow = True, 64, 8, 3, 1
        got = run()
        assert eng._side is not None and eng._side_mark is not None, "the side stream must have been used"
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 1e-5 * scale
        # ... and under capture: fork to the side stream, join before the capture ends
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            run()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=cap):
                eng.forward(x, t)
                eng.zero_grad()
                eng.backward(cot, nsets=2)
        torch.cuda.current_stream().wait_stream(cap)
        eng.ps.grads.fill_(7.0)
        graph.replay()
        torch.cuda.synchronize()
        assert float((eng.ps.grads - ref).abs().max()) <= 1e-5 * scale
    finally:
        eng.wgrad_side, eng.side_max_px, eng.side_blocks, eng.group_max, eng.side_follow = saved


def test_mnist_tshirt_config_step_matches_oracle():
    """BASELINE config 1 on the HIP path: MNIST 28x28 UNet (64/128/256 channels, 16/32-head attention with
    head_dim 8, downsample_padding=1, flip_sin_to_cos), t ~ U{0..999}, inf guard (delete_tshirt.py)."""
    from siss_amd.config import UNet2DConfig
    from siss_amd.step import SISSStepper
    from siss_amd.unet import UNetEngine
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    from oracle.unet import OracleUNet2D, UNetConfig
    eng = UNetEngine(UNet2DConfig.mnist_tshirt(), "cuda:0")
    sd = eng.init_random(seed=11)
    net = OracleUNet2D(UNetConfig.mnist_tshirt())
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(46)
    B = 4
    x = torch.randn(B, 1, 28, 28, generator=g)
    t = torch.tensor([999, 500, 20, 731])
    with torch.no_grad():
        ref = net(x, t)[0]
    got = eng.forward(x.cuda(), t.cuda()).cpu()
    assert (got - ref).abs().max() <= 3e-2 * ref.abs().max()
    ac = S.alphas_cumprod()
    opt = torch.optim.AdamW(net.parameters(), lr=5e-5, betas=(0.95, 0.999), weight_decay=1e-6)
    st = SISSStepper(eng, ac, lr=5e-5, betas=(0.95, 0.999), weight_decay=1e-6, scaling_norm=5.0, lambd=0.5,
                     train_batch_size=B, inf_guard=True, mixed_precision=None)
    mb = dict(x0=torch.rand(B, 1, 28, 28, generator=g) * 2 - 1, a0=torch.rand(B, 1, 28, 28, generator=g) * 2 - 1,
              noise=torch.randn(B, 1, 28, 28, generator=g), t=torch.tensor([999, 800, 950, 600]),
              u=torch.tensor([0.9, 0.1, 0.6, 0.3]))
    r, *_ = unlearning_step(net, opt, OracleDeletionLoss(*S.gamma_sigma(ac)), "importance_sampling_with_mixture", ac,
                            [mb], train_batch_size=B, scaling_norm=5.0, loss_params={"lambd": 0.5}, inf_guard=True)
    st.step(mb["x0"], mb["a0"], mb["noise"], mb["t"].cuda(), mb["u"])
    _check_scalars(r, st.stats())


def test_mnist_tshirt_config_at_yaml_batch_64():
    """config/delete_tshirt.yaml's own train_batch_size (64; BASELINE quotes 32): 128 cotangent rows through the
    batched time-embedding backward and the per-(sample, head) attention kernels, against the oracle step."""
    from siss_amd.config import UNet2DConfig
    from siss_amd.step import SISSStepper
    from siss_amd.unet import UNetEngine
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    from oracle.unet import OracleUNet2D, UNetConfig
    eng = UNetEngine(UNet2DConfig.mnist_tshirt(), "cuda:0")
    sd = eng.init_random(seed=12)
    net = OracleUNet2D(UNetConfig.mnist_tshirt())
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(47)
    B = 64
    ac = S.alphas_cumprod()
    opt = torch.optim.AdamW(net.parameters(), lr=5e-5, betas=(0.95, 0.999), weight_decay=1e-6)
    st = SISSStepper(eng, ac, lr=5e-5, betas=(0.95, 0.999), weight_decay=1e-6, scaling_norm=5.0, lambd=0.5,
                     train_batch_size=B, inf_guard=True, mixed_precision=None)
    mb = dict(x0=torch.rand(B, 1, 28, 28, generator=g) * 2 - 1, a0=torch.rand(B, 1, 28, 28, generator=g) * 2 - 1,
              noise=torch.randn(B, 1, 28, 28, generator=g), t=torch.randint(900, 1000, (B,), generator=g),
              u=torch.rand(B, generator=g))
    r, *_ = unlearning_step(net, opt, OracleDeletionLoss(*S.gamma_sigma(ac)), "importance_sampling_with_mixture", ac,
                            [mb], train_batch_size=B, scaling_norm=5.0, loss_params={"lambd": 0.5}, inf_guard=True)
    st.step(mb["x0"], mb["a0"], mb["noise"], mb["t"].cuda(), mb["u"])
    _check_scalars(r, st.stats())


@pytest.mark.parametrize("mode", ["bf16", "f32"])
def test_mnist_tshirt_config_at_baseline_batch_32(mode):
    """BASELINE configs[0] at ITS batch: delete_tshirt.yaml's MNIST UNet, SISS, bs = 32, t ~ U{0..999} (delete_tshirt.py:504-557),
    inf guard -- in the two REAL modes of the HIP path: bf16 engine + bf16 I/O (`mixed_precision: bf16`; oracle fed the same
    bf16-rounded inputs, step scalars rel 5e-2) and the f32 engine (`mixed_precision: null`, what the reference's CPU path of this
    config computes in; step scalars rel 2e-4, masked update cosine >= 0.9999)."""
    from siss_amd.config import UNet2DConfig
    from siss_amd.step import SISSStepper
    from siss_amd.unet import UNetEngine
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    from oracle.unet import OracleUNet2D, UNetConfig
    from parity_util import check_scalars, final_gradient_cosine, masked_update_cosine
    f32 = mode == "f32"
    eng = UNetEngine(UNet2DConfig.mnist_tshirt(), "cuda:0", dtype=torch.float32 if f32 else torch.bfloat16)
    sd = eng.init_random(seed=13)
    net = OracleUNet2D(UNetConfig.mnist_tshirt())
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(46)               # delete_tshirt.yaml:11 random_seed
    B = 32
    ac = S.alphas_cumprod()
    okw = dict(lr=5e-5, betas=(0.95, 0.999), weight_decay=1e-6)
    opt = torch.optim.AdamW(net.parameters(), **okw)
    st = SISSStepper(eng, ac, scaling_norm=5.0, lambd=0.5, train_batch_size=B, inf_guard=True,
                     mixed_precision=None if f32 else "bf16", **okw)
    mb = dict(x0=torch.rand(B, 1, 28, 28, generator=g) * 2 - 1, a0=torch.rand(B, 1, 28, 28, generator=g) * 2 - 1,
              noise=torch.randn(B, 1, 28, 28, generator=g), t=torch.randint(0, 1000, (B,), generator=g),
              u=torch.rand(B, generator=g))
    rb = mb if f32 else {k: (v.to(torch.bfloat16).float() if v.dtype == torch.float32 and k != "u" else v) for k, v in mb.items()}
    r, _, _, gfin = unlearning_step(net, opt, OracleDeletionLoss(*S.gamma_sigma(ac)), "importance_sampling_with_mixture", ac,
                                    [rb], train_batch_size=B, scaling_norm=5.0, loss_params={"lambd": 0.5}, inf_guard=True)
    cast = (lambda v: v) if f32 else (lambda v: v.to(torch.bfloat16))
    st.step(cast(mb["x0"]), cast(mb["a0"]), cast(mb["noise"]), mb["t"].cuda(), mb["u"])
    got = st.stats()
    check_scalars(r, got, tol=2e-4 if f32 else 5e-2)
    cos, frac = masked_update_cosine(sd, dict(net.named_parameters()), eng.state_dict(), gfin)
    print(f"\nconfigs[0] B=32 {mode}: " + ", ".join(f"{k} {got[k]:.6g} / {getattr(r, k):.6g}" for k in
          ("norm_loss_x", "norm_loss_a", "scaling_factor", "pre_clip_norm")) + f"; masked update cosine {cos:.6f}")
    # bf16: the masked update direction is SIGN-like under AdamW's first step and this batch draws t down to 0, where a few samples
    # carry 1 / sigma-weighted gradients and most elements' gradients sit at the bf16 noise floor: measured 0.919 (the t = 999 steps of
    # the other tests: >= 0.99).  The gradient-weighted cosine of the final gradient g_x - s g_a is the bound that is held at 0.99.
    gcos = final_gradient_cosine(eng, gfin, got["scaling_factor"])
    print(f"final-gradient cosine {gcos:.6f}")
    assert gcos >= (0.999999 if f32 else 0.99), gcos
    assert cos >= (0.9999 if f32 else 0.85) and frac > 0.3, (cos, frac)


@pytest.mark.parametrize("lambd,B", [(0.0, 4), (1.0, 4), (0.5, 1), (0.3, 3)])
def test_siss_step_edge_cases_match_oracle(setup, lambd, B):
    """Edges of the defensive mixture: lambd = 0 (every row keeps; iw_x = 1, the forget term still carries
    iw_a = e^d), lambd = 1 (every row forgets), a single-sample batch, an odd batch with an uneven mask."""
    from siss_amd.step import SISSStepper
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    eng, net, sd = _fresh(setup)
    ac = S.alphas_cumprod()
    okw = dict(lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6)
    opt = torch.optim.AdamW(net.parameters(), **okw)
    st = SISSStepper(eng, ac, scaling_norm=5.0, lambd=lambd, train_batch_size=B, mixed_precision=None, **okw)
    mb = _batch(torch.Generator().manual_seed(int(lambd * 10) + B), B=B)
    ref, *_ = unlearning_step(net, opt, OracleDeletionLoss(*S.gamma_sigma(ac)), "importance_sampling_with_mixture", ac,
                              [mb], train_batch_size=B, scaling_norm=5.0, loss_params={"lambd": lambd})
    st.step(mb["x0"], mb["a0"], mb["noise"], mb["t"].cuda(), mb["u"])
    got = st.stats()
    _check_scalars(ref, got)
    inv = (1 - lambd) * st.last["iw_x"] + lambd * st.last["iw_a"]
    assert torch.allclose(inv.cpu(), torch.ones(B), atol=1e-4)


def test_direct_concat_writes_equal_the_copying_concat(monkeypatch):
    """Up-path convs write straight into the head columns of the concat buffer (ActView, ldc = C + C_skip), conv-produced
    skips into its tail columns (every down-path reader takes the row stride): forward and gradients must be BITWISE
    what the copying concat gives (same kernels, same operand values, different strides)."""
    from siss_amd import lib
    from siss_amd.layout import ActView
    from siss_amd.unet import UNetEngine
    hc, _ = _cfgs()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 16, 16, generator=g).cuda()
    t = torch.tensor([999, 500, 3, 999]).cuda()
    cot = (torch.randn(8, 3, 16, 16, generator=g) * 1e-2).cuda()
    outs = {}
    for mode in ("1", "0"):
        eng = UNetEngine(hc, "cuda:0")
        eng.direct_cat = mode == "1"
        eng.init_random(seed=1)
        calls = []
        orig = lib.call
        monkeypatch.setattr(lib, "call", lambda name, *a, _o=orig, _c=calls: (_c.append(name), _o(name, *a))[1])
        pred = eng.forward(x, t).clone()
        monkeypatch.setattr(lib, "call", orig)
        eng.zero_grad()
        eng.backward(cot, nsets=2)
        torch.cuda.synchronize()
        outs[mode] = (pred, eng.ps.grads.clone(), calls)
    tails, fulls = outs["1"][2].count("siss_concat_tail"), outs["1"][2].count("siss_concat")
    # conv-produced skips live in the tail columns of their concat buffer from the start and conv-produced partners are
    # written into its head columns: those concats copy NOTHING; attention outputs (no view) still copy
    total = outs["0"][2].count("siss_concat")
    assert total == 6 and outs["0"][2].count("siss_concat_tail") == 0 and tails + fulls < total, (tails, fulls, total)
    assert torch.equal(outs["1"][0], outs["0"][0])
    ga, gb = outs["1"][1], outs["0"][1]
    # wgrad accumulates through f32 atomics: equal up to summation order
    assert float((ga - gb).norm() / gb.norm()) < 1e-5
    v = ActView(eng._act("probe", 2, 4, 4, 24), 0, 16)
    assert v.ld == 24 and tuple(v.data.shape) == (v.rows, 16) and v.data.data_ptr() == v.base.data.data_ptr()


def test_depth_to_space_in_the_dgrad_epilogue_equals_the_separate_pass():
    """Downsample2D backward: the four plane GEMMs write their pixels straight to their place in dx (siss_gemm_nt_d2s, adding the
    cotangent x already carries) instead of a dz tensor + siss_depth_to_space.  Same products, same bf16 roundings in the same
    order -> the gradients agree up to the wgrad atomics' summation order (delete_celeb.py:691,:702 differentiate through
    diffusers' Downsample2D)."""
    from siss_amd import lib
    from siss_amd.unet import UNetEngine
    hc, _ = _cfgs()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 3, 16, 16, generator=g).cuda()
    t = torch.tensor([999, 500, 3, 999]).cuda()
    cot = (torch.randn(8, 3, 16, 16, generator=g) * 1e-2).cuda()
    outs = {}
    for fused in (True, False, "one launch"):
        eng = UNetEngine(hc, "cuda:0")
        eng.init_random(seed=1)
        eng.d2s_epilogue, eng.phase_launch = bool(fused), fused == "one launch"
        calls = []
        orig = lib.call
        lib.call = lambda name, *a, _o=orig, _c=calls, **k: (_c.append(name), _o(name, *a, **k))[1]
        try:
            eng.forward(x, t)
            eng.zero_grad()
            eng.backward(cot, nsets=2)
            torch.cuda.synchronize()
        finally:
            lib.call = orig
        outs[fused] = (eng.ps.grads.clone(), calls)
    assert outs[True][1].count("siss_gemm_nt_d2s") == 4 and outs[True][1].count("siss_depth_to_space") == 0
    assert outs[False][1].count("siss_gemm_nt_d2s") == 0 and outs[False][1].count("siss_depth_to_space") == 1
    # ... and the four plane products as ONE launch (siss_gemm_nt_d2s_phases: UNetEngine.phase_launch, the default)
    assert outs["one launch"][1].count("siss_gemm_nt_d2s_phases") == 1 and outs["one launch"][1].count("siss_gemm_nt_d2s") == 0
    ga, gb = outs[True][0], outs[False][0]
    assert float((ga - gb).norm() / gb.norm()) < 1e-5
    assert float((outs["one launch"][0] - gb).norm() / gb.norm()) < 1e-5


# ---------------------------------------------------------------------------------------------------------------------
# a-6: the statistics block the reference logs every micro-step (delete_celeb.py:626-663) -- every key the reference
# emits for the objective, against the oracle's literal restatement (oracle/step.py::batch_stats) on the same step.
# Tolerances: loss statistics rel 5e-2 (bf16 compute vs fp32), importance weights rel 2e-3, std unbiased (nan for one row).
# ---------------------------------------------------------------------------------------------------------------------
def _check_block(ref_block, got, what):
    import math
    assert ref_block, what
    for k, r in ref_block.items():
        assert k in got, (what, "missing", k, sorted(got))
        v = got[k]
        if isinstance(r, float) and math.isnan(r):
            assert math.isnan(v), (what, k, v, r)
            continue
        tol = 2e-3 if k.startswith("importance_weight") else (0.0 if k == "superfactor" else 5e-2)
        # (std over the FEW selected rows of a block -- 2 forget rows here -- is a difference of per-sample means: its error is the
        # bf16 noise of those means, ~1e-3 of the mean, whatever the std's own size; bound = 5e-2 x max(|std|, 2e-2 |mean|))
        scale = abs(r) if not k.endswith("/std") else max(abs(r), 2e-2 * abs(ref_block[k[:-4] + "/mean"]))
        assert abs(v - r) <= tol * scale + 1e-12, (what, k, v, r)
    extra = {k for k in got if "/" in k} - set(ref_block)
    assert not extra, (what, "keys the reference does not log", extra)


@pytest.mark.parametrize("loss_fn", ["importance_sampling_with_mixture", "double_forward_with_neg_del", "erasediff",
                                     "simple_neg_del", "naive_del", "subscore_bernoulli"])
def test_stats_block_matches_the_reference_block(setup, loss_fn):
    from siss_amd.step import SISSStepper
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    eng, net, sd = _fresh(setup)
    ac = S.alphas_cumprod()
    okw = dict(lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6)
    opt = torch.optim.AdamW(net.parameters(), **okw)
    B = 5
    mb = _batch(torch.Generator().manual_seed(77), B=B)
    mb["t"] = torch.tensor([999, 940, 999, 870, 999])          # unequal per-sample losses / weights
    mb["u"] = torch.tensor([0.9, 0.1, 0.7, 0.2, 0.6])
    lp = {"lambd": 0.5} if loss_fn in ("importance_sampling_with_mixture", "subscore_bernoulli") else (
        {"superfactor": 3.0} if loss_fn == "simple_neg_del" else {})
    kw = dict(scaling_norm=5.0) if loss_fn != "erasediff" else dict(eta=1e-2)
    st = SISSStepper(eng, ac, lambd=0.5, train_batch_size=B, loss_fn=loss_fn, mixed_precision=None, superfactor=3.0,
                     superfactor_decay=0.5, inf_guard=True, **okw, **kw)
    torch.manual_seed(99)
    ref, *_ = unlearning_step(net, opt, OracleDeletionLoss(*S.gamma_sigma(ac)), loss_fn, ac, [mb], train_batch_size=B,
                              scaling_norm=5.0, eta=1e-2, loss_params=lp, inf_guard=True, superfactor_decay=0.5)
    torch.manual_seed(99)
    target = torch.rand(mb["noise"].shape) if loss_fn == "erasediff" else None
    st.step(mb["x0"], mb["a0"], mb["noise"], mb["t"].cuda(), mb["u"], erase_target=target)
    _check_block(ref.batch_stats[0], st.stats(), loss_fn)
    if loss_fn == "simple_neg_del":                              # superfactor *= superfactor_decay (:658-662)
        assert st.superfactor == 1.5


@pytest.mark.parametrize("u", [[0.9, 0.8, 0.7], [0.1, 0.2, 0.3], [0.9, 0.1, 0.2]])
def test_subscore_stats_row_selection_and_zero_size_guards(setup, u):
    """subscore_bernoulli logs statistics over the SELECTED rows (loss[mask] / (1 - lambd), loss[~mask]); with no keep
    rows both entries are one zero, with no forget rows loss_a is (ddpm_deletion_loss.py:113-120); one selected row
    has std nan."""
    from siss_amd.step import SISSStepper
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    eng, net, sd = _fresh(setup)
    ac = S.alphas_cumprod()
    okw = dict(lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6)
    opt = torch.optim.AdamW(net.parameters(), **okw)
    mb = _batch(torch.Generator().manual_seed(78), B=3)
    mb["u"] = torch.tensor(u)
    st = SISSStepper(eng, ac, lambd=0.5, train_batch_size=3, loss_fn="subscore_bernoulli", mixed_precision=None,
                     scaling_norm=5.0, inf_guard=True, **okw)
    # the block is computed from the loss call alone (:622-663), before any backward: with NO keep rows the reference's
    # step goes on to crash at :694 (param.grad is None for a loss that is a fresh zeros leaf), so only the block is compared
    from oracle.step import batch_stats, prep_inputs
    keep, forget = prep_inputs(ac, mb["x0"], mb["a0"], mb["noise"], mb["t"])
    with torch.no_grad():
        items = OracleDeletionLoss(*S.gamma_sigma(ac)).subscore_bernoulli(net, mb["t"], mb["noise"], {}, keep, forget,
                                                                          lambd=0.5, u=mb["u"])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref_block = batch_stats(items, {"lambd": 0.5})
    st.step(mb["x0"], mb["a0"], mb["noise"], mb["t"].cuda(), mb["u"])
    _check_block(ref_block, st.stats(), f"subscore u={u}")


def test_folded_shortcut_equals_the_separate_1x1_product():
    """A resnet whose input and output widths differ adds conv_shortcut(x) to conv2(h) (diffusers ResnetBlock2D; reached from
    losses/ddpm_deletion_loss.py:24).  Where the persistent 3x3 kernel takes the conv2 product, the 1x1 shortcut rides in it
    as extra K-groups (siss_conv3x3_sc) instead of a product of its own + a residual read.  Same math, one bf16 rounding
    less (the shortcut's output is no longer rounded before the add): prediction within 1e-2 of scale, gradients cosine
    >= 0.999 per set against the two-launch form on a 128 x 128 network whose up path folds."""
    from siss_amd import lib
    from siss_amd.config import UNet2DConfig
    from siss_amd.unet import UNetEngine
    kw = dict(sample_size=128, in_channels=3, out_channels=3, block_out_channels=(128, 128, 256),
              down_block_types=("DownBlock2D", "DownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "UpBlock2D", "UpBlock2D"),
              layers_per_block=1, attention_head_dim=None, norm_num_groups=32, norm_eps=1e-6,
              downsample_padding=0, flip_sin_to_cos=False, freq_shift=1)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 3, 128, 128, generator=g).cuda()
    t = torch.tensor([999, 400]).cuda()
    cot = (torch.randn(4, 3, 128, 128, generator=g) * 1e-2).cuda()
    outs = {}
    for fold in (True, False):
        eng = UNetEngine(UNet2DConfig(**kw), "cuda:0")
        eng.init_random(seed=4)
        eng.fold_shortcut = fold
        calls = []
        orig = lib.call
        lib.call = lambda name, *a, _o=orig, _c=calls, **k: (_c.append(name), _o(name, *a, **k))[1]
        try:
            pred = eng.forward(x, t).clone()
            eng.zero_grad()
            eng.backward(cot, nsets=2)
            torch.cuda.synchronize()
        finally:
            lib.call = orig
        outs[fold] = (pred, eng.ps.grads.clone(), calls)
    # the two 128 x 128 up resnets (256 -> 128 channels) fold; the 64 x 64 / 32 x 32 ones are below the persistent kernel's grid size
    assert outs[True][2].count("siss_conv3x3_sc") == 2 and outs[False][2].count("siss_conv3x3_sc") == 0
    # ... and so do their backward tails: the shortcut's dgrad rides in conv2's 3x3 dgrad over the same cotangent
    assert outs[True][2].count("siss_conv3x3_dgrad_sc") == 2 and outs[False][2].count("siss_conv3x3_dgrad_sc") == 0
    pa, pb = outs[True][0], outs[False][0]
    assert float((pa - pb).abs().max()) <= 1e-2 * float(pb.abs().max())
    for s in range(2):
        ga, gb = outs[True][1][s].double(), outs[False][1][s].double()
        assert float((ga * gb).sum() / (ga.norm() * gb.norm())) >= 0.999, s
        assert abs(float(ga.norm() / gb.norm()) - 1) < 1e-2, s


def test_dgrad_weight_copies_from_the_shadow_equal_those_from_the_master(setup):
    """refresh_weights() builds the transposed / tap-reversed dgrad operands from the bf16 shadow (siss_conv_weight_dgrad_multi_bf16,
    16-B accesses; conv_in's 3 input channels take its scalar walk): bitwise what the f32-master form writes."""
    from siss_amd import lib
    eng, _, _ = setup
    eng.refresh_weights(cast_shadow=True)
    got = eng._wt_all.clone()
    assert float(got.float().abs().max()) > 0
    ref = torch.zeros_like(got)
    lib.call("siss_conv_weight_dgrad_multi", eng.ps.flat, ref, eng._wt_jobs, eng._wt_njobs, eng._wt_tiles)
    torch.cuda.synchronize()
    # (the buffer's alignment gaps between weights are never written by either form)
    for n, w in eng.wT.items():
        off = w.data_ptr() - eng._wt_all.data_ptr()
        assert off % 2 == 0
        r = ref[off // 2: off // 2 + w.numel()].view_as(w)
        assert torch.equal(w, r), n


def test_dgrad_weight_copy_kernel_on_odd_shapes():
    """The same kernel driven directly: a job table mixing vector-path weights (co, ci multiples of 8, partial 64 x 64 tiles) with
    ones that take the scalar walk (co or ci not a multiple of 8), against torch's permute."""
    import numpy as np
    from siss_amd import lib
    dev = torch.device("cuda:0")
    shapes = [(9, 72, 136), (1, 320, 768), (9, 12, 20), (1, 5, 64), (9, 64, 3), (1, 128, 128)]
    g = torch.Generator().manual_seed(5)
    rec = np.zeros(len(shapes), dtype=np.dtype([("src", "<i8"), ("dst", "<i8"), ("taps", "<i4"), ("co", "<i4"), ("ci", "<i4"), ("tile0", "<i4")]))
    off = tiles = 0
    for i, (t, co, ci) in enumerate(shapes):
        rec[i] = (off, off, t, co, ci, tiles)
        off += -(-t * co * ci // 64) * 64
        tiles += t * (-(-co // 64)) * (-(-ci // 64))
    master = torch.randn(off, generator=g).to(dev)
    shadow = master.to(torch.bfloat16)
    out = torch.zeros(off, dtype=torch.bfloat16, device=dev)
    jobs = torch.from_numpy(rec.view(np.uint8)).to(dev)
    lib.call("siss_conv_weight_dgrad_multi_bf16", shadow, out, jobs, len(shapes), tiles)
    torch.cuda.synchronize()
    for (t, co, ci), r in zip(shapes, rec):
        o = int(r["src"])
        w = shadow[o:o + t * co * ci].view(t, co, ci)
        want = w.flip(0).permute(0, 2, 1).contiguous()
        assert torch.equal(out[o:o + t * co * ci].view(t, ci, co), want), (t, co, ci)


def test_subpixel_upsample_equals_the_upsample_copy_form():
    """Upsample2D as four 2x2-tap phase convolutions on the low-resolution input (UNetEngine.subpixel_up, the default) against the
    literal form (nearest-2x copy, then the 3x3 convolution): same network, same weights and inputs -- prediction within bf16
    rounding of each other, and the gradients of the upsampler's weight / bias (the fold of the 16 phase-tap gradients onto the nine
    taps, the bias sums over the four planes) and of everything upstream of it (the phase dgrad) agree."""
    from siss_amd.unet import UNetEngine
    hc, _ = _cfgs()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 3, 16, 16, generator=g).cuda()
    t = torch.tensor([999, 500, 3, 40]).cuda()
    cot = torch.randn(8, 3, 16, 16, generator=g).cuda()
    res = []
    sd = None
    for sub in (1, 0):
        eng = UNetEngine(hc, "cuda:0")
        eng.subpixel_up, eng.subpixel_min_px = sub, 0
        if sd is None:
            sd = eng.init_random(seed=9)
        else:
            eng.load_state_dict(sd)
        pred = eng.forward(x, t).clone()
        eng.zero_grad()
        eng.backward(cot.contiguous(), nsets=2)
        torch.cuda.synchronize()
        res.append((pred, [eng.ps.grads_ref(s) for s in range(2)]))
    (p1, g1), (p0, g0) = res
    assert (p1 - p0).abs().max().item() <= 1e-2 * p0.abs().max().item()
    for s in range(2):
        tot = float(torch.sqrt(sum(v.square().sum() for v in g0[s].values())))
        for n in g0[s]:
            if n.endswith("to_k.bias") or float(g0[s][n].norm()) < 1e-6 * tot:   # (key biases: an identically zero gradient, bf16 noise)
                continue
            c = _cos(g1[s][n].float(), g0[s][n].float())
            assert c >= (0.9995 if "upsamplers" in n else 0.995), (s, n, c)
        for n in ("up_blocks.0.upsamplers.0.conv.weight", "up_blocks.0.upsamplers.0.conv.bias"):
            rel = float((g1[s][n] - g0[s][n]).norm() / g0[s][n].norm())
            assert rel <= 2e-2, (s, n, rel)


@pytest.mark.parametrize("ga", [1, 2])
def test_sparse_gradient_fill_is_bitwise_the_full_fill(setup, ga):
    """zero_grad(sparse_key=...): from the second step under a key the fill skips what the backward pass overwrites
    (siss_gemm_tn nsplits = -2 + siss_gemm_tn_overwrite_log + siss_zero_ranges).  Same inputs, gradient buffer poisoned before
    every step: the gradients of each of three steps are those of the full fill -- bit for bit where a product overwrote them,
    to the float atomics' ordering elsewhere -- and so are the parameters; a pass that does not overwrite what the fill skipped raises.
    ga = 2: the second micro-batch of a step ADDS to the tiles the first one overwrote (delete_celeb.py:705-711)."""
    from siss_amd.step import SISSStepper
    from oracle import schedule as S
    eng, _, sd = setup
    ac = S.alphas_cumprod()
    batches = [[_batch(torch.Generator().manual_seed(11 + 7 * i + j)) for j in range(ga)] for i in range(3)]

    def run(sparse):
        eng._fill_plans.clear()
        eng.sparse_fill, eng.sparse_min_floats = sparse, 256
        out = []
        for mbs in batches:                      # every step from the same state (the atomics' noise would drift apart otherwise)
            eng.load_state_dict(sd)
            st = SISSStepper(eng, ac, lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6, scaling_norm=5.0, lambd=0.5,
                             train_batch_size=4, grad_accum=ga, mixed_precision=None)
            eng.ps.grads.fill_(float("nan"))
            for mb in mbs:
                st.micro_step(mb["x0"], mb["a0"], mb["noise"], mb["t"].cuda(), mb["u"])
            out.append((eng.ps.grads.clone(), {k: v.clone() for k, v in eng.state_dict().items()}))
        return out, st

    try:
        full, _ = run(False)
        assert not eng._fill_plans
        got, st = run(True)
        (key, plan), = eng._fill_plans.items()
        assert plan["skipped_bytes"] > 0 and plan["n"] > 0, plan
        skipped = torch.zeros(eng.ps.grads.numel(), dtype=torch.bool, device="cuda")
        for a, n in plan["stretches"]:
            if n >= eng.sparse_min_floats:
                skipped[-(-a // 4) * 4:(a + n) // 4 * 4] = True
        assert int(skipped.sum()) * 4 == plan["skipped_bytes"]
        for i, ((ga, pa), (gb, pb)) in enumerate(zip(full, got)):
            assert torch.isfinite(gb).all(), i
            ga, gb = ga.view(-1), gb.view(-1)
            assert torch.equal(ga[skipped], gb[skipped]), i                     # one-split products: deterministic
            torch.testing.assert_close(gb, ga, rtol=1e-3, atol=1e-6 * float(ga.abs().max()))
            for k in pa:
                assert torch.isfinite(pb[k]).all(), (i, k)
                # (an Adam step is lr * g / (|g| + eps): the atomics' last bit can turn a near-zero gradient's update around)
                torch.testing.assert_close(pb[k], pa[k], rtol=0, atol=2.5e-4)
                assert float((pb[k] - pa[k]).abs().mean()) < 2e-6, (i, k)
        # a pass that overwrites something else than the fill assumed is an error, not a wrong gradient
        plan["stretches"] = plan["stretches"][:-1]
        mb = batches[0][0]
        with pytest.raises(RuntimeError, match="sparse gradient fill"):
            st.micro_step(mb["x0"], mb["a0"], mb["noise"], mb["t"].cuda(), mb["u"])
        assert not eng._fill_plans
    finally:
        eng.sparse_fill, eng.sparse_min_floats = True, 16384
        eng._fill_plans.clear()


@pytest.mark.parametrize("loss_fn", ["naive_del", "double_forward_with_neg_del"])
def test_sparse_gradient_fill_with_other_objectives(setup, loss_fn):
    """The sparse fill under a ONE-set objective (naive_del: nsets = 1 -- the second gradient set is never written and must come
    out of every fill as zeros) and under the two-forward objective (a batch-2B pass: another key, another record)."""
    from siss_amd.step import SISSStepper
    from oracle import schedule as S
    eng, _, sd = setup
    ac = S.alphas_cumprod()
    batches = [_batch(torch.Generator().manual_seed(41 + i)) for i in range(3)]

    def run(sparse):
        eng._fill_plans.clear()
        eng.sparse_fill, eng.sparse_min_floats = sparse, 256
        out = []
        for mb in batches:
            eng.load_state_dict(sd)
            st = SISSStepper(eng, ac, lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6, scaling_norm=5.0, lambd=0.5,
                             train_batch_size=4, loss_fn=loss_fn, mixed_precision=None, superfactor=3.0)
            eng.ps.grads.fill_(float("nan"))
            st.step(mb["x0"], mb["a0"], mb["noise"], mb["t"].cuda(), mb["u"])
            out.append(eng.ps.grads.clone())
        return out

    try:
        full = run(False)
        got = run(True)
        (plan,) = eng._fill_plans.values()
        assert plan["skipped_bytes"] > 0
        for i, (ga, gb) in enumerate(zip(full, got)):
            assert torch.isfinite(gb).all(), i
            torch.testing.assert_close(gb, ga, rtol=1e-3, atol=1e-6 * float(ga.abs().max()))
            if loss_fn == "naive_del":
                assert float(gb[1].abs().max()) == 0.0, i        # the set nobody writes
    finally:
        eng.sparse_fill, eng.sparse_min_floats = True, 16384
        eng._fill_plans.clear()
