"""Full-size OPTIMIZER STEPS of BASELINE configs[3] and configs[4] against the fp32 oracle step on the same GPU (round 4; configs[1]
has had this since round 3: tests/test_hip_vs_torch_rocm.py).  The forward / dual-backward parity of the two configurations is in
test_hip_large_kernels.py / test_hip_unet_cond.py; here the WHOLE step is compared: the reference loop (oracle/step.py =
delete_celeb.py:682-773, delete_sd.py:977-1127: backward passes, clone / subtract split, norm fix, clip_grad_norm_, AdamW) over the
fp32 torch network gives ||g_x||, ||g_a||, s, the pre-clip norm and the parameter update; the fused bf16 step must match the
scalars within 5e-2 and the masked update direction with cosine >= 0.99 (tests/parity_util.py)."""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu


def _report(what, ref, got, cos):
    print(f"\n{what} vs fp32 oracle: " + ", ".join(f"{k} {got[k]:.5g} / {getattr(ref, k):.5g}" for k in
          ("norm_loss_x", "norm_loss_a", "scaling_factor", "pre_clip_norm")) + f"; masked update cosine {cos:.5f}")


def test_full_size_no_is_step_matches_the_fp32_oracle_step():
    """BASELINE configs[3]: SISS-No-IS (`double_forward_with_neg_del`, losses/ddpm_deletion_loss.py:60-67), CelebA-HQ 256 x 256,
    B = 16.  The oracle walks the 16 samples as four micro-batches of 4 with gradient accumulation (the arithmetic of the step:
    the loss is normalised by train_batch_size x GA = 16, samples do not interact) to bound the fp32 autograd memory; the HIP step
    is ONE 32-image forward + dual backward."""
    from siss_amd.config import UNet2DConfig
    from siss_amd.step import SISSStepper
    from siss_amd.unet import UNetEngine
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    from oracle.unet import OracleUNet2D, UNetConfig
    from parity_util import assert_update_direction, check_scalars
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    dev = torch.device("cuda:0")
    B, HW, MB = 16, 256, 4
    loss_fn = "double_forward_with_neg_del"
    eng = UNetEngine(UNet2DConfig.celebahq256(), dev)
    sd = eng.init_random(seed=44)
    ac = S.alphas_cumprod().to(dev)
    okw = dict(lr=5e-6, betas=(0.95, 0.999), weight_decay=1e-6)
    g = torch.Generator(device=dev).manual_seed(17)
    x0 = (torch.rand(B, 3, HW, HW, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    a0 = (torch.rand(1, 3, HW, HW, generator=g, device=dev) * 2 - 1).repeat(B, 1, 1, 1).to(torch.bfloat16)
    noise = torch.randn(B, 3, HW, HW, generator=g, device=dev).to(torch.bfloat16)
    t = torch.full((B,), 999, dtype=torch.long, device=dev)
    u = torch.rand(B, generator=g, device=dev)

    net = OracleUNet2D(UNetConfig.celebahq256())
    net.load_state_dict(sd)
    net = net.to(dev).float()
    opt = torch.optim.AdamW(net.parameters(), **okw)
    mbs = [dict(x0=x0[i:i + MB].float(), a0=a0[i:i + MB].float(), noise=noise[i:i + MB].float(), t=t[i:i + MB], u=u[i:i + MB])
           for i in range(0, B, MB)]
    ref, _, _, gfin = unlearning_step(net, opt, OracleDeletionLoss(*S.gamma_sigma(ac)), loss_fn, ac, mbs, train_batch_size=MB,
                                      scaling_norm=500.0)
    after = {n: p.detach().cpu() for n, p in net.named_parameters()}
    gfin = {n: v.detach().cpu() for n, v in gfin.items()}
    del net, opt
    gc.collect(); torch.cuda.empty_cache()

    st = SISSStepper(eng, ac, scaling_norm=500.0, train_batch_size=B, mixed_precision="bf16", loss_fn=loss_fn, **okw)
    st.step(x0, a0, noise, t, u)
    got = st.stats()
    check_scalars(ref, got)
    cos = assert_update_direction(sd, after, eng.state_dict(), gfin, "full-size No-IS step")
    _report("full-size No-IS step (B = 16)", ref, got, cos)


def test_sd15_full_size_step_matches_the_fp32_oracle_step():
    """BASELINE configs[4]: the SD v1.5 UNet (859,520,964 parameters), B = 4 (the batch bench.py --workload sd15 runs), 64 x 64 x 4 latents, 77 x 768 text conditioning, SISS
    lambd = 0.5 with config/delete_sd.yaml's optimizer and scaling_norm (delete_sd.py:977-1127)."""
    from siss_amd.config import UNet2DConditionConfig
    from siss_amd.step import SISSStepper
    from siss_amd.unet_cond import UNetCondEngine
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    from oracle.unet_cond import OracleUNet2DCondition, UNetCondConfig
    from parity_util import assert_update_direction, check_scalars
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    dev = torch.device("cuda:0")
    B = 4
    eng = UNetCondEngine(UNet2DConditionConfig.sd15(), dev)
    sd = eng.init_random(seed=3)
    ac = S.alphas_cumprod(beta_schedule="scaled_linear", beta_start=0.00085, beta_end=0.012).to(dev)
    okw = dict(lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)          # config/delete_sd.yaml:85,95-98
    g = torch.Generator(device=dev).manual_seed(23)
    x0 = (0.18215 * torch.randn(B, 4, 64, 64, generator=g, device=dev)).to(torch.bfloat16)     # delete_sd.py:883,888
    a0 = (0.18215 * torch.randn(1, 4, 64, 64, generator=g, device=dev)).repeat(B, 1, 1, 1).to(torch.bfloat16)
    noise = torch.randn(B, 4, 64, 64, generator=g, device=dev).to(torch.bfloat16)
    t = torch.full((B,), 999, dtype=torch.long, device=dev)
    u = torch.tensor([0.9, 0.2, 0.6, 0.35], device=dev)
    ctx = torch.randn(1, 77, 768, generator=g, device=dev).repeat(B, 1, 1).to(torch.bfloat16)  # one prompt repeated (:941-944)

    net = OracleUNet2DCondition(UNetCondConfig.sd15())
    net.load_state_dict(sd)
    net = net.to(dev).float()
    opt = torch.optim.AdamW(net.parameters(), **okw)
    mb = dict(x0=x0.float(), a0=a0.float(), noise=noise.float(), t=t, u=u)
    ref, _, _, gfin = unlearning_step(net, opt, OracleDeletionLoss(*S.gamma_sigma(ac)), "importance_sampling_with_mixture", ac,
                                      [mb], train_batch_size=B, scaling_norm=750.0, loss_params={"lambd": 0.5},
                                      conditioning={"encoder_hidden_states": ctx.float()})
    after = {n: p.detach().cpu() for n, p in net.named_parameters()}
    gfin = {n: v.detach().cpu() for n, v in gfin.items()}
    del net, opt
    gc.collect(); torch.cuda.empty_cache()

    st = SISSStepper(eng, ac, scaling_norm=750.0, lambd=0.5, train_batch_size=B, mixed_precision="bf16", **okw)
    st.step(x0, a0, noise, t, u, conditioning={"encoder_hidden_states": ctx})
    got = st.stats()
    check_scalars(ref, got)
    cos = assert_update_direction(sd, after, eng.state_dict(), gfin, "full-size SD v1.5 step")
    _report("full-size SD v1.5 step (B = 4)", ref, got, cos)
