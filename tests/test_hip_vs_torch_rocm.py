"""The reference's own stack on the SAME GPU as a baseline: the torch-only restatement of the step (oracle/step.py =
delete_celeb.py:557-773: two backward calls with retain_graph, per-parameter clone / subtract / norm loops,
clip_grad_norm_, torch.optim.AdamW) driving the torch UNet (oracle/unet.py) through PyTorch-ROCm's own kernels
(MIOpen / hipBLASLt) under bf16 autocast, at BASELINE configs[1] (CelebA-HQ 256x256, B = 16) -- against this
repository's fused step.  The reference publishes no throughput number (BASELINE.json `published: {}`), so this is
the closest thing to "the reference on MI355X".  Asserts that the HIP path is faster.

Parity at FULL SIZE is asserted against the fp32 oracle, not against the autocast run: the same reference loop over the
fp32 torch UNet (no autocast) on the same GPU gives the step's scalars (||g_x||, ||g_a||, s, pre-clip ||g||) and the
parameter update; the fused bf16 step must match them within SURVEY.md section 8c's tolerances -- scalars rel 5e-2, masked
update-direction cosine >= 0.99 (tests/parity_util.py) -- at BASELINE configs[1]'s real size (B = 16, 113.7 M parameters).
The UNet oracle itself is unpinned by the reference (diffusers absent: DESIGN.md section 6)."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


class _Autocast(torch.nn.Module):
    """accelerate.prepare(mixed_precision='bf16'): forward under autocast, outputs cast back to fp32."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, *a, **k):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = self.net(*a, **k)
        return tuple(o.float() for o in out)

    def named_parameters(self, *a, **k):
        return self.net.named_parameters(*a, **k)

    def parameters(self, *a, **k):
        return self.net.parameters(*a, **k)


def test_fused_step_beats_pytorch_rocm_eager_on_the_same_gpu():
    from siss_amd.config import UNet2DConfig
    from siss_amd.step import SISSStepper
    from siss_amd.unet import UNetEngine
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    from oracle.unet import OracleUNet2D, UNetConfig
    import gc
    from parity_util import assert_update_direction, check_scalars
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    dev = torch.device("cuda:0")
    B, HW = 16, 256
    eng = UNetEngine(UNet2DConfig.celebahq256(), dev)
    sd = eng.init_random(seed=42)
    ac = S.alphas_cumprod().to(dev)
    okw = dict(lr=5e-6, betas=(0.95, 0.999), weight_decay=1e-6)
    L = OracleDeletionLoss(*S.gamma_sigma(ac))
    g = torch.Generator(device=dev).manual_seed(42)
    x0 = (torch.rand(B, 3, HW, HW, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    a0 = (torch.rand(1, 3, HW, HW, generator=g, device=dev) * 2 - 1).repeat(B, 1, 1, 1).to(torch.bfloat16)
    noise = torch.randn(B, 3, HW, HW, generator=g, device=dev).to(torch.bfloat16)
    t = torch.full((B,), 999, dtype=torch.long, device=dev)
    u = torch.rand(B, generator=g, device=dev)
    mb = dict(x0=x0, a0=a0, noise=noise, t=t, u=u)

    # ---- the fp32 oracle step (no autocast) at full size: the parity reference ----
    net32 = OracleUNet2D(UNetConfig.celebahq256())
    net32.load_state_dict(sd)
    net32 = net32.to(dev).float()
    opt32 = torch.optim.AdamW(net32.parameters(), **okw)
    mb32 = dict(x0=x0.float(), a0=a0.float(), noise=noise.float(), t=t, u=u)     # bf16-rounded inputs, fp32 arithmetic
    ref32, _, _, gfin = unlearning_step(net32, opt32, L, "importance_sampling_with_mixture", ac, [mb32], train_batch_size=B,
                                        scaling_norm=500.0, loss_params={"lambd": 0.5})
    after32 = {n: p.detach().cpu() for n, p in net32.named_parameters()}
    gfin = {n: v.detach().cpu() for n, v in gfin.items()}
    del net32, opt32
    gc.collect(); torch.cuda.empty_cache()

    net = OracleUNet2D(UNetConfig.celebahq256())
    net.load_state_dict(sd)
    net = _Autocast(net.to(dev))
    opt = torch.optim.AdamW(net.parameters(), **okw)

    def ref_step():
        return unlearning_step(net, opt, L, "importance_sampling_with_mixture", ac, [mb], train_batch_size=B,
                               scaling_norm=500.0, loss_params={"lambd": 0.5})[0]
    ref_step()                                        # warm-up (MIOpen kernel selection)
    torch.cuda.synchronize()
    n_ref = 3
    t0 = time.perf_counter()
    for _ in range(n_ref):
        ref_step()
    torch.cuda.synchronize()
    ms_ref = (time.perf_counter() - t0) / n_ref * 1e3

    del net, opt
    gc.collect(); torch.cuda.empty_cache()

    st = SISSStepper(eng, ac, scaling_norm=500.0, lambd=0.5, train_batch_size=B, mixed_precision="bf16", **okw)
    st.step(x0, a0, noise, t, u)
    got = st.stats()
    check_scalars(ref32, got)                          # rel 5e-2 vs the fp32 oracle, full size
    cos = assert_update_direction(sd, after32, eng.state_dict(), gfin, "full-size SISS step")
    print(f"\nfull-size step vs fp32 oracle: " + ", ".join(f"{k} {got[k]:.5g} / {getattr(ref32, k):.5g}" for k in
          ("norm_loss_x", "norm_loss_a", "scaling_factor", "pre_clip_norm")) + f"; masked update cosine {cos:.5f}")
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        st.step(x0, a0, noise, t, u)
    torch.cuda.synchronize()
    ms_hip = (time.perf_counter() - t0) / n * 1e3
    import os
    print(f"(MIOPEN_FIND_MODE={os.environ.get('MIOPEN_FIND_MODE', 'default')}: with MIOpen's exhaustive default the reference stack measured "
          f"269.8 ms/step in round 1, docs/history.md)")
    print(f"PyTorch-ROCm eager (reference loop, bf16 autocast): {ms_ref:.1f} ms/step = {B / ms_ref * 1e3:.1f} samples/s; "
          f"fused HIP step (eager launches): {ms_hip:.1f} ms/step = {B / ms_hip * 1e3:.1f} samples/s; "
          f"speed-up {ms_ref / ms_hip:.2f}x")
    assert ms_hip < ms_ref
    # ... and against the reference stack's RECORDED time with MIOpen's exhaustively tuned kernels (269.8 ms, round 1), not only
    # against today's run, whose oracle side uses MIOpen's heuristic find mode (tests/conftest.py) and may be slower than that
    assert ms_hip < 269.8 / 3.0, ms_hip
