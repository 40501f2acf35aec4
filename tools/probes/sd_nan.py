"""Where do the NaN step scalars of `bench.py --config sd15` come from?  One step at B = 2 with the bench's inputs; finiteness of the
prediction, the cotangent, the gradient sets and the stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from siss_amd import lib
from siss_amd.config import UNet2DConditionConfig
from siss_amd.unet_cond import UNetCondEngine
from siss_amd.step import SISSStepper
dev = torch.device("cuda:0")
cfg = UNet2DConditionConfig.sd15()
eng = UNetCondEngine(cfg, dev)
eng.init_random(seed=42)
B = int(os.environ.get("B", "2")); hw, cin = cfg.sample_size, cfg.in_channels
g = torch.Generator(device=dev).manual_seed(42)
ac = torch.cumprod(1.0 - torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2, 0)
st = SISSStepper(eng, ac, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, scaling_norm=750.0, lambd=0.5, train_batch_size=B, mixed_precision="bf16")
x0 = (0.18215 * torch.randn(B, cin, hw, hw, generator=g, device=dev)).to(torch.bfloat16)
a0 = (0.18215 * torch.randn(1, cin, hw, hw, generator=g, device=dev)).repeat(B, 1, 1, 1).to(torch.bfloat16)
cond = {"encoder_hidden_states": torch.randn(1, 77, cfg.cross_attention_dim, generator=g, device=dev).repeat(B, 1, 1).to(torch.bfloat16)}
torch.cuda.manual_seed(42)
rms = lambda v: float(v.float().pow(2).mean().sqrt())
print("rms x0", rms(x0), "a0", rms(a0), "x0 - a0", rms(x0.float() - a0.float()))
for T in (999, 500):
    noise = torch.randn(B, cin, hw, hw, device=dev, dtype=torch.bfloat16)
    t = torch.full((B,), T, device=dev, dtype=torch.long)
    u = torch.rand(B, device=dev)
    from siss_amd.loss import mixture_fwd
    m = mixture_fwd(x0, a0, noise, t, u, st.ac, st.gamma_tab, st.sigma_tab, st.lambd)
    print("t", T, "x_mix finite", bool(torch.isfinite(m.x_mix.float()).all()), "iw_x", m.iw_x.tolist(), "iw_a", m.iw_a.tolist())
    from oracle.loss import siss_terms
    _, _, dx, da, iwx, iwa = siss_terms(m.x_mix.cpu(), x0.cpu(), a0.cpu(), st.gamma_tab.cpu()[t.cpu()], st.sigma_tab.cpu()[t.cpu()], 0.5)
    exp_keep = (st.gamma_tab[T].to(torch.bfloat16) * x0 + st.sigma_tab[T].to(torch.bfloat16) * noise)
    exp_forg = (st.gamma_tab[T].to(torch.bfloat16) * a0 + st.sigma_tab[T].to(torch.bfloat16) * noise)
    print("   rms noise", rms(noise), "x_mix", rms(m.x_mix), "| rows equal to the keep form", [bool(torch.equal(m.x_mix[i], exp_keep[i])) for i in range(B)],
          "to the forget form", [bool(torch.equal(m.x_mix[i], exp_forg[i])) for i in range(B)])
    print("   u", [round(float(v), 2) for v in u.cpu()], "gamma_t", float(m.gamma_t[0]), "sigma_t", float(m.sigma_t[0]))
    print("   hip dist_x - dist_a", [round(float(v), 1) for v in (m.dist_x - m.dist_a).cpu()])
    print("   ora dist_x - dist_a", [round(float(v), 1) for v in (dx - da)], "ora iw_x", [round(float(v), 2) for v in iwx])
    pred = eng.forward(m.x_mix, t, **cond)
    print("   pred finite", bool(torch.isfinite(pred.float()).all()), "abs max", float(pred.float().abs().max()))
    st.micro_step(x0, a0, noise, t, u, cond)
    gr = eng.ps.grads
    print("   grads finite", bool(torch.isfinite(gr).all()), "max", float(gr.abs().max()))
    s = st.stats()
    print("   stats", {k: s[k] for k in ("norm_loss_x", "norm_loss_a", "scaling_factor", "pre_clip_norm")})
