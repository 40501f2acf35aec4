"""siss_mixture_fwd against oracle/loss.py::siss_terms on SD-shaped inputs (4 x 64 x 64 latents, scaled-linear schedule)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from siss_amd import lib
from siss_amd.loss import mixture_fwd
from oracle.loss import siss_terms, mix
dev = torch.device("cuda:0"); lib.load()
torch.manual_seed(0)
B = 8
for name, shape, ac in (("sd", (B, 4, 64, 64), torch.cumprod(1.0 - torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000) ** 2, 0)),
                        ("celeb", (B, 3, 256, 256), torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, 1000), 0))):
    gam, sig = ac.sqrt(), (1 - ac).sqrt()
    sc = 0.18215 if name == "sd" else 1.0
    for T in (999, 700, 300):
        x0 = (sc * torch.randn(shape)).to(torch.bfloat16); a0 = (sc * torch.randn(1, *shape[1:])).repeat(B, 1, 1, 1).to(torch.bfloat16)
        noise = torch.randn(shape).to(torch.bfloat16); u = torch.rand(B); t = torch.full((B,), T, dtype=torch.long)
        m = mixture_fwd(x0.to(dev), a0.to(dev), noise.to(dev), t.to(dev), u.to(dev), ac.to(dev), gam.to(dev), sig.to(dev), 0.5)     # (the tables must be DEVICE tensors)
        g, s = gam[T].to(torch.bfloat16), sig[T].to(torch.bfloat16)
        nk, nf = g * x0 + s * noise, g * a0 + s * noise
        xm = mix(nk, nf, u > 0.5)
        _, _, dx, da, iwx, iwa = siss_terms(xm, x0, a0, gam[t], sig[t], 0.5)
        print(name, T, "x_mix equal", bool(torch.equal(m.x_mix.cpu(), xm)), "\n   hip iw_x", [round(float(v), 3) for v in m.iw_x.cpu()], "\n   ora iw_x", [round(float(v), 3) for v in iwx],
              "\n   dx-da", [round(float(v), 1) for v in (dx - da)])
