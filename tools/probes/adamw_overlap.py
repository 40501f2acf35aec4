#!/usr/bin/env python
"""Can the AdamW pass (HBM-bound, 56 VGPRs, no LDS: its waves fit beside the persistent conv kernel's blocks) hide beside the NEXT
step's forward pass?  Times forward + dual backward alone, with the optimizer kernel behind it on the same stream, and with the
optimizer kernel (on copies of its buffers) on a side stream beside the forward pass.

    python tools/probes/adamw_overlap.py [celebahq256|sd15] [batch]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from siss_amd import lib                                   # noqa: E402
from siss_amd.step import SISSStepper                      # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "celebahq256"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
cond = {}
if which == "sd15":
    from siss_amd.config import UNet2DConditionConfig
    from siss_amd.unet_cond import UNetCondEngine
    cfg = UNet2DConditionConfig.sd15()
    eng = UNetCondEngine(cfg, dev)
    cond = {"encoder_hidden_states": torch.randn(B, 77, 768, generator=g, device=dev).to(torch.bfloat16)}
    okw = dict(lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-2, scaling_norm=750.0)
else:
    from siss_amd.config import UNet2DConfig
    from siss_amd.unet import UNetEngine
    cfg = UNet2DConfig.celebahq256()
    eng = UNetEngine(cfg, dev)
    okw = dict(lr=5e-6, betas=(0.95, 0.999), weight_decay=1e-6, scaling_norm=500.0)
eng.init_random(seed=1)
ac = torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, 1000), 0)
st = SISSStepper(eng, ac, lambd=0.5, train_batch_size=B, mixed_precision="bf16", **okw)
hw, c = cfg.sample_size, cfg.in_channels
x = torch.randn(B, c, hw, hw, generator=g, device=dev).to(torch.bfloat16)
t = torch.full((B,), 999, dtype=torch.long, device=dev)
cot = (torch.randn(2 * B, c, hw, hw, generator=g, device=dev) * 1e-2).contiguous()
o = st.opt
n = o.p.numel()
p2, m2, v2, sh2 = o.p.clone(), o.m.clone(), o.v.clone(), o.shadow.clone()
g2 = torch.zeros(2, n, device=dev)
lib.call("siss_grad_norms_scale", g2[0], g2[1], n, 0, 500.0, o.max_grad_norm, o.betas[0], o.betas[1], o.partials, o.scalars)


def adam():
    lib.call("siss_recombine_clip_adamw", g2[0], g2[1], p2, m2, v2, sh2, None, n, o.lr, o.betas[0], o.betas[1], o.eps, o.wd, o.scalars)


def fwd():
    if cond:
        eng.forward(x, t, **cond)
    else:
        eng.forward(x, t)


def bwd():
    eng.zero_grad()
    eng.backward(cot, nsets=2)


side = torch.cuda.Stream()


def timed(fn, k=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k


def serial():
    fwd(); bwd(); adam()


def beside_forward():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        adam()
    fwd()
    torch.cuda.current_stream().wait_stream(side)
    bwd()


def beside_backward():
    fwd()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        adam()
    bwd()
    torch.cuda.current_stream().wait_stream(side)


rows = [("forward + backward", lambda: (fwd(), bwd())), ("AdamW alone", adam), ("... + AdamW behind them", serial),
        ("AdamW beside the forward pass", beside_forward), ("AdamW beside the backward pass", beside_backward),
        ("forward + backward", lambda: (fwd(), bwd())), ("... + AdamW behind them", serial), ("AdamW beside the forward pass", beside_forward)]
print(f"{which} B = {B}, {n / 1e6:.1f} M parameters (eager launches, ms per iteration)")
for label, fn in rows:
    print(f"  {label:36s} {timed(fn):8.3f}", flush=True)
