#!/usr/bin/env python
"""Where in the backward pass does the main stream wait for the side stream?  Logs every `_join_side` that actually waits and every
side flush of ONE eager CelebA-HQ step (call stack + the sequence number of the launch it precedes)."""
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from siss_amd import lib                                   # noqa: E402
from siss_amd.config import UNet2DConfig                   # noqa: E402
from siss_amd.step import SISSStepper                      # noqa: E402
from siss_amd.unet import UNetEngine                       # noqa: E402

dev = torch.device("cuda:0")
B = 16
eng = UNetEngine(UNet2DConfig.celebahq256(), dev)
eng.init_random(seed=42)
ac = torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, 1000), 0)
st = SISSStepper(eng, ac, lambd=0.5, train_batch_size=B, mixed_precision="bf16", lr=5e-6, betas=(0.95, 0.999), weight_decay=1e-6,
                 scaling_norm=500.0)
g = torch.Generator(device=dev).manual_seed(42)
x0 = torch.randn(B, 3, 256, 256, generator=g, device=dev).to(torch.bfloat16)
a0 = torch.randn(1, 3, 256, 256, generator=g, device=dev).repeat(B, 1, 1, 1).to(torch.bfloat16)
noise = torch.randn(B, 3, 256, 256, generator=g, device=dev).to(torch.bfloat16)
t = torch.full((B,), 999, dtype=torch.long, device=dev)
u = torch.rand(B, generator=g, device=dev)
for _ in range(2):
    st.step(x0, a0, noise, t, u)
torch.cuda.synchronize()

log = []
join0, flush0 = eng._join_side, eng._flush_wgrads_side


def where():
    fr = traceback.extract_stack()[:-2]
    return " <- ".join(f"{f.name}:{f.lineno}" for f in reversed(fr) if f.filename.endswith("unet.py"))[:200]


def join():
    if eng._side_busy:
        log.append((len(lib.PROF), "JOIN ", where()))
    join0()


def flush():
    log.append((len(lib.PROF), f"FLUSH {len(eng._wq)} jobs", where()))
    flush0()


eng._join_side, eng._flush_wgrads_side = join, flush
lib.PROF = []
st.step(x0, a0, noise, t, u)
torch.cuda.synchronize()
prof, lib.PROF = lib.PROF, None
t0 = prof[0][1]
for n, what, w in log:
    name, s, e, _work, key = prof[min(n, len(prof) - 1)][:5]
    print(f"launch {n:4d} at {t0.elapsed_time(s):7.2f} ms  {what:16s} next: {name} {' '.join(map(str, key))[:60]} ({s.elapsed_time(e) * 1e3:.0f} us)\n      {w}")
print(f"step: {t0.elapsed_time(prof[-1][2]):.2f} ms, {len(prof)} launches")
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        for i, (name, s, e, _work, key, *_r) in enumerate(prof):
            f.write(f"{i:4d} {t0.elapsed_time(s):8.3f} {s.elapsed_time(e) * 1e3:8.1f} us  {name} {' '.join(map(str, key))}\n")
