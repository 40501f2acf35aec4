#!/usr/bin/env python
"""Upper bound of what folding GroupNorm + SiLU into the consuming convolution's prologue could save at the low-resolution
sites: the hipGraph-replayed CelebA-HQ step with those GroupNorm launches simply LEFT OUT (wrong numbers,
valid kernel times: no kernel here branches on data).  A fused prologue costs more than nothing, so the step cannot gain more
than `baseline - without`.

    python tools/probes/lowres_bound.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from siss_amd import lib                                   # noqa: E402
from siss_amd.config import UNet2DConfig                   # noqa: E402
from siss_amd.step import SISSStepper                      # noqa: E402
from siss_amd.unet import UNetEngine                       # noqa: E402

dev = torch.device("cuda:0")
B = 16
H_AT = {"siss_groupnorm_fwd": 8, "siss_groupnorm_fwd_ld": 8, "siss_groupnorm_fwd_qs": 11,
        "siss_groupnorm_bwd": 21, "siss_groupnorm_bwd_ld": 21, "siss_groupnorm_bwd_ld_s2d": 21}
skip = {"fwd": 0, "bwd": 0}
call0 = lib.call


def call(name, *args):
    at = H_AT.get(name)
    if at is not None and args[at] <= skip["bwd" if "bwd" in name else "fwd"]:
        skipped[0] += 1
        return 0
    return call0(name, *args)


skipped = [0]
lib.call = call
import siss_amd.unet as U                                  # noqa: E402
assert U.lib is lib


def measure(fwd, bwd, steps=20):
    skip["fwd"], skip["bwd"] = fwd, bwd
    eng = UNetEngine(UNet2DConfig.celebahq256(), dev)
    eng.init_random(seed=42)
    ac = torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, 1000), 0)
    st = SISSStepper(eng, ac, lambd=0.5, train_batch_size=B, mixed_precision="bf16", lr=5e-6, betas=(0.95, 0.999),
                     weight_decay=1e-6, scaling_norm=500.0)
    g = torch.Generator(device=dev).manual_seed(42)
    x0 = (torch.rand(B, 3, 256, 256, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    a0 = (torch.rand(1, 3, 256, 256, generator=g, device=dev) * 2 - 1).repeat(B, 1, 1, 1).to(torch.bfloat16)

    def one_step():
        noise = torch.randn(B, 3, 256, 256, device=dev, dtype=torch.bfloat16)
        t = torch.randint(999, 1000, (B,), device=dev)
        u = torch.rand(B, device=dev)
        st.micro_step(x0, a0, noise, t, u, None)

    for _ in range(3):
        one_step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        one_step()
        graph = torch.cuda.CUDAGraph()
        skipped[0] = 0
        with torch.cuda.graph(graph, stream=side):
            one_step()
    n_skipped = skipped[0]
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    for i in range(steps):
        marks[i].record()
        graph.replay()
    marks[steps].record()
    torch.cuda.synchronize()
    per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    del graph, st, eng
    torch.cuda.empty_cache()
    return per[len(per) // 2], n_skipped


rows = [("baseline", 0, 0), ("no GroupNorm forward at <= 16 x 16", 16, 0), ("baseline", 0, 0),
        ("no GroupNorm forward at <= 32 x 32", 32, 0), ("no GroupNorm forward or backward at <= 16 x 16", 16, 16), ("baseline", 0, 0)]
for label, f, b in rows:
    ms, n = measure(f, b)
    print(f"{label:50s} {ms:8.3f} ms / step (median of 20 replays), {n} launches left out", flush=True)
