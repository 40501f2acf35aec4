#!/usr/bin/env python
"""Per-(kernel, problem shape) time of ONE eager SISS step (HIP events around every launch, lib.PROF):
where the step's milliseconds go and what each GEMM shape achieves.

    python tools/step_breakdown.py [--config celebahq256|sd15] [--batch 16] [--top 60]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siss_amd import lib                                   # noqa: E402
from siss_amd.step import SISSStepper                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="celebahq256")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--engine-attr", action="append", default=[], help="name=int: set a schedule switch of the engine")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.batch
    g = torch.Generator(device=dev).manual_seed(42)
    cond = None
    if a.config == "sd15":
        from siss_amd.config import UNet2DConditionConfig
        from siss_amd.unet_cond import UNetCondEngine
        cfg = UNet2DConditionConfig.sd15()
        eng = UNetCondEngine(cfg, dev)
        ac = torch.cumprod(1.0 - torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000) ** 2, 0)
        kw = dict(lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-2, scaling_norm=750.0)
        sc = 0.18215
        cond = {"encoder_hidden_states": torch.randn(1, 77, 768, generator=g, device=dev).repeat(B, 1, 1)}
    else:
        from siss_amd.config import UNet2DConfig
        from siss_amd.unet import UNetEngine
        cfg = UNet2DConfig.celebahq256()
        eng = UNetEngine(cfg, dev)
        ac = torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, 1000), 0)
        kw = dict(lr=5e-6, betas=(0.95, 0.999), weight_decay=1e-6, scaling_norm=500.0)
        sc = 1.0
    for kv in a.engine_attr:
        k_, v_ = kv.split("=")
        setattr(eng, k_, int(v_))
    eng.init_random(seed=42)
    st = SISSStepper(eng, ac, lambd=0.5, train_batch_size=B, mixed_precision="bf16", **kw)
    hw, c = cfg.sample_size, cfg.in_channels
    x0 = (sc * torch.randn(B, c, hw, hw, generator=g, device=dev)).to(torch.bfloat16)
    a0 = (sc * torch.randn(1, c, hw, hw, generator=g, device=dev)).repeat(B, 1, 1, 1).to(torch.bfloat16)
    noise = torch.randn(B, c, hw, hw, generator=g, device=dev).to(torch.bfloat16)
    t = torch.full((B,), 999, dtype=torch.long, device=dev)
    u = torch.rand(B, generator=g, device=dev)
    for _ in range(2):
        st.step(x0, a0, noise, t, u, cond)
    torch.cuda.synchronize()
    lib.PROF = []
    st.step(x0, a0, noise, t, u, cond)
    torch.cuda.synchronize()
    prof, lib.PROF = lib.PROF, None
    rows = {}
    for name, s, e, work, key, _sym, _bytes in prof:
        d = rows.setdefault((name, key), [0, 0.0, 0.0])
        d[0] += 1; d[1] += s.elapsed_time(e); d[2] += work
    tot = sum(v[1] for v in rows.values())
    print(f"total event time {tot:.2f} ms over {len(prof)} launches")
    by_name = {}
    for (name, key), v in rows.items():
        d = by_name.setdefault(name, [0, 0.0, 0.0])
        d[0] += v[0]; d[1] += v[1]; d[2] += v[2]
    for name, v in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:25]:
        tf = f"{v[2] / (v[1] * 1e-3) / 1e12:7.0f} TF/s" if v[2] else ""
        print(f"  {name:34s} {v[0]:5d} launches {v[1]:8.3f} ms {100 * v[1] / tot:5.1f} %  {tf}")
    print("--- by shape ---")
    for (name, key), v in sorted(rows.items(), key=lambda kv: -kv[1][1])[:a.top]:
        tf = f"{v[2] / (v[1] * 1e-3) / 1e12:7.0f} TF/s" if v[2] else ""
        ks = " ".join(str(k) for k in key)
        print(f"  {name:24s} {ks:58s} x{v[0]:3d} {v[1]:8.3f} ms  {1e3 * v[1] / v[0]:8.1f} us/launch {tf}")


if __name__ == "__main__":
    main()
