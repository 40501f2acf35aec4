#!/usr/bin/env python
"""Sampler / evaluation path throughput (evaluate.py:37-79): UNet forwards per second of the
inject-then-denoise loop at the CelebA-HQ 256x256 architecture, eager launches vs the captured forward.

    python tools/bench_sampler.py [--batch 1 16] [--steps 20]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siss_amd.config import UNet2DConfig           # noqa: E402
from siss_amd.model import UNet2DModel             # noqa: E402
from siss_amd.sampler import Evaluator             # noqa: E402
from siss_amd.scheduler import DDPMScheduler       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 16])
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    unet = UNet2DModel(UNet2DConfig.celebahq256(), device="cuda:0")
    unet.engine.init_random(seed=0)
    sch = DDPMScheduler()
    for B in a.batch:
        x = torch.randn(B, 3, 256, 256, device="cuda:0")
        for graph in (False, True):
            ev = Evaluator(use_graph=graph)
            ev.load_model(unet, sch)
            ev.denoise_images(x, 2)                      # warm-up (and capture)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev.denoise_images(x, a.steps - 1)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"B={B:3d} graph={int(graph)}  {dt / a.steps * 1e3:8.2f} ms / denoising step   "
                  f"{B * a.steps / dt:8.1f} image-steps/s   (251-step denoise of the batch: {251 * dt / a.steps:6.2f} s)")


if __name__ == "__main__":
    main()
