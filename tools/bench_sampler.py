#!/usr/bin/env python
"""Sampler / evaluation path throughput.  The reference's `log_metrics` runs, at `sampling_steps: 1`
(config/delete_celeb.yaml:97), ~301 UNet forwards per optimizer step: a 50-step DDPM sampling of `eval_batch_size` images
(evaluate.py:37-50, pipeline.num_inference_steps: 50) and the 251-step inject-then-denoise of the forget image
(evaluate.py:64-79, metrics.denoising_injections.timestep: 250; delete_celeb.py:376-436, :486-503).  Forward only, CelebA-HQ
256 x 256 architecture, random-init weights, synthetic inputs.

    python tools/bench_sampler.py [--batch 1 16] > profiles/rNN_sampler_celeb.json

One JSON line: per batch size, the two schedules timed end to end (captured forward replayed per denoising step vs eager launches),
images / s, UNet forwards / s, and the forward's fraction of the bf16 MFMA roof (1 x 498.35 GFLOP per sample and forward, SURVEY.md
§8d, against 2.5 PFLOP/s)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siss_amd.config import UNet2DConfig           # noqa: E402
from siss_amd.model import UNet2DModel             # noqa: E402
from siss_amd.sampler import Evaluator             # noqa: E402
from siss_amd.scheduler import DDPMScheduler       # noqa: E402

FWD_GFLOP_PER_SAMPLE = 498.35
PEAK_BF16_TFLOPS = 2500.0


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 16], help="eval_batch_size (config/delete_celeb.yaml:101 ships 1)")
    ap.add_argument("--sample-steps", type=int, default=50)
    ap.add_argument("--inject-t", type=int, default=250)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    unet = UNet2DModel(UNet2DConfig.celebahq256(), device=dev)
    unet.engine.init_random(seed=0)
    sch = DDPMScheduler()
    rows = []
    for B in a.batch:
        x = torch.randn(B, 3, 256, 256, device=dev)
        row = {"eval_batch_size": B}
        for graph in (True, False):
            ev = Evaluator(use_graph=graph)
            ev.load_model(unet, sch)
            ev.denoise_images(x, 2)                       # warm-up (and the capture)
            dt_s = timed(lambda: ev.sample_images(B, num_inference_steps=a.sample_steps))
            dt_d = timed(lambda: ev.denoise_images(x, a.inject_t))
            nd = a.inject_t + 1
            fps = (a.sample_steps + nd) * B / (dt_s + dt_d)                      # sample-forwards per second over both schedules
            row["hipgraph" if graph else "eager"] = {
                "sample_%d_steps_s" % a.sample_steps: round(dt_s, 4), "sample_images_per_s": round(B / dt_s, 3),
                "inject_denoise_%d_steps_s" % nd: round(dt_d, 4), "denoised_images_per_s": round(B / dt_d, 4),
                "ms_per_denoising_step": round(dt_d / nd * 1e3, 3), "unet_forwards_per_s": round(fps, 1),
                "forward_mfma_frac": round(fps * FWD_GFLOP_PER_SAMPLE / 1e3 / PEAK_BF16_TFLOPS, 4)}
        row["graph_vs_eager"] = round(row["eager"]["ms_per_denoising_step"] / row["hipgraph"]["ms_per_denoising_step"], 3)
        rows.append(row)
    print(json.dumps({"metric": "sampler / eval path, CelebA-HQ-256 DDPM UNet forward-only (evaluate.py:37-79)", "dtype": "bf16",
                      "data": "synthetic", "device": torch.cuda.get_device_name(0),
                      "forward_gflop_per_sample": FWD_GFLOP_PER_SAMPLE, "peak_tflops": PEAK_BF16_TFLOPS, "results": rows}))


if __name__ == "__main__":
    main()
