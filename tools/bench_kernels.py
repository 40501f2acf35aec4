#!/usr/bin/env python
"""Isolated timing of the two MFMA GEMM kernels and the GroupNorm kernels at the CelebA-HQ layer shapes
(HIP events around back-to-back launches).  Usage: python tools/bench_kernels.py [--iters 20] [--only nt,tn,gn]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siss_amd import lib, ops          # noqa: E402
from siss_amd.layout import Act       # noqa: E402


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="nt,tn,gn")
    ap.add_argument("--zeros", action="store_true", help="zero operands (clock / power probe; never quote)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib.load()
    only = set(a.only.split(","))
    B = 16
    shapes = [(256, 128, 128), (256, 256, 128), (128, 128, 128), (64, 256, 256), (32, 256, 256), (16, 512, 512), (8, 512, 512)]
    for (hw, ci, co) in shapes:
        x = Act(B, hw, hw, ci, dev); dy = Act(2 * B, hw, hw, co, dev)
        if not a.zeros:
            x.interior().normal_(); dy.interior().normal_()
        w = (torch.randn(9, co, ci, device=dev) / (3 * ci ** 0.5)) * (0.0 if a.zeros else 1.0)
        wb = w.to(torch.bfloat16)
        wT = ops.dgrad_weight(w)
        y = Act(B, hw, hw, co, dev)
        dx = Act(2 * B, hw, hw, ci, dev)
        dW = torch.zeros(2, 9, co, ci, device=dev)
        bias = torch.zeros(co, device=dev)
        fl = 2.0 * B * (hw + 2) ** 2 * ci * co * 9
        if "nt" in only:
            t = timeit(lambda: ops.conv_fprop(x, wb, y, bias=bias), a.iters)
            print(f"fprop  {hw:4d}^2 {ci:4d}->{co:4d}  {t*1e3:8.1f} us  {fl/t/1e9:8.1f} TFLOP/s")
            t = timeit(lambda: ops.conv_dgrad(dy, wT, dx), a.iters)
            print(f"dgrad  {hw:4d}^2 {co:4d}->{ci:4d}  {t*1e3:8.1f} us  {2*fl/t/1e9:8.1f} TFLOP/s")
        if "tn" in only:
            t = timeit(lambda: ops.conv_wgrad(dy, x, dW, nsets=2), a.iters)
            print(f"wgrad  {hw:4d}^2 {ci:4d}x{co:4d}  {t*1e3:8.1f} us  {2*fl/t/1e9:8.1f} TFLOP/s")
        if "gn" in only:
            G = 32
            gamma, beta = torch.ones(ci, device=dev), torch.zeros(ci, device=dev)
            mean, rstd = torch.zeros(B, G, device=dev), torch.ones(B, G, device=dev)
            part = torch.zeros(lib.query("siss_gn_partial_words", 2 * B, hw, hw, ci, G), device=dev)
            yy = Act(B, hw, hw, ci, dev)
            dyy = Act(2 * B, hw, hw, ci, dev); dyy.interior().normal_()
            dxx = Act(2 * B, hw, hw, ci, dev)
            dg = torch.zeros(2, ci, device=dev); db = torch.zeros(2, ci, device=dev)
            xb = x.rows * ci * 2 / 1e9
            t = timeit(lambda: lib.call("siss_groupnorm_fwd", x.data, gamma, beta, yy.data, mean, rstd, part, B, hw, hw, ci, G, 1e-6, 1, 0), a.iters)
            print(f"gn fwd {hw:4d}^2 C={ci:4d}     {t*1e3:8.1f} us  {3*xb/t*1e3:8.1f} GB/s (3X)")
            t = timeit(lambda: lib.call("siss_groupnorm_bwd", dyy.data, x.data, gamma, beta, mean, rstd, dxx.data, None, None, None, 0, 0, dg, db, None, 0, part,
                                        2 * B, B, B, ci, hw, hw, ci, G, 1, 0), a.iters)
            print(f"gn bwd {hw:4d}^2 C={ci:4d}     {t*1e3:8.1f} us  {8*xb/t*1e3:8.1f} GB/s (8X)")


if __name__ == "__main__":
    main()
