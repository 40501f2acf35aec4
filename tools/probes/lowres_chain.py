#!/usr/bin/env python
"""What could a persistent, weight-prefetching kernel for the <= 16 x 16 chain buy?  (one measured attempt.)

The proposal: ONE persistent kernel per low-resolution resnet (GN -> conv -> GN -> conv, 512 channels, B = 16) that prefetches the
next layer's weights during the current layer's drain and replaces the kernel boundaries by an XCD-local barrier.  Its gain has two
possible sources, and both can be measured WITHOUT building it, on the product kernels:

  (1) weight prefetch: a layer of the step meets its weights COLD (4.7 MB of bf16 per 512 -> 512 3x3 conv, last touched a step ago,
      ~1 GB of other traffic in between).  Upper bound of what prefetching can hide = time of the layer's launches with cold caches
      minus their time with the weights (and activations) warm in L2 / Infinity Cache.  Measured: the same 4-launch forward chain of
      one 8 x 8 (and one 16 x 16) resnet replayed from a hipGraph back to back (warm) and after a 1 GiB write that evicts L2 and the
      256 MiB Infinity Cache (cold), HIP events around the chain only.
  (2) kernel boundaries: chain time (one graph replay of the 4 launches) minus the sum of the 4 launches' own durations measured
      each alone back to back.  That difference is everything a persistent kernel's barrier would have to beat.

Prints one table; copy it to profiles/ when quoting.  python tools/probes/lowres_chain.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from siss_amd import lib, ops          # noqa: E402
from siss_amd.layout import Act       # noqa: E402


def ev_time(fn, iters, before=None):
    ts = []
    for _ in range(iters):
        if before is not None:
            before()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    lib.load(); lib.ensure_workspace(dev)
    B, C, G = 16, 512, 32
    evict = torch.empty(1 << 28, dtype=torch.float32, device=dev)           # 1 GiB
    rows = []
    for hw in (8, 16):
        x = Act(B, hw, hw, C, dev); x.interior().normal_()
        a1, h, a2, out = (Act(B, hw, hw, C, dev) for _ in range(4))
        # 8 distinct weight sets: a "layer" of the real step never meets the weights of the previous launch of the same shape
        ws = [(torch.randn(9, C, C, device=dev) / (3 * C ** 0.5)).to(torch.bfloat16) for _ in range(8)]
        bias = torch.zeros(C, device=dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        mean, rstd = torch.zeros(B, G, device=dev), torch.ones(B, G, device=dev)
        part = torch.zeros(lib.query("siss_gn_partial_words", B, hw, hw, C, G), device=dev)

        def gn(src, dst):
            lib.call("siss_groupnorm_fwd", src.data, gamma, beta, dst.data, mean, rstd, part, B, hw, hw, C, G, 1e-6, 1, 0)
        steps = [lambda: gn(x, a1), lambda: ops.conv_fprop(a1, ws[0], h, bias=bias), lambda: gn(h, a2),
                 lambda: ops.conv_fprop(a2, ws[1], out, bias=bias, residual=x)]

        def chain():
            for f in steps:
                f()
        chain(); torch.cuda.synchronize()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            chain()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                chain()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        warm = ev_time(graph.replay, 30)
        cold = ev_time(graph.replay, 15, before=lambda: evict.fill_(1.0))
        alone = []
        for f in steps:
            f(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(30):
                f()
            e.record(); torch.cuda.synchronize()
            alone.append(s.elapsed_time(e) * 1e3 / 30)
        flop = 2 * 2.0 * B * (hw + 2) ** 2 * C * C * 9
        rows.append((hw, warm, cold, sum(alone), alone, flop))
    print("forward chain of ONE resnet (GN+SiLU -> conv3x3 -> GN+SiLU -> conv3x3 + residual), 512 channels, B = 16, bf16")
    print(" grid   | chain warm | chain cold (after a 1 GiB write) | sum of the 4 launches alone (back to back) | alone: gn conv gn conv | conv flops at warm-chain time")
    for hw, warm, cold, tot, alone, flop in rows:
        print(f" {hw:2d} x {hw:2d} | {warm:8.1f} us | {cold:8.1f} us                      | {tot:8.1f} us"
              f"                                | " + " ".join(f"{t:6.1f}" for t in alone) + f" | {flop / warm / 1e6:6.0f} TF/s")
    for hw, warm, cold, tot, alone, flop in rows:
        print(f" {hw:2d} x {hw:2d}: weight / activation prefetch could hide at most cold - warm = {cold - warm:6.1f} us of {cold:6.1f} "
              f"({(cold - warm) / cold * 100:4.1f} %); kernel boundaries cost chain - sum(alone) = {warm - tot:6.1f} us")


if __name__ == "__main__":
    main()
