#!/usr/bin/env python
"""Sub-pixel form of Upsample2D (nearest 2x -> conv3x3) against today's form, timed launch by launch at the step's top two sites
("priced twice and never run -- build the phase dgrad / wgrad and measure").

nearest-2x followed by a 3x3 'same' conv is four 2x2-tap PHASE convolutions on the LOW-resolution input (output pixel (2Y + py,
2X + px) reads low-resolution rows {Y - 1 + py, Y + py} and columns {X - 1 + px, X + px}; the phase weights are sums of the 3x3
taps): 16 instead of 36 tap products per low-resolution pixel.  Every piece of that form exists as a product launcher, so its cost
can be measured WITHOUT writing a kernel -- with the data movement it needs on today's kernels:
  forward : 4 x siss_gemm_nt_d2s (4 panels on the low-res rows, depth-to-space scatter epilogue)   [+ the consumer GroupNorm's own
            statistics pass: the persistent 3x3 kernel's epilogue statistics are lost]
  dgrad   : space-to-depth of the hi-res cotangent (siss_space_to_depth) + 16 (plane, tap) panels = 2 x siss_gemm_nt of 8 panels
  wgrad   : 4 x siss_gemm_tn (one per plane: Y = the plane's columns of the space-to-depth cotangent, 4 shifted X panels)
            [+ a fold of the 16 phase-tap gradients onto the 9 taps: 16 C^2 floats, negligible]
against: siss_upsample2x + conv3x3 fprop (persistent kernel, statistics in the epilogue); conv3x3 dgrad + siss_upsample2x_bwd;
the fused 3-tap wgrad.  Random data, B = 16 (cotangents 2B).  python tools/probes/subpixel_upsample.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from siss_amd import lib, ops          # noqa: E402
from siss_amd.layout import Act       # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib.load(); lib.ensure_workspace(dev)
    B, it = 16, 10
    print("site (low -> high res, C) | form      | forward us | dgrad us | wgrad us | total us")
    for (lo, C) in ((128, 128), (64, 256)):
        hi = 2 * lo
        x = Act(B, lo, lo, C, dev); x.interior().normal_()
        u = Act(B, hi, hi, C, dev)
        y = Act(B, hi, hi, C, dev)
        dy = Act(2 * B, hi, hi, C, dev); dy.interior().normal_()
        du = Act(2 * B, hi, hi, C, dev)
        dxl = Act(2 * B, lo, lo, C, dev)
        w = (torch.randn(9, C, C, device=dev) / (3 * C ** 0.5))
        wb, wT = w.to(torch.bfloat16), ops.dgrad_weight(w)
        bias = torch.zeros(C, device=dev)
        dW = torch.zeros(2, 9, C, C, device=dev)
        qs = torch.zeros(lib.query("siss_conv_qstats_words", u.rows, C), device=dev)
        # ---- today's form
        f_now = timeit(lambda: (lib.call("siss_upsample2x", x.data, u.data, B, lo, lo, C), ops.conv_fprop_qstats(u, wb, y, qs, bias=bias)), it)
        d_now = timeit(lambda: (ops.conv_dgrad(dy, wT, du), lib.call("siss_upsample2x_bwd", du.data, dxl.data, 2 * B, lo, lo, C)), it)
        w_now = timeit(lambda: ops.conv_wgrad(dy, u, dW, nsets=2), it)
        # ---- sub-pixel form on today's product launchers
        wp = lo + 2
        w4 = (torch.randn(4, 4, C, C, device=dev) / (2 * C ** 0.5)).to(torch.bfloat16)          # [plane][tap][Cout][Cin]
        def phase_shifts(py, px):
            return [(a + py - 1) * wp + (b + px - 1) for a in range(2) for b in range(2)]

        def fwd_sub():
            for pl in range(4):
                sh = phase_shifts(pl >> 1, pl & 1)
                lib.call("siss_gemm_nt_d2s", x.data, C, w4[pl], y.data, C, None, 0, x.rows, C, C, 4, lib.int_array(sh),
                         lib.int_array([0] * 4), x.rows_per_image, x.hp, x.wp, pl)
        f_sub = timeit(fwd_sub, it)
        # the consumer GroupNorm's statistics pass that the epilogue statistics would have saved (two-pass forward minus apply-only)
        G = 32
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        mean, rstd = torch.zeros(B, G, device=dev), torch.ones(B, G, device=dev)
        part = torch.zeros(lib.query("siss_gn_partial_words", B, hi, hi, C, G), device=dev)
        yy = Act(B, hi, hi, C, dev)
        ops.conv_fprop_qstats(u, wb, y, qs, bias=bias)
        g2 = timeit(lambda: lib.call("siss_groupnorm_fwd_ld", y.data, gamma, beta, yy.data, mean, rstd, part, B, hi, hi, C, G, 1e-6, 1, 0, 0), it)
        g1 = timeit(lambda: lib.call("siss_groupnorm_fwd_qs", y.data, gamma, beta, yy.data, mean, rstd, part, qs, C, None, B, hi, hi, C, G,
                                     1e-6, 1, 0, 0), it)
        z = Act(2 * B, lo, lo, 4 * C, dev)
        wd = (torch.randn(16, C, C, device=dev) / (4 * C ** 0.5)).to(torch.bfloat16)
        sh16 = [-(s) for pl in range(4) for s in phase_shifts(pl >> 1, pl & 1)]
        co16 = [pl * C for pl in range(4) for _ in range(4)]

        def dgrad_sub():
            lib.call("siss_space_to_depth", dy.data, z.data, 2 * B, hi, hi, C)
            ops.gemm_nt(lib.ptr(z.data), 4 * C, wd[:8], lib.ptr(dxl.data), C, z.rows, C, C, sh16[:8], co16[:8],
                        rows_per_image=z.rows_per_image, hp=z.hp, wp=z.wp)
            ops.gemm_nt(lib.ptr(z.data), 4 * C, wd[8:], lib.ptr(dxl.data), C, z.rows, C, C, sh16[8:], co16[8:],
                        res_ptr=lib.ptr(dxl.data), ldr=C, rows_per_image=z.rows_per_image, hp=z.hp, wp=z.wp)
        d_sub = timeit(dgrad_sub, it)
        dW4 = torch.zeros(2, 4, 4, C, C, device=dev)
        zp = ops.zero_page(dev)
        rps = B * z.rows_per_image
        rb, re = z.wp + 1, rps - (z.wp + 1)

        def wgrad_sub():                                                   # (reads the space-to-depth cotangent dgrad_sub wrote)
            for pl in range(4):
                sh = phase_shifts(pl >> 1, pl & 1)
                lib.call("siss_gemm_tn", z.data[:, pl * C:], 4 * C, x.data, C, dW4[:, pl], dW4[0].numel(), C, C, 4, lib.int_array(sh),
                         lib.int_array([0] * 4), 2, rps, 0, rb, re, 0, zp, None, None)
        w_sub = timeit(wgrad_sub, it)
        print(f" {lo:3d} -> {hi:3d}, C = {C:3d}       | today     | {f_now * 1e3:10.1f} | {d_now * 1e3:8.1f} | {w_now * 1e3:8.1f} | {(f_now + d_now + w_now) * 1e3:8.1f}")
        print(f"                           | sub-pixel | {f_sub * 1e3:7.1f}+{(g2 - g1) * 1e3:.0f} | {d_sub * 1e3:8.1f} | {w_sub * 1e3:8.1f} | "
              f"{(f_sub + (g2 - g1) + d_sub + w_sub) * 1e3:8.1f}   (+N = the GroupNorm statistics pass the epilogue statistics save today)")


if __name__ == "__main__":
    main()
