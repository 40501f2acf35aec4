"""Probe: siss_gemm_tn_pair (a one-panel weight gradient inside a 3-tap weight gradient's launch) against the two launches apart,
at the CelebA-HQ top-resolution shapes (B = 16, 256 x 256, two cotangent sets), over the balance knob siss_gemm_tn_set_pair_cost
(relative cost of a one-tap K-step, permille).  Prints us per launch (HIP events, median of 5).

    python tools/probes/tn_pair.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from siss_amd import lib, ops                      # noqa: E402
from siss_amd.layout import Act, conv3x3_panels    # noqa: E402


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    lib.load()
    n, hw, nsets = 16, 256, 2
    zp = ops.zero_page(dev)
    g = torch.Generator(device=dev).manual_seed(1)

    def act(nn, c):
        a = Act(nn, hw, hw, c, dev)
        a.interior().copy_(torch.randn(nn, hw, hw, c, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16))
        return a
    dy = act(nsets * n, 128)
    col = act(nsets * n, 64)
    xs = {128: act(n, 128), 256: act(n, 256)}
    rps = n * dy.rows_per_image
    rb, re = dy.wp + 1, rps - (dy.wp + 1)
    z9 = (lib.I * 9)(*([0] * 9))
    for (c3, n1, c1, ycol) in ((128, 128, 256, False), (256, 128, 256, False), (128, 27, 128, True), (128, 128, 64, False)):
        s3, cf3 = conv3x3_panels(dy.wp, c3)
        dW3 = torch.zeros(nsets, 9 * 128 * c3, device=dev)
        dW1 = torch.zeros(nsets, n1 * c1, device=dev)
        y1 = col if ycol else dy
        x1 = xs[c1] if c1 in xs else col
        j3 = lib.TNJob(Y=dy.data.data_ptr(), ldy=128, X=xs[c3].data.data_ptr(), ldx=c3, dW=dW3.data_ptr(), set_stride=dW3.shape[1], N=128, C=c3,
                       npanels=9, nsets=nsets, rows_per_set=rps, row_begin=rb, row_end=re, nsplits=0, x_set_rows=0, zero_page=zp.data_ptr(),
                       dbias=None, dbias2=None, shifts=(lib.I * 9)(*s3), coffs=(lib.I * 9)(*cf3))
        j1 = lib.TNJob(Y=y1.data.data_ptr(), ldy=y1.c, X=x1.data.data_ptr(), ldx=x1.c, dW=dW1.data_ptr(), set_stride=dW1.shape[1], N=n1, C=c1,
                       npanels=1, nsets=nsets, rows_per_set=rps, row_begin=rb, row_end=re, nsplits=0, x_set_rows=0, zero_page=zp.data_ptr(),
                       dbias=None, dbias2=None, shifts=z9, coffs=z9)
        t3 = timed(lambda: lib.call("siss_gemm_tn", dy.data, 128, xs[c3].data, c3, dW3, dW3.shape[1], 128, c3, 9, lib.int_array(s3),
                                    lib.int_array(cf3), nsets, rps, 0, rb, re, 0, zp, None, None))
        t1 = timed(lambda: lib.call("siss_gemm_tn", y1.data, y1.c, x1.data, x1.c, dW1, dW1.shape[1], n1, c1, 1, lib.int_array([0]),
                                    lib.int_array([0]), nsets, rps, 0, rb, re, 0, zp, None, None))
        line = f"3-tap N 128 C {c3}: {t3:7.1f} us | one-panel N {n1} C {c1}: {t1:6.1f} us | apart {t3 + t1:7.1f} | paired, cost permille:"
        for pm in (700, 900, 1000, 1100, 1200, 1300, 1500, 1800):
            lib.query("siss_gemm_tn_set_pair_cost", pm)
            tp = timed(lambda: lib.call("siss_gemm_tn_pair", lib.C.byref(j3), lib.C.byref(j1), 0))
            line += f" {pm}: {tp:6.1f}"
        print(line, flush=True)
    lib.query("siss_gemm_tn_set_pair_cost", 0)


if __name__ == "__main__":
    main()
