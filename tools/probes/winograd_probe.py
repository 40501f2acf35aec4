#!/usr/bin/env python
"""PROBE: Winograd F(2x2, 3x3) at ONE site of the CelebA-HQ step -- 128 -> 128 channels at 256 x 256, forward,
B = 16 -- measured against the persistent direct-convolution kernel (gemm_nt_c3p) on the same tensors.  Go = >= 1.35x faster with
activations within 1e-2 of scale of the f32 convolution; no-go = both numbers go to docs/experiments.md.

Measured form: UNFUSED -- input transform (tools/probes/winograd_f2x2.hip: HBM-bound, x read once, V = 4x the input written), the
16 per-frequency products as ONE batched launch of the product NT GEMM (M = 262,144 tiles, N = K = 128: 137 GFLOP instead of 309),
output transform (reads the 16 product planes, writes y).  Why not a fused kernel: see the arithmetic in docs/experiments.md
(round 5) -- per 64-channel step a block needs the 16 transformed weight tiles (16 x N x 64 x 2 B) AND the 16 transformed input
tiles (16 x T x 64 x 2 B) in LDS and 16 T N f32 accumulators in registers; with 160 KiB of LDS and 512 KiB of registers per CU the
largest tile (T = N = 64, 32-channel steps) moves 84 KB from L2 per 1024 MFMA cycles = 82 B / clk / CU against the ~32 B / clk / CU the
L2 -> LDS path delivers (the direct kernel needs 26): the 2.25x fewer MACs are paid back 2.5x in operand delivery.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/winograd_f2x2.hip -o tools/probes/_probe_build/libwino.so
    python tools/probes/winograd_probe.py
"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from siss_amd import lib, ops          # noqa: E402
from siss_amd.layout import Act       # noqa: E402


def timed(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    so = os.path.join(ROOT, "tools", "probes", "_probe_build", "libwino.so")
    if not os.path.exists(so):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                               os.path.join(ROOT, "tools", "probes", "winograd_f2x2.hip"), "-o", so])
    w = C.CDLL(so)
    dev = torch.device("cuda:0")
    lib.load(); lib.ensure_workspace(dev)
    B, H, Ci, Co = 16, 256, 128, 128
    g = torch.Generator(device=dev).manual_seed(0)
    xf = torch.randn(B, Ci, H, H, device=dev, generator=g)
    x = Act.from_nchw(xf, dev)
    wt = torch.randn(Co, Ci, 3, 3, device=dev, generator=g) / (3 * Ci ** 0.5)
    bias = 0.1 * torch.randn(Co, device=dev, generator=g)
    wn = ops.conv_w_to_native(wt).to(torch.bfloat16)                # [9][Co][Ci]
    ref = torch.nn.functional.conv2d(x.to_nchw(), wn.float().view(3, 3, Co, Ci).permute(2, 3, 0, 1), bias, padding=1)
    scale = float(ref.abs().max())

    # ---- direct: the persistent 3x3 kernel
    y = Act(B, H, H, Co, dev)
    direct = lambda: ops.conv_fprop(x, wn, y, bias=bias)
    lib.dispatch_counts(reset=True)
    t_direct = timed(direct)
    assert lib.dispatch_counts()["gemm_nt_c3p_kernel"] > 0
    e_direct = float((y.to_nchw() - ref).abs().max()) / scale

    # ---- Winograd F(2x2, 3x3), unfused
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], device=dev)
    gk = wn.float().view(3, 3, Co, Ci)                               # the bf16-rounded weights both forms use
    U = torch.einsum("ai,ijoc,bj->aboc", G, gk, G).reshape(16, Co, Ci).to(torch.bfloat16).contiguous()   # f32 sums, ONE rounding
    T = B * (H // 2) * (H // 2)
    V = torch.empty(16, T, Ci, dtype=torch.bfloat16, device=dev)
    M = torch.empty(16, T, Co, dtype=torch.bfloat16, device=dev)
    y2 = Act(B, H, H, Co, dev)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    f_in = lambda: w.wino_input(P(x.data), P(V), B, H, H, Ci, st())
    f_mm = lambda: ops.gemm_nt(lib.ptr(V), Ci, U, lib.ptr(M), Co, T, Co, Ci, [0], [0], batch=16, stride_a=T * Ci, stride_w=Co * Ci,
                               stride_c=T * Co)
    f_out = lambda: w.wino_output(P(M), P(bias), P(y2.data), B, H, H, Co, st())
    t_in, t_mm, t_out = timed(f_in), timed(f_mm), timed(f_out)
    f_in(); f_mm(); f_out(); torch.cuda.synchronize()
    e_wino = float((y2.to_nchw() - ref).abs().max()) / scale
    t_all = timed(lambda: (f_in(), f_mm(), f_out()))
    gb = lambda n: n / 1e9
    print(f"site: {Ci} -> {Co} channels, {H} x {H}, B = {B}, forward")
    print(f"direct (gemm_nt_c3p)        : {t_direct:7.1f} us   ({2 * B * H * H * 9 * Ci * Co / t_direct * 1e-6:6.0f} TF/s algorithmic)   max err {e_direct:.2e} of scale")
    print(f"Winograd F(2x2,3x3) unfused : {t_all:7.1f} us   = input transform {t_in:.1f} ({gb(x.data.numel() * 2 + V.numel() * 2) / t_in * 1e6:.2f} TB/s)"
          f" + 16 batched products {t_mm:.1f} ({2 * 16 * T * Ci * Co / t_mm * 1e-6:.0f} TF/s, {gb((V.numel() + M.numel()) * 2) / t_mm * 1e6:.2f} TB/s)"
          f" + output transform {t_out:.1f} ({gb(M.numel() * 2 + y2.data.numel() * 2) / t_out * 1e6:.2f} TB/s)   max err {e_wino:.2e} of scale")
    print(f"speed-up over the direct kernel: {t_direct / t_all:.2f}x (go needs >= 1.35x)   bytes through HBM: direct {gb((x.data.numel() + y.data.numel()) * 2):.2f} GB, "
          f"Winograd unfused {gb((x.data.numel() + 2 * V.numel() + 2 * M.numel() + y2.data.numel()) * 2):.2f} GB")


if __name__ == "__main__":
    main()
