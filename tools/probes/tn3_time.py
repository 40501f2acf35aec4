"""Time the fused 3-tap weight-gradient kernel on the CelebA-HQ top-level shapes.  A/B builds through SISS_LIB_PATH."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from siss_amd import lib, ops
from siss_amd.layout import Act, conv3x3_panels
dev = torch.device("cuda:0"); lib.load(); lib.ensure_workspace("cuda:0")
zp = ops.zero_page(dev)
for (n, hw, ci, co) in ((16, 256, 128, 128), (16, 256, 256, 128), (16, 128, 256, 256), (16, 64, 256, 256)):
    x = Act(n, hw, hw, ci, dev); dy = Act(2 * n, hw, hw, co, dev)
    x.data.normal_(); dy.data.normal_()
    shifts, coffs = conv3x3_panels(dy.wp, ci)
    rps = n * dy.rows_per_image
    rb, re = dy.wp + 1, rps - (dy.wp + 1)
    dW = torch.zeros(2, 9 * co * ci, device=dev)
    f = lambda: lib.call("siss_gemm_tn", dy.data, co, x.data, ci, dW, dW.shape[1], co, ci, 9, lib.int_array(shifts), lib.int_array(coffs),
                         2, rps, 0, rb, re, 0, zp, None, None)
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10 * 1e3
    fl = 2.0 * 2 * n * hw * hw * 9 * ci * co
    print(f"wgrad {hw}x{hw} {ci}->{co}: {t:8.1f} us  {fl / t / 1e6:6.0f} TF/s (true pixels)", flush=True)
# one-panel products (transformer linears / 1x1 shortcuts): rows, N, C
for (rows, N, C) in ((65536, 2048, 768), (65536, 320, 320), (16384, 1280, 1280), (1064506 // 2, 128, 256)):
    y = torch.randn(2 * rows, N, device=dev).to(torch.bfloat16); x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    dW = torch.zeros(2, N * C, device=dev)
    z9 = (lib.I * 9)(*([0] * 9))
    job = lib.TNJob(Y=y.data_ptr(), ldy=N, X=x.data_ptr(), ldx=C, dW=dW.data_ptr(), set_stride=dW.shape[1], N=N, C=C, npanels=1, nsets=2,
                    rows_per_set=rows, row_begin=0, row_end=rows, nsplits=0, x_set_rows=0, zero_page=zp.data_ptr(), dbias=None, dbias2=None,
                    shifts=z9, coffs=z9)
    arr = (lib.TNJob * 1)(job)
    f = lambda: lib.call("siss_gemm_tn_grouped", arr, 1)
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10 * 1e3
    print(f"one-panel rows {rows} N {N} C {C}: {t:8.1f} us  {2.0 * 2 * rows * N * C / t / 1e6:6.0f} TF/s", flush=True)
