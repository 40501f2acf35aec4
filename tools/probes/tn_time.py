"""Time the one-panel weight-gradient products of the SD v1.5 transformer blocks (two cotangent sets sharing X) through
siss_gemm_tn_grouped, ALL jobs of a level in ONE call (as the step queues them), with the producer / consumer kernel on (default) and
off.  python tools/probes/tn_time.py [B]
Needs a variant library built from tools/probes/gemm_tn_pc.hip (SISS_LIB_PATH); the product library has no such setter."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from siss_amd import lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
lib.load(); lib.ensure_workspace("cuda:0")
zp = ops.zero_page(dev)
z9 = (lib.I * 9)(*([0] * 9))
keep = []


def job(rows, N, C):
    y = torch.randn(2 * rows, N, device=dev).to(torch.bfloat16)
    x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    dW = torch.zeros(2, N * C + N, device=dev)
    keep.append((y, x, dW))
    return lib.TNJob(Y=y.data_ptr(), ldy=N, X=x.data_ptr(), ldx=C, dW=dW.data_ptr(), set_stride=dW.shape[1], N=N, C=C, npanels=1,
                     nsets=2, rows_per_set=rows, row_begin=0, row_end=rows, nsplits=0, x_set_rows=0, zero_page=zp.data_ptr(),
                     dbias=None, dbias2=None, shifts=z9, coffs=z9), 2.0 * 2 * rows * N * C


for S, C in ((4096, 320), (1024, 640), (256, 1280)):
    jobs, fl = zip(*[job(B * S, n, c) for n, c in ((3 * C, C), (C, C), (C, C), (C, C), (8 * C, C), (C, 4 * C), (C, C), (C, C))])
    arr = (lib.TNJob * len(jobs))(*jobs)
    t = []
    for pc in (True, False):
        lib.query("siss_gemm_tn_set_pc_min_rows", 4096 if pc else 1 << 30)
        for _ in range(3):
            lib.call("siss_gemm_tn_grouped", arr, len(jobs))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            lib.call("siss_gemm_tn_grouped", arr, len(jobs))
        e.record(); torch.cuda.synchronize()
        t.append(s.elapsed_time(e) / 10 * 1e3)
    lib.query("siss_gemm_tn_set_pc_min_rows", 4096)
    print(f"B {B} S {S} C {C}: 8 jobs of a transformer block: pc {t[0]:8.1f} us ({sum(fl) / t[0] / 1e6:6.0f} TF/s)   128x128 {t[1]:8.1f} us ({sum(fl) / t[1] / 1e6:6.0f} TF/s)", flush=True)
    keep.clear()
