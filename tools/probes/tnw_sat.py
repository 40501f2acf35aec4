"""Saturated single-job timing of the one-panel weight-gradient kernels: rows 65536, N 2048, C 768 (or argv[1]), two sets.
Needs a VARIANT library built from tools/probes/gemm_tn_pc.hip (or gemm_tn_wide.hip with its setter name) and loaded through
SISS_LIB_PATH: the product library has no siss_gemm_tn_set_pc_min_rows.  Record of docs/experiments.md's rejected kernels."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from siss_amd import lib, ops
dev = torch.device("cuda:0")
lib.load(); lib.ensure_workspace("cuda:0")
zp = ops.zero_page(dev); z9 = (lib.I * 9)(*([0] * 9))
rows, N, C = 65536, 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 768
y = torch.randn(2 * rows, N, device=dev).to(torch.bfloat16); x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
dW = torch.zeros(2, N * C + N, device=dev)
job = lib.TNJob(Y=y.data_ptr(), ldy=N, X=x.data_ptr(), ldx=C, dW=dW.data_ptr(), set_stride=dW.shape[1], N=N, C=C, npanels=1, nsets=2,
                rows_per_set=rows, row_begin=0, row_end=rows, nsplits=0, x_set_rows=0, zero_page=zp.data_ptr(), dbias=None, dbias2=None, shifts=z9, coffs=z9)
arr = (lib.TNJob * 1)(job)
for wide in (True, False):
    lib.query("siss_gemm_tn_set_pc_min_rows", 4096 if wide else 1 << 30)
    for _ in range(3): lib.call("siss_gemm_tn_grouped", arr, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): lib.call("siss_gemm_tn_grouped", arr, 1)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10 * 1e3
    print(f"{os.environ.get('SISS_LIB_PATH', 'product')}: {'wide' if wide else '128x128'} {t:8.1f} us {2.0 * 2 * rows * N * C / t / 1e6:6.0f} TF/s", flush=True)
