"""Reference point: what torch.matmul (hipBLASLt / rocBLAS) reaches on plain bf16 GEMMs of the same shapes as the two
dominant kernels of the step (no halo, no taps, no fused epilogue -- an upper bound for a library call):
  fprop / dgrad 128->128 @256^2 :  [M, 1152] x [1152, 128]   (M = 1.06 M / 2.13 M rows)
  wgrad 128x128 @256^2 (one set):  [128, M]  x [M, 1152]
"""
import torch

def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it

dev = "cuda:0"
for M in (1065024, 2130048):
    a = torch.randn(M, 1152, device=dev, dtype=torch.bfloat16)
    w = torch.randn(1152, 128, device=dev, dtype=torch.bfloat16)
    ms = t(lambda: a @ w)
    print(f"NT-like  [{M} x 1152] x [1152 x 128]  {ms*1e3:8.1f} us  {2*M*1152*128/ms/1e9:8.1f} TFLOP/s")
    y = torch.randn(M, 128, device=dev, dtype=torch.bfloat16)
    ms = t(lambda: y.t() @ a)
    print(f"TN-like  [128 x {M}] x [{M} x 1152]  {ms*1e3:8.1f} us  {2*M*1152*128/ms/1e9:8.1f} TFLOP/s")
    del a, y
for n in (4096, 8192):
    a = torch.randn(n, n, device=dev, dtype=torch.bfloat16); b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    ms = t(lambda: a @ b.t())
    print(f"square   {n}^3 (A B^T)                  {ms*1e3:8.1f} us  {2*n**3/ms/1e9:8.1f} TFLOP/s")
