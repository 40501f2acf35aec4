"""Read / copy bandwidth against working-set size: does the 256 MB Infinity Cache (MALL) deliver more than HBM when a
tensor is re-read right after it was read?  (decides whether sample-chunked GroupNorm passes can win)"""
import torch
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    a = torch.randn(n, device=dev); b = torch.empty_like(a)
    t_r = timeit(lambda: a.sum())
    t_c = timeit(lambda: torch.add(a, 1.0, out=b))
    print(f"{mb:5d} MB  read {mb / 1024 / t_r / 1.024:7.2f} TB/s ({t_r*1e6:7.1f} us)   read+write {2 * mb / 1024 / t_c / 1.024:7.2f} TB/s ({t_c*1e6:7.1f} us)")
