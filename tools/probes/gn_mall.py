"""Does a GroupNorm site that fits in the 256 MB Infinity Cache run faster per byte?  (decides whether chunking the
batch so that stats + apply of a chunk run back to back is worth it)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from siss_amd import lib
from siss_amd.layout import Act
from tools.bench_kernels import timeit
dev = torch.device("cuda:0"); lib.load()
G = 32
for (hw, ci) in [(256, 128), (128, 128)]:
    for B in (1, 2, 4, 8, 16):
        x = Act(B, hw, hw, ci, dev); x.interior().normal_()
        gamma, beta = torch.ones(ci, device=dev), torch.zeros(ci, device=dev)
        mean, rstd = torch.zeros(B, G, device=dev), torch.ones(B, G, device=dev)
        part = torch.zeros(lib.query("siss_gn_partial_words", 2 * B, hw, hw, ci, G), device=dev)
        yy = Act(B, hw, hw, ci, dev)
        dyy = Act(2 * B, hw, hw, ci, dev); dyy.interior().normal_()
        dxx = Act(2 * B, hw, hw, ci, dev)
        dg = torch.zeros(2, ci, device=dev); db = torch.zeros(2, ci, device=dev)
        xb = x.rows * ci * 2 / 1e9
        t = timeit(lambda: lib.call("siss_groupnorm_fwd", x.data, gamma, beta, yy.data, mean, rstd, part, B, hw, hw, ci, G, 1e-6, 1, 0), 20)
        t2 = timeit(lambda: lib.call("siss_groupnorm_bwd", dyy.data, x.data, gamma, beta, mean, rstd, dxx.data, None, None, None, 0, 0, dg, db, None, 0, part,
                                     2 * B, B, B, ci, hw, hw, ci, G, 1, 0), 20)
        print(f"{hw}^2 C={ci} B={B:2d}  x={xb*1e3:7.1f} MB  fwd {t*1e3:7.1f} us {3*xb/t*1e3:7.0f} GB/s | bwd {t2*1e3:7.1f} us {8*xb/t2*1e3:7.0f} GB/s  per-image fwd {t*1e3/B:6.1f} bwd {t2*1e3/B:6.1f}")
