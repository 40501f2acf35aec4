"""GroupNorm sites of the CelebA-HQ step (B = 16): two-pass kernels vs the default (slab kernels at the small sites), us per launch and
algorithmic GB/s (x + y forward; x + 2 dy + 2 dx backward)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siss_amd import lib
from siss_amd.layout import Act
from tools.bench_kernels import timeit
dev = torch.device("cuda:0"); lib.load()
G, B = 32, int(os.environ.get("B", "16"))
SITES = [(256, 128), (256, 256), (128, 128), (128, 256), (128, 384), (64, 256), (64, 512), (32, 256), (32, 512), (16, 512), (16, 1024), (8, 512), (8, 1024)]
for (hw, ci) in SITES:
    x = Act(B, hw, hw, ci, dev); x.interior().normal_()
    gamma, beta = torch.ones(ci, device=dev), torch.zeros(ci, device=dev)
    mean, rstd = torch.zeros(B, G, device=dev), torch.ones(B, G, device=dev)
    part = torch.zeros(lib.query("siss_gn_partial_words", 2 * B, hw, hw, ci, G), device=dev)
    yy = Act(B, hw, hw, ci, dev)
    dyy = Act(2 * B, hw, hw, ci, dev); dyy.interior().normal_()
    dxx = Act(2 * B, hw, hw, ci, dev)
    acc = Act(2 * B, hw, hw, ci, dev)
    dg = torch.zeros(2, ci, device=dev); db = torch.zeros(2, ci, device=dev)
    xb = x.rows * ci * 2 / 1e9
    line = f"{hw:3d}^2 C={ci:4d} "
    for mode in (0, 1):
        lib.query("siss_groupnorm_set_slab", mode)
        t = timeit(lambda: lib.call("siss_groupnorm_fwd", x.data, gamma, beta, yy.data, mean, rstd, part, B, hw, hw, ci, G, 1e-6, 1, 0), 20)
        t2 = timeit(lambda: lib.call("siss_groupnorm_bwd", dyy.data, x.data, gamma, beta, mean, rstd, dxx.data, None, None, None, 0, 0, dg, db, None, 0, part,
                                     2 * B, B, B, ci, hw, hw, ci, G, 1, 0), 20)
        t3 = timeit(lambda: lib.call("siss_groupnorm_bwd", dyy.data, x.data, gamma, beta, mean, rstd, dxx.data, acc.data, None, None, 0, 0, dg, db, None, 0, part,
                                     2 * B, B, B, ci, hw, hw, ci, G, 1, 0), 20)
        line += f"| {'2pass' if mode == 0 else 'slab  '} fwd {t*1e3:7.1f} us {2*xb/t*1e3:5.0f} GB/s  bwd {t2*1e3:7.1f} us {5*xb/t2*1e3:5.0f} GB/s  bwd+acc {t3*1e3:7.1f} us "
    print(line, flush=True)
lib.query("siss_groupnorm_set_slab", -1)
