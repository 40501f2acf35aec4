"""GroupNorm backward at the 256 x 256 sites: ONE launch pair over 16 samples (x + two cotangent sets = 805 MB: the apply pass
re-reads from HBM) against FOUR launch pairs over 4 samples each (201 MB: the apply pass could re-read from the 256 MB
Infinity Cache).  The cache is flushed (a 1 GiB fill) before every timed run, as the real step leaves it."""
import sys
import torch
sys.path.insert(0, ".")
from siss_amd import lib
from siss_amd.layout import Act

lib.load()
dev = torch.device("cuda:0")
G = 32
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
for (hw, ci) in [(256, 128), (256, 256)]:
    res = {}
    for B in (16, 8, 4):
        x = Act(B, hw, hw, ci, dev); x.interior().normal_()
        gamma, beta = torch.ones(ci, device=dev), torch.zeros(ci, device=dev)
        mean, rstd = torch.zeros(B, G, device=dev), torch.ones(B, G, device=dev)
        part = torch.zeros(lib.query("siss_gn_partial_words", 2 * B, hw, hw, ci, G), device=dev)
        dyy = Act(2 * B, hw, hw, ci, dev); dyy.interior().normal_()
        dxx = Act(2 * B, hw, hw, ci, dev)
        dg = torch.zeros(2, ci, device=dev); db = torch.zeros(2, ci, device=dev)
        ts = []
        for it in range(6):
            flush.fill_(float(it))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(16 // B):
                lib.call("siss_groupnorm_bwd", dyy.data, x.data, gamma, beta, mean, rstd, dxx.data, None, None, None, 0, 0, dg, db,
                         None, 0, part, 2 * B, B, B, ci, hw, hw, ci, G, 1, 0)
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        res[B] = sorted(ts)[len(ts) // 2]
        del x, dyy, dxx
    print(f"{hw}^2 C={ci}: 16 samples as " + "  ".join(f"{16 // B} x {B}: {res[B]:.0f} us" for B in (16, 8, 4)))
