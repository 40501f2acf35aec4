"""dgrad launches of the persistent 3x3 kernel (no bias / row bias / residual): HIP events over 30 launches."""
import sys
import torch
sys.path.insert(0, ".")
from siss_amd import lib, ops
from siss_amd.layout import Act

lib.load(); lib.ensure_workspace("cuda:0")
dev = torch.device("cuda:0")
for (n, h, ci, co) in [(32, 256, 128, 128), (32, 256, 128, 256), (32, 128, 128, 128), (32, 64, 256, 256)]:
    dy = Act.from_nchw(torch.randn(n, co, h, h).bfloat16().float(), dev)
    wT = (torch.randn(9, ci, co) * 0.03).to(dev).to(torch.bfloat16)
    dx = Act(n, h, h, ci, dev)
    ts = []
    for rep in range(3):
        for _ in range(3): ops.conv_dgrad(dy, wT, dx)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30): ops.conv_dgrad(dy, wT, dx)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 30 * 1e3)
    fl = 2.0 * n * h * h * co * ci * 9
    print(f"dgrad n {n} h {h} {co} -> {ci}: {ts[1]:.1f} / {ts[2]:.1f} us ({fl / ts[2] * 1e-6:.0f} TF/s)")
