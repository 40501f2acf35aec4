"""Cost of the GroupNorm-statistics hand-over in the persistent 3x3 kernel: the same convolution with and without
NTParams::qstats (HIP events, 30 launches each).  SISS_NT_ABLATE=32 skips the per-tile fold + store of the entries,
64 the per-row accumulation (probes: results are then wrong)."""
import sys
import torch
sys.path.insert(0, ".")
from siss_amd import lib, ops
from siss_amd.layout import Act

lib.load(); lib.ensure_workspace("cuda:0")
dev = torch.device("cuda:0")
for (n, h, ci, co, res) in [(16, 256, 128, 128, True), (16, 256, 128, 128, False), (16, 256, 256, 128, False), (16, 128, 128, 128, True), (16, 64, 256, 256, True)]:
    x = Act.from_nchw(torch.randn(n, ci, h, h).bfloat16().float(), dev)
    w = (torch.randn(9, co, ci) * 0.03).to(dev).to(torch.bfloat16)
    out = Act(n, h, h, co, dev)
    r = Act.from_nchw(torch.randn(n, co, h, h), dev) if res else None
    bias = torch.randn(co, device=dev)
    qs = torch.zeros(lib.query("siss_conv_qstats_words", x.rows, co), device=dev)
    def run(q):
        if q:
            assert ops.conv_fprop_qstats(x, w, out, qs, bias=bias, residual=r)
        else:
            ops.conv_fprop(x, w, out, bias=bias, residual=r)
    ts = []
    for q in (False, True, False, True):
        for _ in range(3): run(q)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30): run(q)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 30 * 1e3)
    fl = 2.0 * n * h * h * co * ci * 9
    print(f"n {n} h {h} ci {ci} co {co} res {int(res)}: plain {ts[0]:.1f} / {ts[2]:.1f} us ({fl / ts[2] * 1e-6:.0f} TF/s)   qstats {ts[1]:.1f} / {ts[3]:.1f} us  (+{(ts[3] / ts[2] - 1) * 100:.1f} %)")
