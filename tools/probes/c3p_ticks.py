# needs the probe build: tools/probes/build_probe.sh, then SISS_LIB_PATH=tools/probes/libsiss_hip_probe.so
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device("cuda:0")
dbg = torch.zeros(64, dtype=torch.int64, device=dev)
os.environ["SISS_NT_DEBUG_PTR"] = str(dbg.data_ptr())
from siss_amd import lib, ops
from siss_amd.layout import Act
lib.load()
B, hw, ci, co = 16, 256, 128, 128
x = Act(B, hw, hw, ci, dev); x.interior().normal_()
w = (torch.randn(9, co, ci, device=dev) / (3 * ci ** 0.5)).to(torch.bfloat16)
y = Act(B, hw, hw, co, dev); bias = torch.zeros(co, device=dev)
for _ in range(3):
    ops.conv_fprop(x, w, y, bias=bias)
torch.cuda.synchronize()
t = dbg.cpu().view(8, 8)
names = ["mainloop", "barrierE", "lds_write", "readback+stores"]
for wv in range(4):
    print("wave", wv, {n: int(t[wv, i]) for i, n in enumerate(names)}, "total", int(t[wv, :4].sum()))
