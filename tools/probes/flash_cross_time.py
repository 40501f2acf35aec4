"""Time the fused attention kernels on SD v1.5's cross-attention shapes (77 keys).  python tools/probes/flash_cross_time.py [B]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
PRE = int(os.environ.get('PRE', '0'))      # 1: q pre-scaled by scale * log2(e) (q_prescaled form)
from siss_amd import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
lib.load(); lib.ensure_workspace("cuda:0")
for S, D in ((4096, 40), (1024, 80), (256, 160)):
    H, Sk, sets = 8, 77, 2
    ld = H * D
    mk = lambda n: torch.randn(n, ld, device=dev).to(torch.bfloat16)
    q, k, v, do = mk(B * S), mk(B * Sk), mk(B * Sk), mk(sets * B * S)
    o = torch.empty_like(q); dq = torch.empty_like(do); dk = mk(sets * B * Sk); dv = mk(sets * B * Sk)
    Sp = -(-S // 64) * 64
    lse = torch.zeros(B * H, Sp, device=dev); delta = torch.zeros(sets * B * H * Sp, device=dev)
    sc = D ** -0.5
    f = lambda: lib.call("siss_flash_attn_fwd_merged", q, ld, k, ld, v, ld, o, ld, lse, B, H, S, Sk, D, sc, PRE)
    b = lambda: lib.call("siss_flash_attn_bwd_merged", q, ld, k, ld, v, ld, o, ld, do, ld, lse, delta, dq, ld, dk, ld, dv, ld, sets * B, B, H, S, Sk, D, sc, PRE)
    for fn, nm in ((f, "fwd"), (b, "bwd")):
        for _ in range(3): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        print(f"B {B} Sq {S} Sk {Sk} D {D} {nm}: {s.elapsed_time(e) / 20 * 1e3:8.1f} us", flush=True)
