"""Probe: the fused single-head attention kernels alone at the CelebA-HQ shapes (B = 16 forward images, 32 cotangent images,
D = 512; S = 256 and 64), us per launch from HIP events -- and the body `rocprofv3 --pmc` runs (tools/pmc_attn1h.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from siss_amd import lib                            # noqa: E402
from siss_amd.layout import Act                     # noqa: E402


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    lib.load()
    B, nb, D = 16, 32, 512
    for S, W in ((256, 16), (64, 8)):
        H = S // W
        qkv = (torch.randn(B * S, 3 * D, device=dev) * 1.0).to(torch.bfloat16)
        o, do = Act(B, H, W, D, dev), Act(nb, H, W, D, dev)
        do.interior().copy_(torch.randn(nb, H, W, D, device=dev).to(torch.bfloat16))
        lse, delta = torch.zeros(B * S, device=dev), torch.zeros(nb * S, device=dev)
        dqkv = torch.zeros(nb * S, 3 * D, dtype=torch.bfloat16, device=dev)
        sc = D ** -0.5
        fwd = lambda: lib.call("siss_attn1h_fwd", qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o.data, D, W, lse, B, S, D, sc)
        bwd = lambda: lib.call("siss_attn1h_bwd", qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o.data, D, do.data, D, W, lse, delta,
                               dqkv, dqkv[:, D:], dqkv[:, 2 * D:], 3 * D, nb, B, S, D, sc)
        tf, tb = timed(fwd), timed(bwd)
        gf = 2.0 * 2 * B * S * S * D
        print(f"S {S:4d}: fwd {tf:6.1f} us ({gf / tf * 1e-6:6.1f} TF/s)   bwd (dq + dkv) {tb:6.1f} us ({2.5 * 2 * gf / tb * 1e-6:6.1f} TF/s algorithmic)", flush=True)


if __name__ == "__main__":
    main()
