"""Time the fused attention kernels on the SD v1.5 self-attention shape at 64 x 64 latents (B = 16: 128 forward (batch, head) entries,
256 cotangent entries, 4096 queries and keys, head dim 40) in the projections' own layout.  A/B builds through SISS_LIB_PATH."""
import os, sys, torch
PRE = int(os.environ.get('PRE', '0'))      # 1: q pre-scaled by scale * log2(e) (q_prescaled form)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from siss_amd import lib
from tools.bench_kernels import timeit
dev = torch.device("cuda:0"); lib.load()
B, H, S, D = int(os.environ.get("B", "16")), 8, int(os.environ.get("S", "4096")), int(os.environ.get("D", "40"))
C = H * D
g = torch.Generator(device=dev).manual_seed(0)
r = lambda n: torch.randn(n, C, generator=g, device=dev).to(torch.bfloat16)
q, k, v, do = r(B * S), r(B * S), r(B * S), r(2 * B * S)
o = torch.empty_like(q); lse = torch.zeros(B * H, S, device=dev)
dq, dk, dv = (torch.empty(2 * B * S, C, dtype=torch.bfloat16, device=dev) for _ in range(3))
delta = torch.zeros(2 * B * H * S, device=dev)
sc = D ** -0.5
if PRE:
    q = (q.float() * (sc * 1.4426950408889634)).to(torch.bfloat16)
tf = timeit(lambda: lib.call("siss_flash_attn_fwd_merged", q, C, k, C, v, C, o, C, lse, B, H, S, S, D, sc, PRE), 5)
tb = timeit(lambda: lib.call("siss_flash_attn_bwd_merged", q, C, k, C, v, C, o, C, do, C, lse, delta, dq, C, dk, C, dv, C, 2 * B, B, H, S, S, D, sc, PRE), 5)
fl = 2.0 * B * H * S * S * D
print(f"B {B} S {S} D {D}: fwd {tf * 1e3:8.1f} us ({2 * fl / tf / 1e9:6.0f} TF/s)   bwd {tb * 1e3:8.1f} us ({2 * 5 * fl / tb / 1e9:6.0f} TF/s)")
if os.environ.get("CLOCK"):
    # shader clock / power while the backward runs back to back (~4 s queued), three rocm-smi samples
    import subprocess, time
    for _ in range(int(4.0 / (tb * 1e-3))):
        lib.call("siss_flash_attn_bwd_merged", q, C, k, C, v, C, o, C, do, C, lse, delta, dq, C, dk, C, dv, C, 2 * B, B, H, S, S, D, sc, PRE)
    for _ in range(3):
        time.sleep(0.8)
        out = subprocess.run([sys.executable, "/opt/rocm/libexec/rocm_smi/rocm_smi.py", "--showpower", "--showclocks"], capture_output=True, text=True).stdout   # (no env-shebang hop)
        print(" | ".join(l.strip() for l in out.splitlines() if "sclk" in l or "Power (W)" in l or "Average Graphics" in l))
    torch.cuda.synchronize()
