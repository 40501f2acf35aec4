"""Alternating same-box runs of bench.py over the VALUES of one switch: python tools/probes/ab_attr_vals.py <flag> <name> v0,v1,... [bench args...]"""
import json, subprocess, sys
flag, name, vals, rest = sys.argv[1], sys.argv[2], sys.argv[3].split(","), sys.argv[4:]
for rep in range(2):
    for v in vals:
        r = subprocess.run([sys.executable, "bench.py", "--steps", "15", "--warmup", "3", "--no-cpu-baseline", "--no-kernel-timing", flag, f"{name}={v}", *rest],
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(name, v, json.loads(line[-1])["ms_per_step"] if line else r.stderr[-300:], flush=True)
