"""Alternating same-box A/B of one bench.py switch: python tools/probes/ab_attr.py <flag> <name> [bench args...]
   e.g. python tools/probes/ab_attr.py --stepper-attr wgrad_overwrite --config sd15 --batch 4"""
import json, subprocess, sys
flag, name, rest = sys.argv[1], sys.argv[2], sys.argv[3:]
for rep in range(2):
    for v in (1, 0):
        r = subprocess.run([sys.executable, "bench.py", "--steps", "15", "--warmup", "3", "--no-cpu-baseline", "--no-kernel-timing", flag, f"{name}={v}", *rest],
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(name, v, json.loads(line[-1])["ms_per_step"] if line else r.stderr[-300:], flush=True)
