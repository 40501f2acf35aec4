#!/bin/bash
# Same-box A/B of two builds of libsiss_hip.so (bench.py --lib): alternating runs, ms per step of each.
#   tools/ab_bench.sh tools/ab/libsiss_base.so [bench.py arguments ...]     (run on the GPU box, from the repo root)
base=$1; shift
for i in 1 2; do
  for which in base head; do
    if [ $which = base ]; then lib="--lib $base"; else lib=""; fi
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing $lib "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['ms_per_step'], d['step_ms'])"
  done
done
