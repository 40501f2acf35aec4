#!/bin/bash
# Probe: the step scalars of bench.py's SD v1.5 and CelebA-HQ lines with the sparse gradient fill on and off (same seeds, same step count).
for cfg in "--config sd15 --batch 4" "--config sd15 --batch 16" ""; do
  for v in 1 0; do
    python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --engine-attr sparse_fill=$v $cfg 2>/dev/null < /dev/null |
      python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$cfg', 'sparse_fill=$v', d['ms_per_step'], d['step_scalars'])"
  done
done
