#!/bin/bash
# Probe: the step as a LINEAR hipGraph (no side streams: the runtime's fast launch path) against the two-stream schedule.
# Usage (GPU box): bash tools/probes/graph_linear.sh [bench args]
run() {
  timeout -k 10 200 python bench.py "${ARGS[@]}" "$@" --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>gpurun_out/graph_lin.err < /dev/null |
    python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$*', d['ms_per_step'], 'host', d['launch_host_ms_per_step'])" || { echo "$* FAILED"; tail -3 gpurun_out/graph_lin.err; }
}
ARGS=("$@")
for rep in 1 2; do
run
run --engine-attr wgrad_side=0 --engine-attr prep_side=0
run --engine-attr wgrad_side=0
done
