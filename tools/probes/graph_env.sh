#!/bin/bash
# Probe: hipGraphLaunch's host cost under the runtime's graph knobs (bench.py: ms per step, host ms inside the launch calls).
# Usage (GPU box): bash tools/probes/graph_env.sh [bench args]
run() {
  env "$@" timeout -k 10 200 python bench.py "${ARGS[@]}" --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>gpurun_out/graph_env.err < /dev/null |
    python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$*', d['ms_per_step'], 'host', d['launch_host_ms_per_step'], d['step_scalars']['pre_clip_norm'])" || { echo "$* FAILED"; tail -3 gpurun_out/graph_env.err; }
}
ARGS=("$@")
run X=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
