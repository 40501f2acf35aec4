# Timeline of ONE hipGraph replay of the step (rocprofv3 --kernel-trace over bench.py; the last replay is analysed): wall span, time with
# at least one kernel running, idle time between kernels, per-kernel totals.  Usage (GPU box): bash tools/probes/replay_timeline.sh <tag> [bench args]
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/replay_tl_$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/replay_tl_$tag -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/replay_tl_$tag.log 2>&1 < /dev/null
python tools/probes/replay_timeline.py "$tag"
