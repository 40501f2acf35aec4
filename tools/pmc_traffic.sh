#!/bin/bash
# HBM traffic of the step's kernels from the L2 memory-side counters (two separate --pmc passes, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Run on the GPU box:
#   bash tools/pmc_traffic.sh r01c
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 2 --warmup 1 --graph 0 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_$tag -- $CMD > gpurun_out/pmc_fetch_$tag.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_$tag -- $CMD > gpurun_out/pmc_write_$tag.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- $CMD > gpurun_out/prof_$tag.log 2>&1
python tools/parse_pmc.py gpurun_out/pmc_fetch_$tag gpurun_out/pmc_write_$tag gpurun_out/prof_$tag $tag
