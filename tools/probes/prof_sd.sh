# rocprofv3 --kernel-trace --stats of three eager SD v1.5 steps (bench.py --graph 0) at B = 16 and B = 4.  Usage (GPU box): bash tools/probes/prof_sd.sh <tag>
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 16 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sd${b}_$tag -- python bench.py --config sd15 --batch $b --steps 2 --warmup 1 --graph 0 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_sd${b}_$tag.log 2>&1 < /dev/null
  f=$(ls gpurun_out/prof_sd${b}_$tag/*/*kernel_stats.csv | head -1)
  cp $f gpurun_out/${tag}_kernel_stats_sd15_bs$b.csv
  head -12 gpurun_out/${tag}_kernel_stats_sd15_bs$b.csv | cut -c1-160
done
