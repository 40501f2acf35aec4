# Sweep of the background-weight-gradient schedule (UNetEngine.side_blocks x side_flush_gflop), CelebA-HQ B = 16, ms per step.
# Usage (GPU box): bash tools/probes/side_sweep.sh "64 96 128" "400 1000"
mkdir -p gpurun_out
for sb in ${1:-64 80 96 112 128}; do for fg in ${2:-400}; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --engine-attr side_blocks=$sb --engine-attr side_flush_gflop=$fg > gpurun_out/bench_s_${sb}_${fg}.json 2>> gpurun_out/bench_s.err || exit 1
  python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1], d['ms_per_step'])" gpurun_out/bench_s_${sb}_${fg}.json
done; done
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --engine-attr wgrad_side=0 > gpurun_out/bench_s_off.json 2>> gpurun_out/bench_s.err
python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1], d['ms_per_step'])" gpurun_out/bench_s_off.json
