# Sweep of the weight-gradients-beside-GroupNorm schedule (UNetEngine.side_blocks x side_rate [permille: MFLOP / us / CU]), ms per step.
# Usage (GPU box): bash tools/probes/side_sweep.sh "96 128 160" "3 4 5"
mkdir -p gpurun_out
for sb in ${1:-96 128 160}; do for rt in ${2:-3 4 5}; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --engine-attr side_blocks=$sb --engine-attr side_rate=$rt > gpurun_out/bench_s_${sb}_${rt}.json 2>> gpurun_out/bench_s.err || exit 1
  python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1], d['ms_per_step'])" gpurun_out/bench_s_${sb}_${rt}.json
done; done
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --engine-attr wgrad_side=0 > gpurun_out/bench_s_off.json 2>> gpurun_out/bench_s.err
python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1], d['ms_per_step'])" gpurun_out/bench_s_off.json
