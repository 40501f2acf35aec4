# Sweep of the side-stream weight-gradient schedule (UNetEngine.side_blocks x side_budget), CelebA-HQ B = 16, ms per step.
# Usage (GPU box): bash tools/probes/side_sweep.sh "128 160 192" "1500 2500 4000"
mkdir -p gpurun_out
for sb in ${1:-128 160 192 224}; do for bg in ${2:-1500 2000 2500 3000 4000}; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --engine-attr side_blocks=$sb --engine-attr side_budget=$bg > gpurun_out/bench_s_${sb}_${bg}.json 2>> gpurun_out/bench_s.err || exit 1
  python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1], d['ms_per_step'])" gpurun_out/bench_s_${sb}_${bg}.json
done; done
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --engine-attr wgrad_side=0 > gpurun_out/bench_s_off.json 2>> gpurun_out/bench_s.err
python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1], d['ms_per_step'])" gpurun_out/bench_s_off.json
