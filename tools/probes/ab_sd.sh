# SD v1.5 step with / without one engine attribute, alternating on one box:  bash tools/probes/ab_sd.sh wgrad_side=0
mkdir -p gpurun_out
run() { timeout -k 10 400 python bench.py --config sd15 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" 2>> gpurun_out/bench_ab.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'])"; }
for bs in 4 16; do
for i in 1 2; do
echo "B=$bs default $(run --batch $bs)"
echo "B=$bs $1 $(run --batch $bs --engine-attr $1)"
done
done
