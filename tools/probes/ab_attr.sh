# A/B of one engine attribute on the CelebA-HQ bench, alternating runs on one box:  bash tools/probes/ab_attr.sh phase_launch=0 [rounds]
mkdir -p gpurun_out
run() { timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" 2>> gpurun_out/bench_ab.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'])"; }
for i in $(seq 1 ${2:-2}); do
echo "default $(run)"
echo "$1 $(run --engine-attr $1)"
done
