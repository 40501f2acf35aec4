# A/B of an environment setting on the CelebA-HQ bench, alternating on one box:  bash tools/probes/ab_env.sh SISS_C3P_MIN_TILES=300 SISS_C3P_MIN_TILES=600
mkdir -p gpurun_out
run() { timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>> gpurun_out/bench_ab.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'])"; }
for rep in 1 2; do
echo "default $(run)"
for kv in "$@"; do echo "$kv $(export $kv; run)"; done
done
