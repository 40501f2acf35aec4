# A/B of several engine-attribute settings against the default, alternating on one box:  bash tools/probes/ab_attrs.sh "a=1 b=2" "a=3"
mkdir -p gpurun_out
run() { timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" 2>> gpurun_out/bench_ab.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'])"; }
for rep in 1 2; do
echo "default $(run)"
for setting in "$@"; do
  args=""; for kv in $setting; do args="$args --engine-attr $kv"; done
  echo "$setting $(run $args)"
done
done
