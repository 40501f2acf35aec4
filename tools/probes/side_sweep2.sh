mkdir -p gpurun_out
run() { timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" 2>> gpurun_out/bench_s.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'])"; }
echo "off $(run --engine-attr wgrad_side=0)"
for sb in 112 128 144 160 176 192 224; do echo "$sb follow $(run --engine-attr side_follow=1 --engine-attr side_blocks=$sb)"; done
echo "128 follow gmax 64 $(run --engine-attr side_follow=1 --engine-attr side_blocks=128 --engine-attr group_max=64)"
echo "128 follow px1024 $(run --engine-attr side_follow=1 --engine-attr side_blocks=128 --engine-attr side_max_px=1024)"
echo "off $(run --engine-attr wgrad_side=0)"
