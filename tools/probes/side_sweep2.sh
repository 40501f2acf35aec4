mkdir -p gpurun_out
run() { timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" 2>> gpurun_out/bench_s.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'])"; }
echo "off $(run --engine-attr wgrad_side=0)"
echo "96 $(run)"
echo "96 follow $(run --engine-attr side_follow=1)"
echo "96 follow gmax 24 $(run --engine-attr side_follow=1 --engine-attr group_max=24)"
echo "96 gmax 24 $(run --engine-attr group_max=24)"
echo "128 follow $(run --engine-attr side_follow=1 --engine-attr side_blocks=128)"
echo "64 follow $(run --engine-attr side_follow=1 --engine-attr side_blocks=64)"
echo "96 pair off $(run --engine-attr pair_top=0)"
echo "96 $(run)"
echo "off $(run --engine-attr wgrad_side=0)"
