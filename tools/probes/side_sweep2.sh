mkdir -p gpurun_out
run() { timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" 2>> gpurun_out/bench_s.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'])"; }
echo "default $(run)"
for k in 2 3 4; do echo "side_tail=$k $(run --engine-attr side_tail=$k)"; done
echo "group_max 56 $(run --engine-attr group_max=56)"
echo "group_max 32 $(run --engine-attr group_max=32)"
echo "default $(run)"
echo "one-stream $(run --engine-attr wgrad_side=0 --engine-attr prep_side=0 --engine-attr subpixel_queue=0)"
