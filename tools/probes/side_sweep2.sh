mkdir -p gpurun_out
run() { timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" 2>> gpurun_out/bench_s.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'])"; }
echo "default $(run)"
echo "side_tail=0 $(run --engine-attr side_tail=0)"
echo "subpixel_queue=0 $(run --engine-attr subpixel_queue=0)"
echo "both off $(run --engine-attr subpixel_queue=0 --engine-attr side_tail=0)"
echo "default $(run)"
echo "one-stream $(run --engine-attr wgrad_side=0 --engine-attr prep_side=0 --engine-attr subpixel_queue=0)"
