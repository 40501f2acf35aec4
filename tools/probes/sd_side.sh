mkdir -p gpurun_out
run() { timeout -k 10 400 python bench.py --config sd15 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing "$@" 2>> gpurun_out/bench_sd.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'])"; }
echo "sd15 B=4 default $(run --batch 4)"
echo "sd15 B=4 one-stream $(run --batch 4 --engine-attr wgrad_side=0 --engine-attr prep_side=0)"
echo "sd15 B=16 default $(run --batch 16)"
echo "sd15 B=16 one-stream $(run --batch 16 --engine-attr wgrad_side=0 --engine-attr prep_side=0)"
echo "sd15 B=16 side 96 $(run --batch 16 --engine-attr side_blocks=96)"
echo "sd15 B=16 side 160 $(run --batch 16 --engine-attr side_blocks=160)"
echo "sd15 B=16 px1024 $(run --batch 16 --engine-attr side_max_px=1024)"
echo "sd15 B=4 px1024 $(run --batch 4 --engine-attr side_max_px=1024)"
