mkdir -p gpurun_out
run() { timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" 2>> gpurun_out/bench_s.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'])"; }
echo "default $(run)"
for g in 1200 2500 4000 6000 9000; do echo "side_top_gflop $g: $(run --engine-attr side_top_gflop=$g)"; done
echo "side_top 4000 blocks 160: $(run --engine-attr side_top_gflop=4000 --engine-attr side_blocks=160)"
echo "side_top 4000 blocks 96: $(run --engine-attr side_top_gflop=4000 --engine-attr side_blocks=96)"
echo "default $(run)"
echo "one-stream $(run --engine-attr wgrad_side=0 --engine-attr prep_side=0)"
