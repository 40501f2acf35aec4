# Sweep of the side-stream weight-gradient schedule (UNetEngine.side_blocks x side_max_px), CelebA-HQ B = 16, ms per step; every
# sweep ends (and starts) with the one-stream schedule on the same box.
# Usage (GPU box): bash tools/probes/side_sweep.sh "80 96 112" "256 1024"
mkdir -p gpurun_out
run() { timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" 2>> gpurun_out/bench_s.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'])"; }
echo "off $(run --engine-attr wgrad_side=0)"
for px in ${2:-256}; do for sb in ${1:-80 88 96 104 112}; do
  echo "px $px blocks $sb: $(run --engine-attr side_blocks=$sb --engine-attr side_max_px=$px)"
done; done
echo "off $(run --engine-attr wgrad_side=0)"
