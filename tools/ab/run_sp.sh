python -m pytest tests/test_hip_unet.py -q -m gpu -x -k "subpixel or forward_matches or dual_backward or siss_step_matches" > gpurun_out/t8.log 2>&1; echo rc=$?; tail -3 gpurun_out/t8.log
for cfg in "subpixel_up=0" "subpixel_up=1" "subpixel_min_px=1024" "subpixel_min_px=4096" "subpixel_up=0" "subpixel_min_px=4096"; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --engine-attr $cfg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['step_ms']['p50'])"
done
python tools/step_breakdown.py --top 200 > gpurun_out/breakdown_sp.txt 2>&1
