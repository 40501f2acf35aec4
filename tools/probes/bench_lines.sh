# The four bench lines again on the current tree (bench.py changes only; kernel sources unchanged).  Usage (GPU box): bash tools/probes/bench_lines.sh
out=gpurun_out/evidence_r06
mkdir -p $out
# (the traffic fingerprint is profiles/latest_hbm_traffic.json as committed: re-run tools/evidence.sh after a change under siss_amd/csrc)
python bench.py --steps 20 --warmup 3 > $out/bench_celeb_bs16.json 2> $out/bench_celeb_bs16.err
python bench.py --steps 20 --warmup 3 --loss-fn double_forward_with_neg_del --no-cpu-baseline > $out/bench_celeb_bs16_no_is.json 2> /dev/null
python bench.py --config sd15 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_sd15_bs16.json 2> /dev/null
python bench.py --config sd15 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_sd15_bs4.json 2> /dev/null
for f in bench_celeb_bs16 bench_celeb_bs16_no_is bench_sd15_bs16 bench_sd15_bs4; do python -c "
import json,sys
d=json.loads(open('$out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d.get('device_under_load'), d['step_scalars'])"; done
