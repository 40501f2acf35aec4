#!/bin/bash
# ONE end-of-round evidence refresh (run on the GPU box from the repo root): bash tools/evidence.sh r06
# Everything lands in gpurun_out/evidence_<tag>/ ; copy into profiles/ afterwards (tools/evidence_copy.py <tag>).
tag=${1:-rXX}
out=gpurun_out/evidence_$tag
mkdir -p $out
echo "[evidence] pmc traffic first: the bench lines fingerprint profiles/latest_hbm_traffic.json against the kernel sources" | tee $out/progress.txt
bash tools/pmc_traffic.sh $tag > $out/pmc_traffic.log 2>&1
cp gpurun_out/${tag}_hbm_traffic.json profiles/latest_hbm_traffic.json
cp gpurun_out/${tag}_hbm_traffic.json gpurun_out/${tag}_kernel_stats.csv $out/
echo "[evidence] pmc traffic done" >> $out/progress.txt
python bench.py --steps 20 --warmup 3 > $out/bench_celeb_bs16.json 2> $out/bench_celeb_bs16.err
echo "[evidence] celeb done" >> $out/progress.txt
python bench.py --steps 20 --warmup 3 --loss-fn double_forward_with_neg_del --no-cpu-baseline > $out/bench_celeb_bs16_no_is.json 2> /dev/null
python bench.py --config sd15 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_sd15_bs16.json 2> /dev/null
python bench.py --config sd15 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_sd15_bs4.json 2> /dev/null
echo "[evidence] bench lines done" >> $out/progress.txt
python tools/step_breakdown.py --top 200 > $out/step_breakdown_celeb_bs16.txt 2>&1
python tools/step_breakdown.py --config sd15 --batch 16 --top 200 > $out/step_breakdown_sd15_bs16.txt 2>&1
python tools/step_breakdown.py --config sd15 --batch 4 --top 200 > $out/step_breakdown_sd15_bs4.txt 2>&1
echo "[evidence] breakdowns done" >> $out/progress.txt
bash tools/pmc_flash.sh > $out/pmc_flash32.txt 2>&1 < /dev/null
bash tools/probes/prof_flash.sh $tag > $out/prof_flash32.txt 2>&1 < /dev/null
CLOCK=1 PRE=1 python tools/probes/flash_time.py > $out/flash32_power_clock.txt 2>&1
echo "[evidence] all done" >> $out/progress.txt
ls -la $out
