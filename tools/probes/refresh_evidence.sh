set -e
tag=${1:-r05a}
bash tools/pmc_traffic.sh $tag
cp gpurun_out/${tag}_hbm_traffic.json profiles/latest_hbm_traffic.json
echo "== celeb bench"
python bench.py > gpurun_out/${tag}_bench_celeb_bs16.json 2> gpurun_out/${tag}_bench_celeb_bs16.err
tail -c 600 gpurun_out/${tag}_bench_celeb_bs16.json
echo "== celeb breakdown"
python tools/step_breakdown.py --top 130 > gpurun_out/${tag}_step_breakdown_celeb_bs16.txt 2>&1
echo "== no-is"
python bench.py --loss-fn double_forward_with_neg_del --no-cpu-baseline > gpurun_out/${tag}_bench_celeb_bs16_no_is.json 2>/dev/null
for bs in 4 16; do
  echo "== sd $bs"
  python bench.py --config sd15 --batch $bs --steps 10 --no-cpu-baseline > gpurun_out/${tag}_bench_sd15_bs$bs.json 2>/dev/null
  python tools/step_breakdown.py --config sd15 --batch $bs --top 60 > gpurun_out/${tag}_step_breakdown_sd15_bs$bs.txt 2>&1
done
echo done
