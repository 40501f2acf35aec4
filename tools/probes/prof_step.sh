# rocprofv3 kernel stats of one eager step (tools/step_breakdown.py).  Usage (GPU box): bash tools/probes/prof_step.sh <tag> <config> <batch>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-x}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_step_$tag -- python tools/step_breakdown.py --config ${2:-sd15} --batch ${3:-4} --top 5 > gpurun_out/prof_step_$tag.log 2>&1 < /dev/null
python - "$tag" <<'PY'
import csv, glob, sys
fs = glob.glob(f'gpurun_out/prof_step_{sys.argv[1]}/*/*kernel_stats.csv')
for f in fs[:1]:
    for r in list(csv.DictReader(open(f)))[:45]:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
