# Per-kernel times of the attention kernels on SD's cross-attention shapes (rocprofv3 kernel trace of tools/probes/flash_cross_time.py).
# Usage (GPU box): bash tools/probes/prof_cross.sh <tag> [B]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-x}
PRE=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cross_$tag -- python tools/probes/flash_cross_time.py ${2:-16} > gpurun_out/prof_cross_$tag.log 2>&1 < /dev/null
grep "^B " gpurun_out/prof_cross_$tag.log
python - "$tag" <<'PY'
import csv, glob, sys
fs = glob.glob(f'gpurun_out/prof_cross_{sys.argv[1]}/*/*kernel_stats.csv')
for f in fs[:1]:
    for r in list(csv.DictReader(open(f)))[:14]:
        print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
