# Per-kernel times of the fused attention kernels on the SD self-attention shape (rocprofv3 kernel trace of tools/probes/flash_time.py).
# Usage (GPU box): bash tools/probes/prof_flash.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-x}
PRE=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_flash_$tag -- python tools/probes/flash_time.py > gpurun_out/prof_flash_$tag.log 2>&1 < /dev/null
tail -1 gpurun_out/prof_flash_$tag.log
python - "$tag" <<'PY'
import csv, glob, sys
fs = glob.glob(f'gpurun_out/prof_flash_{sys.argv[1]}/*/*kernel_stats.csv')
for f in fs[:1]:
    for r in list(csv.DictReader(open(f)))[:8]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
