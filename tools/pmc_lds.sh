# LDS bank conflicts of the step's kernels (rocprofv3 --pmc, counters-only pass): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel
# over one eager CelebA-HQ step.  Usage (GPU box): bash tools/pmc_lds.sh [extra bench.py args] > gpurun_out/pmc_lds.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_lds -- python bench.py --steps 1 --warmup 1 --graph 0 --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/pmc_lds.log 2>&1
python - <<'PY'
import csv, glob, collections
f = sorted(glob.glob('gpurun_out/pmc_lds/*/*_counter_collection.csv'))[-1]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    k = k[k.find('::') + 2:][:48] if '::' in k else k[:48]
    e = d.setdefault(k, collections.defaultdict(float))
    e[r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
        e['t'] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print(f"{'kernel':50s} {'ms':>8s} {'conflict/active':>16s} {'wait_lds/wave':>14s}")
for k, e in sorted(d.items(), key=lambda kv: -kv[1]['t'])[:24]:
    act = e['SQ_LDS_IDX_ACTIVE'] or 1
    print(f"{k:50s} {e['t'] / 1e3:8.2f} {e['SQ_LDS_BANK_CONFLICT'] / act:16.3f} {e['SQ_WAIT_INST_LDS'] / (e['SQ_WAVE_CYCLES'] or 1):14.3f}")
PY
