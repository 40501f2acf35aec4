# Executed instruction mix and wave-cycle shares of the step's kernels (rocprofv3 --pmc, counters-only pass) over one eager step.
# Usage (GPU box): bash tools/pmc_mix.sh [extra bench.py args] > gpurun_out/pmc_mix.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_mix -- python bench.py --steps 1 --warmup 1 --graph 0 --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/pmc_mix.log 2>&1
python - <<'PY'
import csv, glob, collections
f = sorted(glob.glob('gpurun_out/pmc_mix/*/*_counter_collection.csv'))[-1]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    k = k[k.find('::') + 2:][:44] if '::' in k else k[:44]
    e = d.setdefault(k, collections.defaultdict(float))
    e[r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
        e['t'] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print(f"{'kernel':46s} {'ms':>7s} {'valu/mfma':>9s} {'salu/mfma':>9s} {'lds/mfma':>8s} {'wait/wave':>9s} {'active/wave':>11s} {'mfma TF/s exec':>14s}")
for k, e in sorted(d.items(), key=lambda kv: -kv[1]['t'])[:14]:
    m = e['SQ_INSTS_MFMA'] or 1
    wc = e['SQ_WAVE_CYCLES'] or 1
    print(f"{k:46s} {e['t'] / 1e3:7.2f} {e['SQ_INSTS_VALU'] / m:9.2f} {e['SQ_INSTS_SALU'] / m:9.2f} {e['SQ_INSTS_LDS'] / m:8.2f} "
          f"{e['SQ_WAIT_INST_ANY'] / wc:9.2f} {e['SQ_ACTIVE_INST_ANY'] / wc:11.2f} {e['SQ_INSTS_MFMA'] * 16384 / (e['t'] * 1e-6) / 1e12:14.0f}")
PY
