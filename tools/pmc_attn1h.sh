# Where the fused single-head attention kernels' time goes (CelebA-HQ shapes): rocprofv3 --pmc passes (counters only, as the pool requires).
# Usage (GPU box): bash tools/pmc_attn1h.sh > gpurun_out/pmc_attn1h.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/probes/attn1h_time.py 2>&1 | grep -v amdgpu
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_attn_a -- python tools/probes/attn1h_time.py > gpurun_out/pmc_attn_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_attn_b -- python tools/probes/attn1h_time.py > gpurun_out/pmc_attn_b.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in ("a", "b"):
    fs = glob.glob(f'gpurun_out/pmc_attn_{tag}/*/*_counter_collection.csv')
    if not fs:
        print("no counters for pass", tag); continue
    d = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if 'attn1h' not in r['Kernel_Name']: continue
        k = r['Kernel_Name'][r['Kernel_Name'].find('attn1h'):][:44] + " grid " + r.get('Grid_Size', '?')
        e = d.setdefault(k, collections.defaultdict(float))
        e[r['Counter_Name']] += float(r['Counter_Value'])
        e['_t_' + r['Counter_Name']] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        e['_n_' + r['Counter_Name']] += 1
    for k, e in d.items():
        names = [c for c in e if not c.startswith('_')]
        n = e['_n_' + names[0]]
        print(k, f"launches {n:.0f} avg {e['_t_' + names[0]] / n:.1f} us")
        for c in names:
            print(f"    {c:26s} {e[c] / n:.4g} per launch")
PY
