# Where the fused attention kernels' time goes (SD v1.5 self-attention shape, B = 16, 4096 keys, D = 40): executed instruction counts
# and wave-cycle shares per kernel from rocprofv3 --pmc (counters-only passes, as the pool requires).
# Usage (GPU box): bash tools/pmc_flash.sh > gpurun_out/pmc_flash.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PRE=1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_flash_a -- python tools/probes/flash_time.py > gpurun_out/pmc_flash_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_flash_b -- python tools/probes/flash_time.py > gpurun_out/pmc_flash_b.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in ("a", "b"):
    fs = glob.glob(f'gpurun_out/pmc_flash_{tag}/*/*_counter_collection.csv')
    if not fs:
        print("no counters for pass", tag); continue
    d = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        kn = r['Kernel_Name']
        if 'flash' not in kn and 'fa32' not in kn: continue
        k = kn[min(i for i in (kn.find('flash'), kn.find('fa32')) if i >= 0):][:40]
        e = d.setdefault(k, collections.defaultdict(float))
        e[r['Counter_Name']] += float(r['Counter_Value'])
        e['_t_' + r['Counter_Name']] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        e['_n_' + r['Counter_Name']] += 1
    for k, e in d.items():
        names = [c for c in e if not c.startswith('_')]
        n = e['_n_' + names[0]]
        print(k, f"launches {n:.0f} avg {e['_t_' + names[0]] / n:.0f} us")
        for c in names:
            print(f"    {c:26s} {e[c] / n:.4g} per launch")
PY
