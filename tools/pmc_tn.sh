cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_tn1 -- python tools/bench_kernels.py --iters 2 --only tn > gpurun_out/pmc_tn1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/pmc_tn2 -- python tools/bench_kernels.py --iters 2 --only tn > gpurun_out/pmc_tn2.log 2>&1
python - <<'PY'
import csv, glob, collections
for d_ in ("gpurun_out/pmc_tn1", "gpurun_out/pmc_tn2"):
    fs = glob.glob(d_ + '/*/*_counter_collection.csv')
    if not fs:
        print("no counters in", d_); continue
    d = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        k = (r['Dispatch_Id'], r['Kernel_Name'][:50], r['Grid_Size'])
        d.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])
        d[k]['t'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    seen = set()
    for (disp, name, grid), c in d.items():
        if 'gemm_tn' not in name: continue
        key = (name, grid)
        if key in seen: continue
        seen.add(key)
        print(name[:40].ljust(40), grid.rjust(8), 'us %.0f' % c['t'], {k: ('%.3g' % v) for k, v in c.items() if k != 't'})
PY
