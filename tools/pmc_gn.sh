cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_gn -- python tools/bench_kernels.py --iters 2 --only gn > gpurun_out/pmc_gn.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_gn/*/*_counter_collection.csv')[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = (r['Dispatch_Id'], r['Kernel_Name'][:60], r['Grid_Size'], r['VGPR_Count'])
    d.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])
    d[k]['t'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
seen = set()
for (disp, name, grid, vg), c in d.items():
    if 'gn_' not in name: continue
    key = (name, grid)
    if key in seen: continue
    seen.add(key)
    wc = c.get('SQ_WAVE_CYCLES', 1)
    print(name[22:60].ljust(38), grid.rjust(8), 'vgpr', vg, 'us %.0f' % c['t'], 'wait_any %.2f' % (c['SQ_WAIT_ANY'] / wc), 'wait_inst %.2f' % (c['SQ_WAIT_INST_ANY'] / wc),
          'active %.2f' % (c['SQ_ACTIVE_INST_ANY'] / wc), 'valu_insts %.3g' % c['SQ_INSTS_VALU'], 'lds_insts %.3g' % c['SQ_INSTS_LDS'])
PY
