# MFMA utilisation of the GEMM kernels from the EXECUTED instruction count (rocprofv3 --pmc, counters-only run as the
# pool requires): every v_mfma_f32_16x16x32_bf16 is 16384 flops and 16 matrix-pipe cycles, so
#   executed TFLOP/s = SQ_INSTS_MFMA * 16384 / time          (compare with the algorithmic rate: padding overhead)
#   MfmaUtil         = SQ_INSTS_MFMA * 16 / (time * 2.4 GHz * 1024 SIMDs) = executed rate / 2.5 PFLOP/s
# (the raw SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE ratio is not normalised consistently across XCDs on this stack).
# Usage (GPU box): bash tools/pmc_mfma.sh > gpurun_out/pmc_mfma.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python tools/bench_kernels.py --iters 2 --only nt,tn > gpurun_out/pmc_mfma.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_mfma/*/*_counter_collection.csv')[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = (r['Dispatch_Id'], r['Kernel_Name'], r['Grid_Size'])
    d.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])
    d[k]['t'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
seen = set()
print("kernel | grid | us | executed MFMA TFLOP/s | MfmaUtil vs 2.5 PF | MFMA instrs | wait_inst / wave_cycles | active_inst / wave_cycles")
for (disp, name, grid), c in d.items():
    if 'gemm_' not in name: continue
    short = name[name.find('gemm_'):][:34]
    key = (short, grid)
    if key in seen: continue
    seen.add(key)
    gui = c.get('GRBM_GUI_ACTIVE', 0) or 1
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    n = c.get('SQ_INSTS_MFMA', 0)
    tf = n * 16384 / (c['t'] * 1e-6) / 1e12
    print(f"{short:34s} {grid:>8s} {c['t']:8.0f}  {tf:7.0f} TF/s  util {tf / 2500:5.3f}  "
          f"mfma {n:.4g}  wait {c.get('SQ_WAIT_INST_ANY', 0) / wc:4.2f}  active {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:4.2f}")
PY
