# Timeline of the side-stream weight gradients against the main stream (rocprofv3 --kernel-trace, last eager step analysed).
# Usage (GPU box): bash tools/probes/side_timeline.sh [side_blocks] > gpurun_out/side_timeline.txt
SB=${1:-96}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/side_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/side_tl -- python bench.py --steps 3 --warmup 2 --graph 0 --no-cpu-baseline --no-kernel-timing --engine-attr side_blocks=$SB > gpurun_out/side_tl.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/side_tl/*/*_kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'mixture' in r['Kernel_Name']]
rows = rows[idx[-1]:]
t0 = int(rows[0]['Start_Timestamp'])
nm = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '')[:44]
us = lambda v: (int(v) - t0) / 1e3
qs = sorted({r.get('Queue_Id', '?') for r in rows})
print("queues", qs, "kernels", len(rows), f"step span {us(rows[-1]['End_Timestamp']):.0f} us")
side = [r for r in rows if 'gemm_tn' in r['Kernel_Name']]
main = [r for r in rows if 'gemm_tn' not in r['Kernel_Name']]
busy = lambda rs: sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e3
print(f"wgrad kernels {len(side)} busy {busy(side):.0f} us; first start {us(side[0]['Start_Timestamp']):.0f}, last end {us(max(side, key=lambda r: int(r['End_Timestamp']))['End_Timestamp']):.0f}")
print(f"other kernels {len(main)} busy {busy(main):.0f} us")
# per kernel-name totals in the backward part (after the first wgrad start)
tb = int(side[0]['Start_Timestamp'])
acc = {}
for r in main:
    if int(r['Start_Timestamp']) < tb: continue
    e = acc.setdefault(nm(r), [0, 0.0]); e[0] += 1; e[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, e in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"   main  {k:44s} {e[0]:4d} {e[1]:9.1f} us")
for r in side[:12] + side[-6:]:
    print(f"   wgrad {nm(r):44s} start {us(r['Start_Timestamp']):9.0f} dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.0f} us wg {r.get('Workgroup_Size_X','?')} grid {r.get('Grid_Size_X', r.get('Grid_Size','?'))}")
PY
