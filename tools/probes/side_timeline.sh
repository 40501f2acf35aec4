# Timeline of the side-stream weight gradients against the main stream (rocprofv3 --kernel-trace, one eager step analysed).
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
# last step: from the last mixture kernel on
idx = [i for i, r in enumerate(rows) if 'mixture' in r['Kernel_Name']]
start = idx[-1]
rows = rows[start:]
t0 = int(rows[0]['Start_Timestamp'])
def nm(r):
    k = r['Kernel_Name']
    k = k.replace('(anonymous namespace)::', '')
    return k[:46]
side = [r for r in rows if 'grouped_capped' in r['Kernel_Name']]
print("step kernels", len(rows), "side launches", len(side))
for r in side:
    print(f"SIDE {nm(r):46s} start {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  grid {r.get('Grid_Size')}")
if side:
    s0, s1 = min(int(r['Start_Timestamp']) for r in side), max(int(r['End_Timestamp']) for r in side)
    print(f"side window {(s0 - t0) / 1e3:.1f} .. {(s1 - t0) / 1e3:.1f} us")
    # main-stream kernels overlapping the window
    acc = {}
    for r in rows:
        if 'grouped_capped' in r['Kernel_Name']: continue
        a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        ov = max(0, min(b, s1) - max(a, s0))
        if ov > 0:
            k = nm(r)
            e = acc.setdefault(k, [0, 0.0, 0.0])
            e[0] += 1; e[1] += ov / 1e3; e[2] += (b - a) / 1e3
    print("main-stream kernels inside the side window: launches, overlapped us, own us")
    for k, e in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"   {k:46s} {e[0]:4d} {e[1]:9.1f} {e[2]:9.1f}")
print(f"step span {(int(rows[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
PY
