"""Analysis half of tools/probes/replay_timeline.sh: python tools/probes/replay_timeline.py <tag> (reads gpurun_out/replay_tl_<tag>/)."""
import csv, glob, sys
f = glob.glob(f'gpurun_out/replay_tl_{sys.argv[1]}/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'mixture' in r['Kernel_Name']]
idx = [i for n, i in enumerate(idx) if n == 0 or int(rows[i]['Start_Timestamp']) - int(rows[idx[n - 1]]['Start_Timestamp']) > 1e6]   # (the mixture is two kernels)
rows = rows[idx[-2]:idx[-1]]                      # one whole replay: mixture kernel to the next one
t0, t1 = int(rows[0]['Start_Timestamp']), int(rows[-1]['End_Timestamp'])
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels {len(rows)}  span {(t1 - t0) / 1e3:.0f} us  busy(union) {busy / 1e3:.0f} us  idle {(t1 - t0 - busy) / 1e3:.0f} us in {len(gaps)} gaps "
      f"(median {sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3:.1f} us)")
nm = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0][:60] + ('<' + r['Kernel_Name'].split('<')[1].split('>')[0][:24] + '>' if '<' in r['Kernel_Name'] and 'at::native' not in r['Kernel_Name'] else '')
acc = {}
for r in rows:
    e = acc.setdefault(nm(r), [0, 0.0]); e[0] += 1; e[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, e in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {k:60s} {e[0]:4d} {e[1]:9.1f} us  avg {e[1] / e[0]:7.1f}")
# idle by the kernel that FOLLOWS the gap
byn = {}
ends = {}
for r in rows: ends.setdefault(int(r['Start_Timestamp']), nm(r))
starts = sorted(ends)
import bisect
for g, at in gaps:
    i = bisect.bisect_left(starts, at)
    k = ends[starts[i]] if i < len(starts) else '?'
    e = byn.setdefault(k, [0, 0.0]); e[0] += 1; e[1] += g / 1e3
print("idle time by the kernel that ends the gap:")
for k, e in sorted(byn.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"  {k:60s} {e[0]:4d} gaps {e[1]:8.1f} us")
if len(sys.argv) > 2:                              # python tools/probes/replay_timeline.py <tag> <first> <count>: the launches themselves
    a, n = int(sys.argv[2]), int(sys.argv[3])
    prev_end = None
    for r in rows[a:a + n]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print(f"  q{r['Queue_Id']:>2s} start {(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} gap {((s - prev_end) / 1e3) if prev_end else 0:6.1f}  {nm(r)} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']} lds {r['LDS_Block_Size']}")
        prev_end = e
