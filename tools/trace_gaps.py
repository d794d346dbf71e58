"""Idle time between consecutive kernels of the last bench step in a rocprofv3 --kernel-trace CSV:
   python tools/trace_gaps.py gpurun_out/stats_dir [kernels_per_step]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 261
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
v = [(r['Kernel_Name'].split('(')[0].replace('void ', '')[:44], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
idx = [i for i, x in enumerate(v) if 'vlmc::' in x[0]]
seg = v[idx[-n]:]
gap, dur = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(seg[:-1], seg[1:]):
    gap[b[0]].append(b[1] - a[2])
for x in seg:
    dur[x[0]].append(x[2] - x[1])
print(f"step span {(seg[-1][2] - seg[0][1]) / 1e3:.1f} us, {len(seg)} kernels, busy {sum(sum(d) for d in dur.values()) / 1e3:.1f} us")
for k in dur:
    g = gap[k] or [0]
    print(f"{k:46s} n={len(dur[k]):4d}  avg {sum(dur[k]) / len(dur[k]) / 1e3:7.2f} us   idle before: avg {sum(g) / len(g) / 1e3:5.2f} us, total {sum(g) / 1e3:7.1f} us")
