"""GPU busy fraction, kernel count and gap histogram over the LAST `frac` of a rocprofv3 kernel trace (warm iterations):
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 <probe>;  python tools/busy_tail.py DIR [frac=0.4]"""
import csv
import glob
import sys

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
t_end = max(e for _, e, _ in rows)
t_begin = rows[0][0]
cut = t_end - (t_end - t_begin) * frac
seg = [r for r in rows if r[0] >= cut]
busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
gaps = []
for s, e, _ in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = seg[-1][1] - seg[0][0]
dur = sorted(e - s for s, e, _ in seg)
print(f"last {frac:.0%} of the trace: wall {wall / 1e6:.1f} ms, {len(seg)} kernels, busy {busy / 1e6:.1f} ms = {busy / wall:.3f}")
print(f"kernel duration us: median {dur[len(dur) // 2] / 1e3:.1f}, mean {sum(dur) / len(dur) / 1e3:.1f}, p90 {dur[int(len(dur) * 0.9)] / 1e3:.1f}")
edges = [2, 5, 10, 20, 50, 500, 10 ** 9]
lo = 0
for hi in edges:
    sel = [g for g in gaps if lo * 1000 <= g < hi * 1000]
    print(f"gaps {lo:>4}-{hi if hi < 10 ** 9 else 'inf':>4} us: {len(sel):6d}  {sum(sel) / 1e6:8.1f} ms")
    lo = hi
