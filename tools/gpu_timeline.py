"""GPU busy fraction of the timed prunes from a rocprofv3 kernel trace.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --kernel-pass 0
    python tools/gpu_timeline.py DIR out.md
Steps are delimited by the weight restore at the start of every bench step (the first of its multi-tensor copy kernels after
a gap in them: a run of >= 300 consecutive copy kernels); the union of the kernels' [start, end] intervals inside a step is its busy time."""
import csv
import glob
import sys

d, out = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# step starts: the weight restore = a run of >= 300 consecutive copy kernels (one per parameter tensor)
starts, run0, run = [], None, 0
for i, (s, e, n) in enumerate(rows):
    if "copyBuffer" in n or "multi_tensor" in n or "direct_copy" in n:
        if run == 0:
            run0 = i
        run += 1
    else:
        if run >= 300:
            starts.append(run0)
        run = 0
hist = []
lines = ["| step | wall ms (first kernel of the step to first kernel of the next) | GPU busy ms | busy | gaps > 0.5 ms (ms) |", "|---|---|---|---|---|"]
for k in range(len(starts) - 1):
    seg = rows[starts[k]:starts[k + 1]]
    t0, t1 = seg[0][0], rows[starts[k + 1]][0]
    busy, cur_s, cur_e, gaps = 0, seg[0][0], seg[0][1], []
    for s, e, _ in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            if s - cur_e > 500_000:
                gaps.append((s - cur_e) / 1e6)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    lines.append(f"| {k} | {(t1 - t0) / 1e6:.1f} | {busy / 1e6:.1f} | {busy / (t1 - t0):.3f} | {', '.join(f'{g:.1f}' for g in gaps) or '-'} |")
    # idle time by the length of the gap (stream-ordered kernels: gap = next start - latest end so far)
    edges = [2, 5, 10, 50, 500, 10 ** 9]
    cnt, tot, end = [0] * len(edges), [0.0] * len(edges), seg[0][1]
    for s, e, _ in seg[1:]:
        g = (s - end) / 1e3
        if g > 0:
            b = next(i for i, x in enumerate(edges) if g < x)
            cnt[b] += 1
            tot[b] += g
        end = max(end, e)
    hist.append((k, len(seg), cnt, tot))
lines += ["", "| step | kernels | idle in gaps < 2 us: n, ms | 2-5 us | 5-10 us | 10-50 us | 50-500 us | > 500 us |", "|---|---|---|---|---|---|---|---|"]
for k, n, cnt, tot in hist:
    lines.append(f"| {k} | {n} | " + " | ".join(f"{c}, {t / 1e3:.1f}" for c, t in zip(cnt, tot)) + " |")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
