"""Kernel time BY PHASE of the bench headline's prune, from a rocprofv3 kernel trace of tools/phase_gpu_bound.py: that tool puts a
spin kernel in front of every phase, so the trace of a prune is six runs of kernels separated by spins.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/phase_gpu_bound.py 60
    python tools/phase_kernels.py DIR out.md [top=22]
The LAST prune of the trace is reported: per phase the span from the first to the last kernel, the busy time, and the kernels by
total time."""
import collections
import csv
import glob
import sys

d, out = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 22
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
spins = [i for i, r in enumerate(rows) if "spin_kernel" in r[2]]
names = ["capture ViT", "walk ViT", "capture T5 encoder", "walk T5 encoder", "capture T5 decoder", "walk T5 decoder"]
assert len(spins) >= 7, f"{len(spins)} spin kernels in the trace"
last = spins[-6:]                                   # (the calibration spin of the tool comes first, then six per prune)
segs = [rows[a + 1:b] for a, b in zip(last, last[1:] + [len(rows)])]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:120]


lines = ["| phase | kernels | first -> last kernel, ms | busy ms | sum of kernel ms |", "|---|---|---|---|---|"]
detail = []
for name, seg in zip(names, segs):
    if not seg:
        continue
    busy, cs, ce = 0, seg[0][0], seg[0][1]
    for s, e, _ in seg[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    tot = sum(e - s for s, e, _ in seg)
    lines.append(f"| {name} | {len(seg)} | {(max(e for _, e, _ in seg) - seg[0][0]) / 1e6:.1f} | {busy / 1e6:.1f} | {tot / 1e6:.1f} |")
    by = collections.defaultdict(lambda: [0, 0])
    for s, e, n in seg:
        by[short(n)][0] += 1
        by[short(n)][1] += e - s
    detail += ["", f"### {name}: {tot / 1e6:.1f} ms of kernels, {len(seg)} launches", "", "| kernel | calls | total ms | avg us |", "|---|---|---|---|"]
    for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:top]:
        detail.append(f"| `{n}` | {c} | {t / 1e6:.2f} | {t / c / 1e3:.1f} |")
open(out, "w").write("\n".join(lines + detail) + "\n")
print("\n".join(lines))
