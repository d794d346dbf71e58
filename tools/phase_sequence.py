"""The kernel SEQUENCE of a stretch of one phase (same trace as tools/phase_kernels.py): python tools/phase_sequence.py DIR phase_index first count"""
import csv
import glob
import sys

d, ph, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
spins = [i for i, r in enumerate(rows) if "spin_kernel" in r[2]][-6:]
segs = [rows[a + 1:b] for a, b in zip(spins, spins[1:] + [len(rows)])]
seg = segs[ph]
prev = seg[first - 1][1] if first > 0 else seg[0][0]
for s, e, n in seg[first:first + count]:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    print(f"{(s - prev) / 1e3:7.1f} gap {(e - s) / 1e3:7.1f} us  {n[:150]}")
    prev = e
