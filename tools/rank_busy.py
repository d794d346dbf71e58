"""GPU busy time per tower from the kernel trace of tools/rank_timeline.py (towers start at the marker cumsum kernel):
    python tools/rank_busy.py DIR"""
import csv
import glob
import sys
from collections import Counter

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
starts = [i for i, r in enumerate(rows) if "cumsum" in r[2].lower() or "scan" in r[2].lower()]
print(f"{len(rows)} kernels, {len(starts)} tower markers")
names = ["vit", "t5enc", "t5dec"]
last = starts[-6:]                      # the last prune: (first, last) marker of each tower
for k in range(3):
    i, j = last[2 * k], last[2 * k + 1]
    seg = rows[i + 1:j]
    busy, cur_s, cur_e, gaps = 0, seg[0][0], seg[0][1], []
    for s, e, _ in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    wall = seg[-1][1] - seg[0][0]
    print(f"{names[k]}: wall {wall / 1e6:.1f} ms, {len(seg)} kernels, busy {busy / 1e6:.1f} ms ({busy / wall:.2f}); gaps: "
          f"<5us {sum(g for g in gaps if g < 5000) / 1e6:.1f} ms ({sum(1 for g in gaps if g < 5000)}), 5-20us {sum(g for g in gaps if 5000 <= g < 20000) / 1e6:.1f} ms "
          f"({sum(1 for g in gaps if 5000 <= g < 20000)}), 20-200us {sum(g for g in gaps if 20000 <= g < 200000) / 1e6:.1f} ms ({sum(1 for g in gaps if 20000 <= g < 200000)}), "
          f">200us {sum(g for g in gaps if g >= 200000) / 1e6:.1f} ms ({sum(1 for g in gaps if g >= 200000)})")
    tot = Counter()
    cnt = Counter()
    for s, e, n in seg:
        key = n.split("(")[0][:90]
        tot[key] += e - s
        cnt[key] += 1
    for key, t in tot.most_common(12):
        print(f"     {t / 1e6:7.2f} ms  {cnt[key]:5d} x {t / cnt[key] / 1e3:7.1f} us  {key}")
