#!/bin/bash
# DSnoT list kernel: per-kernel time + instruction mix (profiles/r02_dsnot_roofline.md).  Run on the GPU box.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/dsnot_prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 tools/bench_methods.py --only dsnot --reps 2 > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -o p -- python3 tools/bench_methods.py --only dsnot --reps 2 > $OUT/pmc.log 2>&1
python3 - <<'PY'
import csv, glob, collections
out = "gpurun_out/dsnot_prof"
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(out + "/kernel_stats.md", "w") as w:
        for r in rows:
            if "dsnot" in r["Name"] or "select" in r["Name"]:
                w.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} us |\n")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:90]
        if "dsnot_lists" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
with open(out + "/pmc.md", "w") as w:
    for k, d in acc.items():
        wv = d["SQ_WAVES"] or 1
        w.write(f"| `{k}` | launches {cnt[k]} | waves/launch {wv/cnt[k]:.0f} | VALU/wave {d['SQ_INSTS_VALU']/wv:.0f} | SALU/wave {d['SQ_INSTS_SALU']/wv:.0f} | LDS/wave {d['SQ_INSTS_LDS']/wv:.0f} | VMEM/wave {d['SQ_INSTS_VMEM']/wv:.0f} | wait/cycles {d['SQ_WAIT_ANY']/max(d['SQ_WAVE_CYCLES'],1):.2f} |\n")
PY
rm -rf $OUT/stats $OUT/pmc
cat $OUT/kernel_stats.md $OUT/pmc.md
