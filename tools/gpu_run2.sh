set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_refops -o refops -- python3 tools/refops_probe.py ${1:-0} ${2:-1} > gpurun_out/refops_prof.log 2>&1 || { tail -20 gpurun_out/refops_prof.log; exit 1; }
tail -2 gpurun_out/refops_prof.log
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_refops/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("gpurun_out/refops_kernel_stats.txt", "w") as out:
    out.write(f"total kernel time {tot/1e6:.1f} ms over the 4 prunes of the probe (1 cold)\n")
    for r in rows[:50]:
        line = f'{r["Name"][:100]:100s} {int(r["Calls"]):7d} {float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["AverageNs"])/1e3:9.1f} us {100*float(r["TotalDurationNs"])/tot:5.1f}%'
        print(line); out.write(line + "\n")
PY
