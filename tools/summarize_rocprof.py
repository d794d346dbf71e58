"""Condense rocprofv3 CSV output into the small summaries kept under profiles/.

    python tools/summarize_rocprof.py stats  <dir> <out.md>    # --kernel-trace --stats
    python tools/summarize_rocprof.py pmc    <dir> <out.md>    # --pmc ... (per-kernel mean of each counter)
Only kernels of this repo (vlmc::) are listed individually; everything else is summed."""
import csv, glob, os, sys, collections


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    n = name.replace("(anonymous namespace)::", "").split("(")[0]
    return n.replace("void ", "")[:110]


def stats(d, out):
    f = find(d, "*kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    ours = [r for r in rows if "vlmc::" in r["Name"]]
    other = [r for r in rows if "vlmc::" not in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as o:
        o.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in sorted(ours, key=lambda r: -float(r["TotalDurationNs"])):
            o.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                    f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {100*float(r['TotalDurationNs'])/tot:.1f} |\n")
        oc = sum(int(r["Calls"]) for r in other)
        ot = sum(float(r["TotalDurationNs"]) for r in other)
        o.write(f"| (all non-vlmc kernels: torch fills/copies/RNG of the harness) | {oc} | {ot/1e6:.3f} | | | | {100*ot/tot:.1f} |\n")
    print(open(out).read())


def stats_all(d, out):
    """Every kernel by name (torch's own elementwise / copy kernels too): where a prune's GPU time goes."""
    f = find(d, "*kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as o:
        o.write(f"total kernel time {tot/1e6:.1f} ms, {sum(int(r['Calls']) for r in rows)} launches\n\n")
        o.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:70]:
            o.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                    f"{100*float(r['TotalDurationNs'])/tot:.1f} |\n")
        ours = sum(float(r["TotalDurationNs"]) for r in rows if "vlmc::" in r["Name"])
        o.write(f"\nvlmc:: kernels {ours/1e6:.1f} ms ({100*ours/tot:.1f} %), everything else {(tot-ours)/1e6:.1f} ms\n")
    print(open(out).read())


def pmc(d, out):
    f = find(d, "*counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "vlmc::" in r["Kernel_Name"]:
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(out, "a") as o:
        for k, cs in acc.items():
            o.write(f"\n`{k}` ({len(next(iter(cs.values())))} dispatches), mean per dispatch:\n\n")
            for c, v in sorted(cs.items()):
                o.write(f"- {c}: {sum(v)/len(v):.1f}\n")
    print(open(out).read())


if __name__ == "__main__":
    {"stats": stats, "stats_all": stats_all, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
