"""Per-kernel means of a rocprofv3 --pmc run over tools/bench_gemm.py (or bench.py), GEMM kernels only.
    python tools/pmc_gemm.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no counter_collection.csv under", d)
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "gemm_nt" in k or "Cijk" in k:
            acc[k[:96]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k)
        print("    calls", len(next(iter(v.values()))), {c: round(sum(x) / len(x)) for c, x in sorted(v.items())})
