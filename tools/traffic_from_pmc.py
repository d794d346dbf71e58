"""profiles/traffic.json from two rocprofv3 PMC passes over bench.py (FETCH_SIZE, WRITE_SIZE).

Per /opt/skills/guides/MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled here;
WRITE_SIZE is exact for 16-B-per-lane streaming stores (our W stores; the 8-B mask stores agree
with the byte count to <1 %, see the per-shape check in the output).

    python tools/traffic_from_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/traffic.json
"""
import csv, glob, json, os, sys, collections


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "vlmc::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0",
       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950 FETCH_SIZE counts 64 B per 128-B request)",
       "kernels": {}}
agg = collections.defaultdict(lambda: [0.0, 0])
for k in sorted(fetch):
    f, w = fetch[k], write.get(k, [0])
    fb, wb = 2 * sum(f) / len(f) * 1024, sum(w) / len(w) * 1024
    out["kernels"][k] = {"dispatches": len(f), "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                         "hbm_bytes_per_launch": round(fb + wb)}
    base = k.split("<")[0].split("::")[-1]                       # all instantiations of one kernel template together
    agg[base][0] += (fb + wb) * len(f)
    agg[base][1] += len(f)
for base, (tot, n) in agg.items():
    out[f"{base}_bytes_per_launch"] = round(tot / n)
gemm = [(v["hbm_bytes_per_launch"], v["dispatches"]) for k, v in out["kernels"].items() if "gemm_nt" in k]
if gemm:                                                          # the three GEMM kernels are one entry point (vlmc_linear_fwd)
    out["gemm_nt_bytes_per_launch"] = round(sum(b * n for b, n in gemm) / sum(n for _, n in gemm))
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
