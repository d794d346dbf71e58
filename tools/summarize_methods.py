"""profiles/r01_method_kernels.md from `rocprofv3 --kernel-trace --stats -- python3 tools/bench_methods.py` (dir + log)."""
import csv, glob, sys
d, log, out = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ours = sorted([r for r in rows if 'vlmc::' in r['Name']], key=lambda r: -float(r["TotalDurationNs"]))
libs = sorted([r for r in rows if 'vlmc::' not in r['Name']], key=lambda r: -float(r["TotalDurationNs"]))
short = lambda n: n.split('(')[0].replace('void ', '')[:90]
o = ["# Round 1 -- kernels of the SparseLoRA / SparseGPT / DSnoT paths at model shapes (1x MI355X)\n",
     f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -d {d} -- python3 tools/bench_methods.py`",
     "(BASELINE.json configs 3-5 shapes: Vicuna-7B linears r=16 fp16 for SparseLoRA; FlanT5-XL linears, 128 x 64-token samples for",
     "SparseGPT; Vicuna-7B + ViT-g linears, wanda init, 100-cycle budget for DSnoT).\n",
     "## Kernel durations (rocprofv3, all calls of the run)\n", "| kernel | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
for r in ours:
    o.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} |")
o.append("\nLargest library kernels of the same run (SparseGPT's cholesky_inverse, sorts and GEMMs run in rocSOLVER / rocPRIM / hipBLASLt):\n")
o += ["| kernel | calls | total ms | avg us |", "|---|---|---|---|"]
for r in libs[:6]:
    o.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.1f} | {float(r['AverageNs'])/1e3:.1f} |")
o.append("\n## Per call (HIP events around the Python entry point, fresh operands; includes launch latencies from an idle stream)\n")
o += ["| kernel / step | shape | median us | note |", "|---|---|---|---|"]
seen = set()
for line in open(log):
    if line.startswith('| ') and not line.startswith('| kernel') and line not in seen:
        seen.add(line); o.append(line.rstrip())
open(out, 'w').write("\n".join(o) + "\n")
print(len(ours), "kernels,", len(seen), "per-call rows")
