"""Regenerate profiles/r01_bench_kernel_stats_v4.md, r01_pmc_traffic_v4.md, traffic.json and r01_bench_line.json from
   gpurun_out/{stats_v4, pmc_fetch_v4, pmc_write_v4, stats_v4.log, bench_v4_default.json} (see the commands in the files)."""
import csv, glob, io, json, os, re, subprocess, sys, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
subprocess.run([sys.executable, "tools/traffic_from_pmc.py", "gpurun_out/pmc_fetch_v4", "gpurun_out/pmc_write_v4", "profiles/traffic.json"],
               check=True, stdout=subprocess.DEVNULL)
t = json.load(open("profiles/traffic.json"))
d = json.loads(open("gpurun_out/bench_v4_default.json").read().strip().splitlines()[-1])
d["roofline"]["traffic"] = t.get("act_sqnorm_kernel_bytes_per_launch")
d["roofline"]["other"][0]["traffic"] = t.get("select_rows_mixed_kernel_bytes_per_launch")
json.dump(d, open("profiles/r01_bench_line.json", "w"), indent=1)
r, o = d["roofline"], d["roofline"]["other"][0]
subprocess.run([sys.executable, "tools/summarize_rocprof.py", "stats", "gpurun_out/stats_v4", "/tmp/stats_v4.md"], check=True, stdout=subprocess.DEVNULL)
stats = open("/tmp/stats_v4.md").read()
rows = {m.group(1): (int(m.group(2)), float(m.group(3))) for m in re.finditer(r"\| `([^`]+)` \| (\d+) \| [\d.]+ \| ([\d.]+) \|", stats)}
sq = [(c, a) for n, (c, a) in rows.items() if "act_sqnorm" in n]
sq_avg = sum(c * a for c, a in sq) / sum(c for c, _ in sq)
mixed_avg = [a for n, (c, a) in rows.items() if "select_rows_mixed" in n][0]
fused_avg = [a for n, (c, a) in rows.items() if "matrix_fused" in n][0]
prof = json.loads([l for l in open("gpurun_out/stats_v4.log") if l.startswith("{")][-1])
gaps = subprocess.run([sys.executable, "tools/trace_gaps.py", "gpurun_out/stats_v4"], check=True, capture_output=True, text=True).stdout
open("profiles/r01_bench_kernel_stats_v4.md", "w").write(f"""# Round 1 (v4: fused matrix-wide select, one mixed per-row launch per T5 block, launch-carried events) -- rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --e2e 0` (1x MI355X)

Command: `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_v4 -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --e2e 0`
(4 steps in the trace: 1 warm-up + 3 timed).  Bench line of the profiled run: {prof['value']} layers/s, {prof['ms_per_step']} ms/step,
`roofline` act_sqnorm_kernel {prof['roofline']['achieved']} GB/s (avg launch {prof['roofline']['avg_launch_us']} us by HIP events under the profiler; this table:
(f16 calls x avg + bf16 calls x avg) / calls = {sq_avg:.2f} us).
The un-profiled default run of the same build (`python bench.py`, 5 steps + end-to-end + CPU baseline leg, `profiles/r01_bench_line.json`):
**{d['value']} layers/s, {d['ms_per_step']} ms/step**; `roofline`: act_sqnorm_kernel {r['achieved']} GB/s = {r['frac']} of 8 TB/s,
avg launch {r['avg_launch_us']} us over {r['launches']} timed launches (this table: {sq_avg:.2f} us, {abs(r['avg_launch_us'] / sq_avg - 1) * 100:.1f} % apart), traffic/launch {t['act_sqnorm_kernel_bytes_per_launch']} B
vs 389734682 B algorithmic (average over all 87 blocks); `roofline.other[0]`: select_rows_mixed_kernel {o['achieved']} GB/s = {o['frac']},
avg launch {o['avg_launch_us']} us (table: {mixed_avg:.2f} us); `end_to_end`: {d['end_to_end']['seconds']} s ({d['end_to_end']['seconds_batched32']} s with 32 samples per forward);
`cpu_baseline`: {d['cpu_baseline']['value']:.3f} layers/s on {d['cpu_baseline']['cores']} host cores ({d['cpu_baseline']['sample'][:80]}...).

{stats}
Against v3 (`r01_bench_kernel_stats_v3.md`): the ViT-g block's matrix-wide select is ONE kernel of {fused_avg:.1f} us
(v3: sample 10.3 + count 25.6 + apply 24.9 + resolve 9.5 = 70.3 us in four launches), the T5 block's per-row select ONE
launch of {mixed_avg:.1f} us (v3: 47.3 + 20.2 = 67.5 us in two).

Idle time between consecutive kernels of the last step of this trace (`python tools/trace_gaps.py gpurun_out/stats_v4`).
Kernels follow each other without a gap unless the launch carries HIP events (here: every 4th block): a timed launch is
preceded and followed by ~5 us of idle GPU, whether the events are recorded by hipEventRecord (torch.cuda.Event, v3: 270
records = 1.4 ms = 11 % of a step) or carried by the dispatch.

```
{gaps}```
""")
k = t["kernels"]
rows_md = "".join(f"| `{n}` | {v['dispatches']} | {v['fetch_bytes_per_launch'] / 1e6:.2f} | {v['write_bytes_per_launch'] / 1e6:.2f} | {v['hbm_bytes_per_launch'] / 1e6:.2f} |\n" for n, v in k.items())
fk = [v for n, v in k.items() if "matrix_fused" in n][0]
step_gb = 33.9 + 13.6 + 39 * t["matrix_fused_kernel_bytes_per_launch"] / 1e9 + 0.6
open("profiles/r01_pmc_traffic_v4.md", "w").write(f"""# Round 1 (v4) -- HBM traffic per launch from PMC counters (1x MI355X)

Commands (separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes):
`rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_v4 -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --e2e 0`
`rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_v4 -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --e2e 0`
then `python tools/traffic_from_pmc.py gpurun_out/pmc_fetch_v4 gpurun_out/pmc_write_v4 profiles/traffic.json` (all of it: `python tools/write_profiles.py`).
Correction: bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE tallies 64 B per 128-B request of a wide coalesced read).

| kernel | dispatches | fetch MB | write MB | HBM MB / launch |
|---|---|---|---|---|
{rows_md}
Per kernel template (what `bench.py` reports as `roofline.traffic`) against the algorithmic bytes:

| kernel | measured MB / launch | algorithmic | ratio |
|---|---|---|---|
| `act_sqnorm_kernel` | {t['act_sqnorm_kernel_bytes_per_launch'] / 1e6:.1f} | 389.7 MB avg (2 B x tokens x in x 128 samples per distinct input + 4 B x in x 128) | {t['act_sqnorm_kernel_bytes_per_launch'] / 389734682:.3f} |
| `select_rows_mixed_kernel` | {t['select_rows_mixed_kernel_bytes_per_launch'] / 1e6:.1f} | 283.2 MB avg per T5 block (5 B per weight + 4 B x in) | {t['select_rows_mixed_kernel_bytes_per_launch'] / 283201536:.3f} |
| `matrix_fused_kernel` | {t['matrix_fused_kernel_bytes_per_launch'] / 1e6:.1f} | 126.2 MB per ViT-g block (5 B per weight) | {t['matrix_fused_kernel_bytes_per_launch'] / 126.2e6:.3f} |

The fused matrix-wide select reads W once (v3: count pass + apply pass = 182.0 MB per ViT-g block, ratio 1.44): what is left above the
algorithmic bytes is the 2048-element sample every workgroup draws for itself (random 2-byte reads, whole cache lines counted:
{fk['fetch_bytes_per_launch'] / 1e6:.1f} MB fetched vs 50.5 MB of weights) and the histogram / candidate exchange in the workspace ({fk['write_bytes_per_launch'] / 1e6 - 75.7:.1f} MB of writes beyond mask + W).

Whole step: 33.9 GB (statistics) + 13.6 GB (per-row select) + {39 * t['matrix_fused_kernel_bytes_per_launch'] / 1e9:.1f} GB (matrix-wide select) + 0.6 GB (running mean) = {step_gb:.1f} GB per
{d['ms_per_step']} ms step = {step_gb / d['ms_per_step']:.2f} TB/s sustained over the whole prune pass (algorithmic: 52.4 GB).
""")
print(d["value"], d["ms_per_step"], r["frac"], d["end_to_end"]["seconds"], d["end_to_end"]["seconds_batched32"], f"sq {sq_avg:.2f} mixed {mixed_avg:.2f} fused {fused_avg:.2f}")
