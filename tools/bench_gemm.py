"""vlmc_linear_fwd / vlmc_hessian_accum against the library GEMMs at the calibration replay's shapes (1x MI355X).
Interleaved rounds in one process, median of the rounds, random data."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from vlmc import ops  # noqa: E402

dev = "cuda:0"


def timeit(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    shapes = [  # (name, dtype, M = 128 samples x tokens, N, K)
        ("vit.qkv", torch.float16, 128 * 257, 4224, 1408), ("vit.proj", torch.float16, 128 * 257, 1408, 1408),
        ("vit.fc1", torch.float16, 128 * 257, 6144, 1408), ("vit.fc2", torch.float16, 128 * 257, 1408, 6144),
        ("t5enc.q", torch.bfloat16, 128 * 64, 2048, 2048), ("t5enc.wi", torch.bfloat16, 128 * 64, 5120, 2048),
        ("t5enc.wo", torch.bfloat16, 128 * 64, 2048, 5120), ("t5dec.q", torch.bfloat16, 128 * 16, 2048, 2048),
        ("t5dec.wi", torch.bfloat16, 128 * 16, 5120, 2048), ("t5dec.wo", torch.bfloat16, 128 * 16, 2048, 5120),
        ("one sample t5enc.q", torch.bfloat16, 64, 2048, 2048), ("one sample vit.fc1", torch.float16, 257, 6144, 1408),
        # one rank's share at 8 / 4 GPUs: 16 / 32 samples
        ("N=8 vit.fc1", torch.float16, 16 * 257, 6144, 1408), ("N=8 vit.fc2", torch.float16, 16 * 257, 1408, 6144),
        ("N=8 t5enc.wi", torch.bfloat16, 16 * 64, 5120, 2048), ("N=8 t5dec.q", torch.bfloat16, 16 * 16, 2048, 2048),
        ("N=4 vit.fc1", torch.float16, 32 * 257, 6144, 1408), ("N=4 t5enc.wi", torch.bfloat16, 32 * 64, 5120, 2048),
        ("N=8 t5enc.q", torch.bfloat16, 16 * 64, 2048, 2048), ("N=8 t5enc.wo", torch.bfloat16, 16 * 64, 2048, 5120),
        ("N=8 t5dec.wi", torch.bfloat16, 16 * 16, 5120, 2048), ("N=8 t5dec.wo", torch.bfloat16, 16 * 16, 2048, 5120),
        ("N=4 t5dec.q", torch.bfloat16, 32 * 16, 2048, 2048), ("N=4 t5dec.wo", torch.bfloat16, 32 * 16, 2048, 5120),
        ("one sample t5dec.q", torch.bfloat16, 16, 2048, 2048), ("one sample t5enc.wo", torch.bfloat16, 64, 2048, 5120),
    ]
    print(f"(VLMC_GEMM_EDGE={os.environ.get('VLMC_GEMM_EDGE', '1')} VLMC_GEMM_BIG_TILES={os.environ.get('VLMC_GEMM_BIG_TILES', '200')})")
    print("| linear | M x N x K | vlmc_linear_fwd us | TFLOP/s | library us | TFLOP/s |")
    print("|---|---|---|---|---|---|")
    cold = os.environ.get("BENCH_GEMM_COLD", "1") == "1"      # every launch reads another copy of W (as a replay does), not the
    print(f"(weights {'cold: a pool of copies > 768 MB per shape' if cold else 'hot: the same W every launch'})")     # Infinity Cache's
    for name, dt, M, N, K in shapes:
        x = (torch.randn(M, K, device=dev) * 0.5).to(dt)
        w0 = (torch.randn(N, K, device=dev) * 0.05).to(dt)
        pool = [w0] + [w0.clone() for _ in range((max(2, min(128, -(-(768 << 20) // (N * K * 2)))) if cold else 1) - 1)]
        reps = max(10, len(pool))
        state = {"i": 0}

        def nxt():
            state["i"] = (state["i"] + 1) % len(pool)
            return pool[state["i"]]
        ours, lib = [], []
        for _ in range(5):
            ours.append(timeit(lambda: ops.linear_fwd(x, nxt()), reps))
            lib.append(timeit(lambda: F.linear(x, nxt()), reps))
        fl = 2.0 * M * N * K
        to, tl = statistics.median(ours), statistics.median(lib)
        print(f"| {name} {str(dt)[6:]} | {M} x {N} x {K} | {to * 1e3:.1f} | {fl / to / 1e9:.0f} | {tl * 1e3:.1f} | {fl / tl / 1e9:.0f} |", flush=True)
    print()
    print("| linears sharing one input | M x (N...) x K | one vlmc_linear_fwd_group launch us | TFLOP/s | separate vlmc_linear_fwd launches us | TFLOP/s | library us | TFLOP/s |")
    print("|---|---|---|---|---|---|---|---|")
    for name, dt, M, Ns, K in [("t5enc q/k/v", torch.bfloat16, 128 * 64, (2048,) * 3, 2048),
                               ("t5enc wi_0/wi_1", torch.bfloat16, 128 * 64, (5120,) * 2, 2048),
                               ("t5dec q/k/v", torch.bfloat16, 128 * 16, (2048,) * 3, 2048),
                               ("t5dec cross k/v", torch.bfloat16, 128 * 64, (2048,) * 2, 2048),
                               ("t5dec wi_0/wi_1", torch.bfloat16, 128 * 16, (5120,) * 2, 2048),
                               ("vicuna q/k/v", torch.float16, 128 * 96, (4096,) * 3, 4096),
                               ("vicuna gate/up", torch.float16, 128 * 96, (11008,) * 2, 4096)]:
        x = (torch.randn(M, K, device=dev) * 0.5).to(dt)
        ws = [(torch.randn(n, K, device=dev) * 0.05).to(dt) for n in Ns]
        grp, sep, lib = [], [], []
        for _ in range(5):
            grp.append(timeit(lambda: ops.linear_fwd_group(x, ws), 10))
            sep.append(timeit(lambda: [ops.linear_fwd(x, w) for w in ws], 10))
            lib.append(timeit(lambda: [F.linear(x, w) for w in ws], 10))
        fl = 2.0 * M * sum(Ns) * K
        tg, ts, tl = statistics.median(grp), statistics.median(sep), statistics.median(lib)
        print(f"| {name} {str(dt)[6:]} | {M} x {Ns} x {K} | {tg * 1e3:.1f} | {fl / tg / 1e9:.0f} | {ts * 1e3:.1f} | {fl / ts / 1e9:.0f} "
              f"| {tl * 1e3:.1f} | {fl / tl / 1e9:.0f} |", flush=True)
    if os.environ.get("BENCH_GEMM_ONLY_LINEAR") == "1":
        return
    print()
    print("| Hessian | rows x in | vlmc_hessian_accum us (transpose + SYRK) | TFLOP/s (2 T in^2) | addmm_ fp32 us | TFLOP/s |")
    print("|---|---|---|---|---|---|")
    for name, dt, T, n in [("t5 d_model bf16", torch.bfloat16, 8192, 2048), ("t5 d_ff bf16", torch.bfloat16, 8192, 5120),
                           ("vit dim fp16", torch.float16, 32896, 1408), ("vit mlp fp16", torch.float16, 32896, 6144),
                           ("fp32 activations (3 bf16 planes)", torch.float32, 4096, 2048)]:
        x = (torch.randn(T, n, device=dev) * 0.5).to(dt)
        H = torch.zeros(n, n, device=dev)
        ours, lib = [], []
        for _ in range(3):
            ours.append(timeit(lambda: ops.hessian_accum(H, x, 0.5, 0.01), 5))
            lib.append(timeit(lambda: H.addmm_(x.float().t(), x.float(), beta=0.5, alpha=0.01), 5))
        fl = 2.0 * T * n * n
        to, tl = statistics.median(ours), statistics.median(lib)
        print(f"| {name} | {T} x {n} | {to * 1e3:.1f} | {fl / to / 1e9:.0f} | {tl * 1e3:.1f} | {fl / tl / 1e9:.0f} |", flush=True)


if __name__ == "__main__":
    main()
