"""`dsnot_lists_kernel` alone: time per launch against the cycle budget (extraction of the list heads scales with the rows,
the walk with rows x cycles), unstructured and 2:4.   python tools/dsnot_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import _lib, dsnot, ops  # noqa: E402
from vlmc.ops import _dtype_code, _stream  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
print("| shape | mode | " + " | ".join(f"{c} cycles us" for c in (1, 25, 50, 100)) + " |")
print("|---|---|---|---|---|---|")
for out_f, in_f in ((4096, 4096), (11008, 4096), (4096, 11008), (6144, 1408)):
    dt = torch.float16
    st = dsnot.DsnotInputStat(in_f, dev)
    for _ in range(16):
        st.add_call((torch.randn(1, 96, in_f, device=dev) + 0.2).to(dt))
    st.finalize()
    W = (torch.randn(out_f, in_f, device=dev) * 0.02).to(dt)
    for tag, n, m in (("unstructured", 0, 0), ("2:4", 2, 4)):
        if n:
            keep, _ = ops.wanda_select(W, st.sqrt_row, "nm", n=n, m=m, apply_zero=False)
        else:
            keep, _ = ops.wanda_select(W, st.sqrt_row, "row", k=in_f // 2, apply_zero=False)
        cells = []
        for mc in (1, 25, 50, 100):
            events = torch.empty((out_f, mc), dtype=torch.int32, device=dev)
            stop = torch.empty(out_f, dtype=torch.int32, device=dev)

            def run():
                _lib.check(lib.vlmc_dsnot_refine(W.data_ptr(), _dtype_code(W), out_f, in_f, W.stride(0), keep.data_ptr(),
                                                 st.sqrt_row.data_ptr(), st.sum_row.data_ptr(), st.var_row.data_ptr(), 1, n, m, mc, 0.1, 1.0, 1,
                                                 events.data_ptr(), stop.data_ptr(), _stream()))
            for _ in range(3):
                run()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                run()
            b.record()
            torch.cuda.synchronize()
            cells.append(f"{a.elapsed_time(b) * 100:.0f}")
        print(f"| {out_f}x{in_f} | {tag} | " + " | ".join(cells) + " |", flush=True)
