"""Race screen for the fused masked-LoRA GEMMs: rows of the identity as activations make every output ONE generated
weight; repeated launches at long K, both tile heights, are compared with the effective-weight kernel bit for bit.

    python tools/lora_gemm_stress.py [--reps 6]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402

from vlmc import _lib, sparse_lora as SL  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--full", action="store_true", help="only the full-identity case of 4096 x 11008")
    args = ap.parse_args()
    dev = "cuda:0"
    lib = _lib.load()
    wd, code, r = torch.float16, 1, 16
    st = torch.cuda.current_stream().cuda_stream
    for out_f, in_f in ([(4096, 11008)] if args.full else [(4096, 11008), (11008, 4096), (4096, 4096)]):
        g = torch.Generator(device=dev).manual_seed(1)
        W = (torch.randn(out_f, in_f, device=dev, generator=g) * 0.05).to(wd)
        A = torch.randn(r, in_f, device=dev, generator=g) * 0.1
        B = torch.randn(out_f, r, device=dev, generator=g) * 0.1
        Mk = torch.rand(out_f, in_f, device=dev, generator=g) > 0.5
        want = SL.effective_weight(W, A, B, Mk, 1.0, SL.FWD_SPARSE, code)
        prep = torch.empty(lib.vlmc_sparse_lora_prep_bytes(out_f, in_f), dtype=torch.uint8, device=dev)
        _lib.check(lib.vlmc_sparse_lora_prep(A.data_ptr(), B.data_ptr(), out_f, in_f, r, code, prep.data_ptr(), st))
        for M in ((in_f,) if args.full else (1536, 2048, in_f)):
            rows = torch.randperm(in_f, device=dev, generator=g)[:M]
            X = torch.zeros(M, in_f, dtype=wd, device=dev)
            X[torch.arange(M, device=dev), rows] = 1
            Y = torch.empty(M, out_f, dtype=wd, device=dev)
            ref = want[:, rows].t().contiguous()
            cols = torch.randperm(out_f, device=dev, generator=g)[:min(M, out_f)]
            Mx = cols.numel()
            dY = torch.zeros(Mx, out_f, dtype=wd, device=dev)
            dY[torch.arange(Mx, device=dev), cols] = 1
            dX = torch.empty(Mx, in_f, dtype=wd, device=dev)
            refx = want[cols].contiguous()
            for rep in range(args.reps):
                Y.zero_(); dX.zero_()
                _lib.check(lib.vlmc_sparse_lora_fwd(X.data_ptr(), M, in_f, W.data_ptr(), code, out_f, in_f, in_f, Mk.data_ptr(), prep.data_ptr(), r, 1.0, 1,
                                                    code, None, Y.data_ptr(), out_f, st))
                _lib.check(lib.vlmc_sparse_lora_bwd_input(dY.data_ptr(), Mx, out_f, W.data_ptr(), code, out_f, in_f, in_f, Mk.data_ptr(), prep.data_ptr(), r,
                                                          1.0, 1, code, dX.data_ptr(), in_f, st))
                torch.cuda.synchronize()
                bad_f = ((Y.float() - ref.float()).abs() > 1e-3).sum().item()
                bad_x = ((dX.float() - refx.float()).abs() > 1e-3).sum().item()
                print(f"{out_f}x{in_f} M={M} rep {rep}: fwd wrong entries {bad_f} of {Y.numel()}, dX wrong entries {bad_x} of {dX.numel()}", flush=True)
                if bad_f:
                    idx = torch.nonzero((Y.float() - ref.float()).abs() > 1e-3)
                    print("   fwd (m, o, k, got, want):", [(m, o, rows[m].item(), Y[m, o].item(), ref[m, o].item()) for m, o in idx[:12].tolist()])
                    tiles = {}
                    for m, o in idx.tolist():
                        tiles.setdefault((m // 256, o // 128), []).append((m % 256, o % 128, rows[m].item() // 64))
                    for tkey, lst in list(tiles.items())[:6]:
                        ms = sorted({x[0] for x in lst}); os_ = sorted({x[1] for x in lst}); ks = sorted({x[2] for x in lst})
                        print(f"   tile (bq, bp) = {tkey}: {len(lst)} wrong; m % 256 in {ms[:24]}; o % 128 in {os_[:20]}; K-steps {ks[:24]}")
                    print("   distinct m // 256:", sorted(set((idx[:, 0] // 256).tolist()))[:20], "distinct o // 128:", sorted(set((idx[:, 1] // 128).tolist()))[:20],
                          "k // 64:", sorted(set((rows[idx[:, 0]] // 64).tolist()))[:20], "o % 128 // 16:", sorted(set((idx[:, 1] % 128 // 16).tolist())))
                if bad_x:
                    idx = torch.nonzero((dX.float() - refx.float()).abs() > 1e-3)[:6]
                    print("   dX (m, i):", idx.tolist(), " k of those rows:", cols[idx[:, 0]].tolist())


if __name__ == "__main__":
    main()
