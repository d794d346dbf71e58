"""Which formulation of 1 / sqrt(x) reproduces torch.rsqrt bit for bit (tools/micro/rsqrt_probe.hip)?  Run on the GPU box."""
import ctypes, os, subprocess, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
outs = {}
for tag, flags in (("plain", []), ("library-flags", ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math"]),
                   ("no-correct-sqrt", ["-fno-hip-fp32-correctly-rounded-divide-sqrt"])):
    so = f"/tmp/rsqrt_probe_{tag}.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-shared", "--offload-arch=gfx950", *flags, os.path.join(root, "tools/micro/rsqrt_probe.hip"), "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lib.run_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    g = torch.Generator(device="cuda:0").manual_seed(0)
    x = torch.cat([torch.rand(1 << 20, generator=g, device="cuda:0") * 4 + 1e-6, torch.randn(1 << 20, generator=g, device="cuda:0").abs() * 1e3 + 1e-8,
                   torch.rand(1 << 18, generator=g, device="cuda:0") * 1e-4 + 1e-7])
    want = torch.rsqrt(x)
    y = torch.empty_like(x)
    for mode in (1, 2, 12, 13, 14):
        lib.run_probe(x.data_ptr(), y.data_ptr(), x.numel(), mode, None)
        torch.cuda.synchronize()
        print(tag, "mode", mode, "mismatches", int((y != want).sum()), "of", x.numel(), flush=True)
