import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgelu_variants.so"))
lib.gelu_variant.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
bits = torch.arange(65536, dtype=torch.int32).to(torch.int16).cuda()
for code, dt in ((1, torch.float16), (2, torch.bfloat16)):
    x = bits.view(dt)
    want = torch.nn.functional.gelu(x).view(torch.int16)
    ok = ~x.float().isnan()
    for kind, name in enumerate(("plain, no contraction", "contract(fast)", "fma(h, erf, h)", "double", "x * normcdf(x)")):
        out = torch.empty_like(bits)
        lib.gelu_variant(kind, code, bits.data_ptr(), out.data_ptr(), 65536)
        d = ((out != want) & ok)
        print(dt, name, int(d.sum()), "of 65536 differ", [float(v) for v in x[d][:4].float()])
