import os, sys, json, statistics, subprocess
ROOT='/root/repo'
CHILD = r"""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.join('/root/repo', 'vlm-compression_amd'))
from vlmc import ops
dev='cuda:0'
def timeit(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
res=[]
for (M,N,K) in [(256,65536,1408),(128,65536,1408),(65536,256,1408),(65536,128,1408),(128,65536*2,1408),(256,65536*2,1408),(128,65536,6144),(256,65536,6144)]:
    x=(torch.randn(M,K,device=dev)*0.5).half(); w=(torch.randn(N,K,device=dev)*0.05).half()
    ts=[timeit(lambda: ops.linear_fwd(x,w),10) for _ in range(5)]
    res.append((M,N,K,round(statistics.median(ts)*1e3,1)))
print(json.dumps(res))
"""
for tune in ("0","1","2"):
    r=subprocess.run([sys.executable,"-c",CHILD],env=dict(os.environ,VLMC_GEMM_TUNE=tune),capture_output=True,text=True)
    print("TUNE",tune,r.stdout.strip().splitlines()[-1] if r.returncode==0 else r.stderr[-800:])
