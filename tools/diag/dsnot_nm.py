import sys; sys.path.insert(0,'vlm-compression_amd'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
from oracle import dsnot as OD
from vlmc import dsnot
DEV='cuda:0'
def run(out_f,in_f,n,m,mc=60):
    g = torch.Generator().manual_seed(in_f + out_f)
    W = (torch.randn(out_f, in_f, generator=g) * 0.02).to(torch.bfloat16)
    xs = [((torch.randn(1, 9, in_f, generator=g) * 0.5) + 0.2).to(torch.bfloat16) for _ in range(4)]
    st = dsnot.DsnotInputStat(in_f, DEV)
    for x in xs: st.add_call(x.to(DEV))
    st.finalize()
    ost = OD.DSnoTStat(in_f)
    ost.scaler_row, ost.sum_metric_row, ost.var = st.scaler_row.cpu(), st.sum_row.cpu(), st.var_row.cpu().reshape(-1, 1)
    Wd = W.clone().to(DEV)
    keep = dsnot.prune_linear(Wd, st, 0.5, prune_n=n, prune_m=m, max_cycle_time=mc, update_threshold=0.05)
    want = OD.prune_nm(W, ost, n, m, max_cycle_time=mc, update_threshold=0.05)
    d = keep.cpu() != ~want
    rows = d.any(1).nonzero().flatten().tolist()
    print((out_f,in_f,n,m,mc), 'diff entries', int(d.sum()), 'rows', rows[:5])
    if rows:
        r = rows[0]; cols = d[r].nonzero().flatten().tolist()
        print('   row', r, 'cols', cols[:12], 'groups', sorted({c//m for c in cols})[:8])
for args in [(24,5120,4,8),(24,5120,2,4),(24,2048,4,8),(24,4096,4,8),(24,1408,4,8),(24,5120,4,8,5),(24,5120,4,8,1)]:
    run(*args)
