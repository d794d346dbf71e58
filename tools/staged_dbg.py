import os, sys, torch
ROOT='/root/repo'
sys.path[:0]=[ROOT, os.path.join(ROOT,'vlm-compression_amd'), os.path.join(ROOT,'tests')]
import toy_models
from lavis.compression.pruners import calibration as cal
from vlmc import forward
toy_models.ToyAttention.use_sdpa = os.environ.get("SDPA","1")=="1"
torch.manual_seed(0)
blks=[toy_models.ToyViTBlock(32,64).half().cuda().eval() for _ in range(3)]
for b in blks:
    for p in b.parameters():
        torch.nn.init.normal_(p, std=0.2)
x=(torch.randn(8,9,32,device='cuda')*0.5).half()
import contextlib
sg=cal.StagedGraphs(contextlib.nullcontext, False)
sg.max_rows=10**9
outs={}
for i,b in enumerate(blks):
    subset=cal.find_layers(b)
    with forward.invariant_linears(subset.values()):
        sig=cal.block_signature(b, subset)
        with torch.no_grad():
            eager=b(x, None)
        done,y=sg.run(b, subset, sig, ('g',), x, {"rel_pos_bias":None}, "full", {})
        print(i, done, None if y is None else float((y.float()-eager.float()).abs().max()), None if y is None else bool(torch.equal(y,eager)))
        if done:
            done2,y2=sg.run(b, subset, sig, ('g',), x, {"rel_pos_bias":None}, "full", {})
            print("  replay twice equal:", bool(torch.equal(y,y2)))
            ent=sg.classes[sig]["graphs"][("full",('g',))]
            # stage by stage
            seen=[]
            hs=[m.register_forward_hook(lambda mod,inp,out,s=seen: s.append((inp[0].clone(), out.clone()))) for m in subset.values()]
            with torch.no_grad(): b(x,None)
            for h in hs: h.remove()
            for (n,xin,out),(exin,eout) in zip(ent["records"],seen):
                print("   ",n,"in equal",bool(torch.equal(xin,exin)),"out equal",bool(torch.equal(out,eout)), float((out.float()-eout.float()).abs().max()))
