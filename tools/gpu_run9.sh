cd /root/repo
export TMPDIR=/tmp
python - <<'PY' 2>&1 | grep -v amdgpu | tail -12
import sys
sys.path.insert(0, "/root/repo/vlm-compression_amd")
import torch
from vlmc import synthetic, forward
from lavis.compression.pruners import calibration as cal
dev = torch.device("cuda:0")
for refops, ragged in ((False, False), (True, True)):
    model = synthetic.InstructBlipT5(reference_ops=refops).to(dev).eval()
    batches = synthetic.calibration_batches(128, dev, vocab=model.t5_model.shared.num_embeddings, ragged=ragged)
    for it in range(4):
        g0, f0 = dict(cal.graph_stats), dict(forward.stats)
        dt, _, info = synthetic.time_prune(dev, model=model, batches=batches)
        gs = {k: v - g0.get(k, 0) for k, v in cal.graph_stats.items() if v != g0.get(k, 0)}
        fs = {k: forward.stats[k] - f0[k] for k in f0 if forward.stats[k] != f0[k]}
        print(f"refops={refops} ragged={ragged} it={it} {dt:.3f}s pruned={info['pruned_fraction']:.4f} graph_stats={gs} forward={fs}", flush=True)
    del model
    torch.cuda.empty_cache()
PY
