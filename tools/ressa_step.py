"""RESSA step time on the synthetic InstructBLIP-Vicuna-7B (BASELINE.json config 5, one GPU's share):
SparseLoRA r=16 on the 224 LLaMA linears, 2:4 masks from the drop-in Wanda pruner (`lora_model=True`), then the drop-in
`ImageTextRetrainTask._train_inner_loop` (dense no-grad forward + sparse forward/backward + KL, AMP fp16, AdamW).
The same loop with the reference's tensor algebra for the layer (lora.py:359-380, torch ops) is timed beside it.

    python tools/ressa_step.py [--layers 32] [--batch 16] [--steps 4]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from lavis.compression import load_pruner  # noqa: E402
from lavis.peft.src.peft.tuners import lora  # noqa: E402
from lavis.tasks.image_text_retrain import ImageTextRetrainTask  # noqa: E402
from vlmc import synthetic  # noqa: E402


def wrap(model, r=16, alpha=32):
    n = 0
    for parent in list(model.llm_model.modules()):
        for cname, child in list(parent.named_children()):
            if type(child) is nn.Linear:
                new = lora.Linear(child.in_features, child.out_features, r=r, lora_alpha=alpha, bias=False, merge_weights=False,
                                  device="meta")
                new = new.to_empty(device=child.weight.device)
                new.weight = child.weight
                new.weight.requires_grad = False
                new.lora_A.weight.data = (torch.randn(r, child.in_features, device=child.weight.device) * 0.01)
                new.lora_B.weight.data = (torch.randn(child.out_features, r, device=child.weight.device) * 0.01)
                new.mask = torch.ones_like(child.weight.data, dtype=torch.bool)
                new.sparse = True
                setattr(parent, cname, new)
                n += 1
    return n


def reference_forward(self, x, dense=False):
    """lora.py:359-380 with torch ops: what the reference materialises on every call."""
    if dense:
        return F.linear(x, self.weight, self.bias)
    w = self.weight * self.mask
    delta = (self.lora_B.weight @ self.lora_A.weight) * self.scaling
    if self.sparse:
        delta = delta * self.mask
    return F.linear(x, w + delta.to(w.dtype), self.bias)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--calib", type=int, default=16)
    ap.add_argument("--variant", default="both", choices=["both", "ours", "ref"])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = synthetic.InstructBlipVicuna(depth=a.layers).to(dev).eval()
    synthetic.randomize_(model)
    n = wrap(model)
    calib = synthetic.calibration_batches(a.calib, dev, vocab=32000)
    cfg = dict(t5_prune_spec="32-0.5-1.0-1.0", vit_prune_spec="39-1.0-1.0-1.0", t5_pruning_method="wanda", vit_pruning_method="wanda",
               num_samples=a.calib, prune_n=2, prune_m=4, t5_model_prefix="llm_model", max_sparsity_per_layer=1.01)
    t0 = time.perf_counter()
    model, _ = load_pruner("blipt5_wanda_pruner", model, calib, cfg=cfg).prune(lora_model=True)
    torch.cuda.synchronize()
    lin = model.llm_model.model.layers[0].mlp.down_proj
    print(f"{n} SparseLoRA linears, 2:4 masks in {time.perf_counter() - t0:.1f} s (kept fraction {float(lin.mask.float().mean()):.3f})", flush=True)
    lora.mark_only_lora_as_trainable(model)
    one = synthetic.calibration_batches(a.batch, dev, vocab=32000, seed=3)
    batch = {k: torch.cat([b[k] for b in one], dim=0) for k in one[0]}
    tokens = batch["text_input"].shape[1] + batch["text_output"].shape[1] + 32

    class Sched:
        def step(self, cur_epoch, cur_step):
            pass

    def loop(label):
        params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.05)
        scaler = torch.amp.GradScaler("cuda")
        task = ImageTextRetrainTask()

        def batches():
            while True:
                yield dict(batch)
        task._train_inner_loop(epoch=0, iters_per_epoch=2, model=model, data_loader=batches(), optimizer=opt, lr_scheduler=Sched(),
                               scaler=scaler, log_freq=0, cuda_enabled=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        task._train_inner_loop(epoch=0, iters_per_epoch=a.steps, model=model, data_loader=batches(), optimizer=opt,
                               lr_scheduler=Sched(), scaler=scaler, log_freq=0, cuda_enabled=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(f"{label:46s} {dt * 1e3:8.1f} ms/step   {a.batch * tokens / dt:9.0f} tokens/s   "
              f"peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB   loss {task.loss_history[-1]:.4f}", flush=True)
        return dt

    ours = ref = None
    if a.variant in ("both", "ours"):
        ours = loop("SparseLoRA kernels (vlmc_lora_effective_weight/_grad)")
    if a.variant in ("both", "ref"):
        orig = lora.Linear.forward
        lora.Linear.forward = reference_forward
        try:
            ref = loop("reference tensor algebra (torch ops, lora.py:359-380)")
        finally:
            lora.Linear.forward = orig
    if ours and ref:
        print(f"speed-up of the step: {ref / ours:.2f}x")


if __name__ == "__main__":
    main()
