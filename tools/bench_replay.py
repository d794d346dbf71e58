"""Block-replay engine benchmark: one Wanda pass over ONE transformer block at FlanT5-XL /
ViT-g dimensions with random weights -- 128 calibration samples forwarded with the statistics hooks attached,
the block's linears pruned, 128 samples forwarded again -- per-sample (the reference's loop) vs batched replay.

    python tools/bench_replay.py            (GPU only; prints a markdown table)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn as nn
import toy_models
from lavis.compression.pruners import calibration as cal
from lavis.compression.pruners.wanda_pruner import T5LayerWandaPruner, VITLayerWandaPruner

dev = "cuda:0"


class Tower(nn.Module):
    """Just enough model around one block for `walk_blocks` + the Wanda block step."""
    def __init__(self, block, vit):
        super().__init__()
        self.blocks = nn.ModuleList([block])
        self.vit = vit

    def maybe_autocast(self, dtype=None):
        import contextlib
        return contextlib.nullcontext()


def run(kind, group, n=128, graph=False):
    torch.manual_seed(0)
    if kind == "vit":
        block = toy_models.ToyViTBlock(1408, 6144, heads=16).to(torch.float16)
        xs = [(torch.randn(1, 257, 1408) * 0.5).to(torch.float16).to(dev) for _ in range(n)]
        caches = [{"rel_pos_bias": None} for _ in range(n)]
        mode, tuple_out = "matrix", False
    else:
        block = toy_models.ToyT5Block(2048, 5120, heads=32, is_decoder=False).to(torch.bfloat16)
        xs = [(torch.randn(1, 64, 2048) * 0.5).to(torch.bfloat16).to(dev) for _ in range(n)]
        caches = [dict(attention_mask=None, position_bias=None, encoder_hidden_states=None, encoder_attention_mask=None,
                       encoder_decoder_position_bias=None, layer_head_mask=None, cross_attn_layer_head_mask=None)
                  for _ in range(n)]
        mode, tuple_out = "row", True
    model = Tower(block.to(dev).eval(), kind == "vit")
    os.environ["VLMC_BATCH_REPLAY"] = str(group)
    os.environ["VLMC_GRAPH_REPLAY"] = "1" if graph else "0"
    helper = object.__new__(VITLayerWandaPruner if kind == "vit" else T5LayerWandaPruner)
    helper.prune_n = helper.prune_m = 0

    class Ratio(dict):
        def __missing__(self, k):
            return 0.5

    def prune_block(i, layer, subset, run_pass, state):
        helper._wanda_block(i, subset, run_pass, n, 1, unstructured_mode=mode, module_to_process="blocks",
                            model_prefix="bench", sparsity_ratio=Ratio(), lora_model=True)   # weights stay dense: repeatable

    outs = [None] * n
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cal.walk_blocks(model, list(xs), outs, caches, "blocks", n, model.maybe_autocast, prune_block, tuple_out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt * 1e3


print("| block | replay | ms per block pass (2 x 128 forwards + statistics + select) | speed-up |")
print("|---|---|---|---|")
for kind, label in (("t5", "T5 encoder block 2048/5120, 64 tokens, bf16"), ("vit", "ViT-g block 1408/6144, 257 tokens, fp16")):
    base = run(kind, 1)
    print(f"| {label} | per sample, eager (the reference's loop) | {base:.1f} | 1.0x |", flush=True)
    ms = run(kind, 1, graph=True)
    print(f"| {label} | per sample, HIP graph (default; bit-identical) | {ms:.1f} | {base / ms:.1f}x |", flush=True)
    for g in (8, 32, 128):
        ms = run(kind, g)
        print(f"| {label} | {g} samples per forward (VLMC_BATCH_REPLAY) | {ms:.1f} | {base / ms:.1f}x |", flush=True)
