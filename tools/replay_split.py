"""Where does a per-sample graph replay of one block spend its time?  Host issue time vs. GPU time of the hooked
(statistics) pass and of the plain pass, T5 encoder block and ViT-g block at model dimensions.
    python tools/replay_split.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import toy_models
from lavis.compression.pruners import calibration as cal
from lavis.compression.pruners.wanda_pruner import WandaStatCollector
import contextlib

dev, n = "cuda:0", 128
for kind in ("t5", "vit"):
    torch.manual_seed(0)
    if kind == "vit":
        block = toy_models.ToyViTBlock(1408, 6144, heads=16).to(torch.float16).to(dev).eval()
        xs = [(torch.randn(1, 257, 1408) * 0.5).to(torch.float16).to(dev) for _ in range(n)]
        cache, tup = {"rel_pos_bias": None}, False
    else:
        block = toy_models.ToyT5Block(2048, 5120, heads=32, is_decoder=False).to(torch.bfloat16).to(dev).eval()
        xs = [(torch.randn(1, 64, 2048) * 0.5).to(torch.bfloat16).to(dev) for _ in range(n)]
        cache, tup = dict(attention_mask=None, position_bias=None, encoder_hidden_states=None, encoder_attention_mask=None,
                          encoder_decoder_position_bias=None, layer_head_mask=None, cross_attn_layer_head_mask=None), True
    subset = cal.find_layers(block)
    g = cal.BlockGraph(block, xs[0], cache, subset, contextlib.nullcontext, tup)
    for label, hooked in (("plain pass", False), ("hooked pass", True), ("replay only", None)):
        for rep in range(3):
            col = WandaStatCollector(subset) if hooked else None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for j in range(n):
                if hooked is None:
                    g.graph.replay()
                else:
                    if col: col.next_sample(j)
                    g.run(xs[j], cache)
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            if col: col.close()
        print(f"{kind:4s} {label:12s}: host issue {1e6*(t1-t0)/n:7.1f} us/sample, until done {1e6*(t2-t0)/n:7.1f} us/sample", flush=True)
