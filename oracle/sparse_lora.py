"""Oracle: SparseLoRA `Linear` forward / merge as plain PyTorch-CPU tensor algebra.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Restates
/root/reference/lavis/peft/src/peft/tuners/lora.py:359-394 (the `fan_in_fan_out=False` case);
gradients come from autograd on these expressions, exactly as in the reference.
Pinned against the reference by tests/golden/sparse_lora.npz (tests/test_oracle_golden.py).
"""
import torch
import torch.nn.functional as F


def effective_weight(weight, lora_A, lora_B, mask, scaling, sparse):
    """The weight handed to F.linear (lora.py:362-375)."""
    delta = (lora_B @ lora_A).to(weight.dtype) * scaling
    if sparse:
        return (weight + delta) * mask
    return weight * mask + delta


def forward(x, weight, lora_A, lora_B, mask, bias, scaling, sparse, dense=False):
    previous_dtype = weight.dtype
    if dense:
        result = F.linear(x, weight, bias=bias)                       # :361-362
    else:
        result = F.linear(x, effective_weight(weight, lora_A, lora_B, mask, scaling, sparse), bias=bias)
    return result.to(previous_dtype) if result.dtype != previous_dtype else result   # :379-380


@torch.no_grad()
def merge(weight, lora_A, lora_B, mask, scaling, sparse):
    """Returns the merged weight (lora.py:384-391); inputs untouched."""
    w = weight.clone()
    if sparse:
        w += (lora_B @ lora_A * scaling) * mask
    else:
        w[~mask] = 0
        w += lora_B @ lora_A * scaling
    return w
