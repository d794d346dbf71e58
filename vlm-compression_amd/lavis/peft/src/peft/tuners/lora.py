"""SparseLoRA drop-in for `lavis/peft/src/peft/tuners/lora.py` (reference :39-87 LoraConfig,
:89-247 LoraModel, :250-265 mark_only_lora_as_trainable, :268-287 LoraLayer, :289-394
`Linear` -- the RESSA layer with a bool `mask` buffer, a `sparse` flag and a per-call
`dense` switch).

Same class names, constructor arguments, parameters (`weight`, `bias?`, `lora_A.weight`
[r,in] fp32, `lora_B.weight` [out,r] fp32), buffer (`mask`, persistent => in the
state_dict), attributes (`sparse, scaling, r, merged, disable_adapters, fan_in_fan_out`)
and methods (`forward(x, dense=False)`, `merge()`, `reset_peft()`, `train()`, `eval()`).

Where the arithmetic runs (vlmc.sparse_lora, gfx950 kernels):
  forward  sparse=True : y = x ((W + s*B@A) . M)^T + b      effective weight generated in
           sparse=False: y = x (W . M + s*B@A)^T + b        one fused kernel (rank-r MFMA
                                                            contraction + mask), main GEMM by
                                                            the library
  backward dA, dB from (dY^T x) . M in one fused pass; dx = dY W_eff
  merge()  in place, one fused pass
  dense=True / disable_adapters / merged: plain F.linear on W (the library GEMM), exactly as
  the reference.  `lora_dropout` is constructed but never applied.

MergedLinear and the bitsandbytes 8-bit variants of the reference file are not on the RESSA
path (enable_lora is None in every script; bitsandbytes is absent) and are not built:
`Linear8bitLt` exists only as a type so that `find_layers`' type list is the same.
"""
import math
import re
from dataclasses import dataclass, field
from typing import List, Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils import PeftConfig, PeftType, transpose


@dataclass
class LoraConfig(PeftConfig):
    r: int = field(default=8, metadata={"help": "Lora attention dimension"})
    target_modules: Optional[Union[List[str], str]] = field(default=None, metadata={
        "help": "List of module name suffixes or a regex of the module names to replace with Lora."})
    lora_alpha: int = field(default=None, metadata={"help": "Lora alpha"})
    lora_dropout: float = field(default=None, metadata={"help": "Lora dropout"})
    merge_weights: bool = field(default=False, metadata={"help": "Merge weights of the original model and the Lora model"})
    fan_in_fan_out: bool = field(default=False, metadata={"help": "True if the layer stores weight like (fan_in, fan_out)"})
    enable_lora: Optional[List[bool]] = field(default=None, metadata={"help": "Used with `lora.MergedLinear` (not built)."})
    bias: str = field(default="none", metadata={"help": "Bias type for Lora. Can be 'none', 'all' or 'lora_only'"})
    modules_to_save: Optional[List[str]] = field(default=None, metadata={"help": "Extra trainable modules"})

    def __post_init__(self):
        self.peft_type = PeftType.LORA


def mark_only_lora_as_trainable(model: nn.Module, bias: str = "none") -> None:
    for n, p in model.named_parameters():
        if "lora_" not in n:
            p.requires_grad = False
    if bias == "none":
        return
    if bias == "all":
        for n, p in model.named_parameters():
            if "bias" in n:
                p.requires_grad = True
    elif bias == "lora_only":
        for m in model.modules():
            if isinstance(m, LoraLayer) and hasattr(m, "bias") and m.bias is not None:
                m.bias.requires_grad = True
    else:
        raise NotImplementedError


class LoraLayer:
    def __init__(self, r: int, lora_alpha: int, lora_dropout: float, merge_weights: bool):
        self.r = r
        self.lora_alpha = lora_alpha
        if lora_dropout is not None and lora_dropout > 0.0:
            self.lora_dropout = nn.Dropout(p=lora_dropout)
        else:
            self.lora_dropout = lambda x: x
        self.merged = False
        self.merge_weights = merge_weights
        self.disable_adapters = False


class Linear8bitLt(nn.Linear):
    """Type placeholder (bitsandbytes is not part of this build); never instantiated."""

    def __init__(self, *a, **k):
        raise NotImplementedError("8-bit LoRA layers need bitsandbytes, which this build does not include")


class Linear(nn.Linear, LoraLayer):
    """SparseLoRA linear (lora.py:289-394)."""

    def __init__(self, in_features: int, out_features: int, r: int = 0, lora_alpha: int = 1, lora_dropout: float = 0.0,
                 fan_in_fan_out: bool = False, merge_weights: bool = True, **kwargs):
        nn.Linear.__init__(self, in_features, out_features, **kwargs)
        LoraLayer.__init__(self, r=r, lora_alpha=lora_alpha, lora_dropout=lora_dropout, merge_weights=merge_weights)
        self.fan_in_fan_out = fan_in_fan_out
        if r > 0:
            self.lora_A = nn.Linear(in_features, r, bias=False)
            self.lora_B = nn.Linear(r, out_features, bias=False)
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
        self.reset_parameters()
        if fan_in_fan_out:
            self.weight.data = self.weight.data.T
        self.register_buffer("mask", torch.ones_like(self.weight.data).bool())
        self.sparse = False

    def reset_parameters(self):
        nn.Linear.reset_parameters(self)
        self.reset_peft()

    def reset_peft(self):
        if hasattr(self, "lora_A"):
            nn.init.kaiming_uniform_(self.lora_A.weight, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B.weight)

    def train(self, mode: bool = True):
        """lora.py:333-352 (incl. merge-on-eval when `merge_weights`)."""
        nn.Linear.train(self, mode)
        if hasattr(self, "lora_A"):
            self.lora_A.train(mode)
            self.lora_B.train(mode)
        if not mode and self.merge_weights and not self.merged:
            if self.r > 0:
                self.weight.data += transpose(self.lora_B.weight @ self.lora_A.weight, self.fan_in_fan_out) * self.scaling
            self.merged = True
        elif self.merge_weights and self.merged:
            if self.r > 0:
                self.weight.data -= transpose(self.lora_B.weight @ self.lora_A.weight, self.fan_in_fan_out) * self.scaling
            self.merged = False
        return self

    def eval(self):
        nn.Linear.eval(self)
        if hasattr(self, "lora_A"):
            self.lora_A.eval()
            self.lora_B.eval()
        return self

    def forward(self, x: torch.Tensor, dense=False):
        previous_dtype = self.weight.dtype
        if dense or self.disable_adapters or not (self.r > 0 and not self.merged):
            if self.fan_in_fan_out:
                result = F.linear(x, transpose(self.weight, self.fan_in_fan_out), bias=self.bias)
            else:
                from vlmc import forward as _fw        # the calibration replay's batch-invariant kernel when it is active
                result = _fw.linear(x, self.weight, self.bias)
        else:
            if self.fan_in_fan_out:
                raise NotImplementedError("SparseLoRA kernels expect [out, in] weights (fan_in_fan_out=False)")
            from vlmc import sparse_lora
            result = sparse_lora.linear(x, self.weight, self.lora_A.weight, self.lora_B.weight, self.mask, self.bias,
                                        self.scaling, self.sparse)
        if result.dtype != previous_dtype:
            result = result.to(previous_dtype)
        return result

    def merge(self):
        """lora.py:384-394: sparse -> W += (s*B@A) . M ; else -> W[~M] = 0; W += s*B@A ; then re-init A, B."""
        if self.fan_in_fan_out:
            raise NotImplementedError("SparseLoRA kernels expect [out, in] weights (fan_in_fan_out=False)")
        from vlmc import sparse_lora
        sparse_lora.merge_(self.weight.data, self.lora_A.weight.data, self.lora_B.weight.data, self.mask, self.scaling,
                           self.sparse)
        self.reset_peft()


class LoraModel(torch.nn.Module):
    """Swap the targeted nn.Linear modules for SparseLoRA `Linear`s sharing weight and bias
    (lora.py:89-247)."""

    def __init__(self, config, model):
        super().__init__()
        self.peft_config = config
        self.model = model
        self._find_and_replace()
        mark_only_lora_as_trainable(self.model, self.peft_config.bias)
        self.forward = self.model.forward

    def _find_and_replace(self):
        if getattr(self.model, "is_loaded_in_8bit", False):
            raise ImportError("To use Lora with 8-bit quantization, please install the `bitsandbytes` package.")
        cfg = self.peft_config
        if cfg.enable_lora is not None:
            raise NotImplementedError("MergedLinear (enable_lora) is not on the RESSA path and is not built")
        kwargs = {
            "r": cfg.r, "lora_alpha": cfg.lora_alpha, "lora_dropout": cfg.lora_dropout,
            "fan_in_fan_out": cfg.fan_in_fan_out,
            "merge_weights": (cfg.merge_weights or cfg.inference_mode) and not hasattr(self.model, "hf_device_map"),
        }
        found = False
        for key in [k for k, _ in self.model.named_modules()]:
            if isinstance(cfg.target_modules, str):
                hit = re.fullmatch(cfg.target_modules, key)
            else:
                hit = any(key.endswith(t) for t in cfg.target_modules)
            if not hit:
                continue
            found = True
            parent, target, target_name = self._get_submodules(key)
            if not isinstance(target, torch.nn.Linear):
                continue
            bias = hasattr(target, "bias") and target.bias is not None
            new_module = Linear(target.in_features, target.out_features, bias=bias, **kwargs)
            self._replace_module(parent, target_name, new_module, target)
        if not found:
            raise ValueError(f"Target modules {cfg.target_modules} not found in the base model. "
                             f"Please check the target modules and try again.")

    def _get_submodules(self, key):
        parent = self.model.get_submodule(".".join(key.split(".")[:-1]))
        return parent, self.model.get_submodule(key), key.split(".")[-1]

    @staticmethod
    def _replace_module(parent, name, new, old):
        """Put the SparseLoRA layer where the plain linear was.  W and b are SHARED with the original module (the pruner
        zeroes `weight.data` in place and train.py:626-637 indexes it); only the adapters are new, and they follow the
        weight to its device (lora.py:196-208)."""
        new.weight = old.weight
        if old.bias is not None:
            new.bias = old.bias
        device = old.weight.device
        for adapter in (getattr(new, "lora_A", None), getattr(new, "lora_B", None)):
            if adapter is not None:
                adapter.to(device)
        new.mask = new.mask.to(device)
        setattr(parent, name, new)

    def __getattr__(self, name):
        """Anything the wrapper does not define is the wrapped model's (train.py reads `model.t5_model`, `.maybe_autocast`, ...
        through it)."""
        modules = self.__dict__.get("_modules", {})
        if name in modules:
            return modules[name]
        try:
            return super().__getattr__(name)
        except AttributeError:
            if "model" not in modules:
                raise
            return getattr(modules["model"], name)

    @property
    def modules_to_save(self):
        return None

    def get_peft_config_as_dict(self, inference: bool = False):
        """The adapter configuration as plain values (lora.py:221-225): enum members by value, `inference_mode` forced on request."""
        from dataclasses import asdict
        from enum import Enum
        config = {k: v.value if isinstance(v, Enum) else v for k, v in asdict(self.peft_config).items()}
        if inference:
            config["inference_mode"] = True
        return config

    def _set_adapter_layers(self, enabled=True):
        for module in self.model.modules():
            if isinstance(module, LoraLayer):
                module.disable_adapters = not enabled

    def enable_adapter_layers(self):
        """(lora.py:232-233) every SparseLoRA layer adds its adapter again."""
        self._set_adapter_layers(enabled=True)

    def disable_adapter_layers(self):
        """(lora.py:235-236) every SparseLoRA layer computes the plain `F.linear(x, W, b)` -- `PeftModel.disable_adapter()`."""
        self._set_adapter_layers(enabled=False)
