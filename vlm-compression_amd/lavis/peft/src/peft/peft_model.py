"""The PEFT wrappers the RESSA driver puts around each tower (`train.py:413-486` -> `get_peft_model`): reference
lavis/peft/src/peft/peft_model.py:46-88, :269-322 (`PeftModel`), :495-521 (`PeftModelForCausalLM`), :656-706
(`PeftModelForViT`, `PeftModelForQformer`).  LoRA only -- the prompt-learning / bottleneck branches of the reference file
are not on the SparseLoRA path.  What matters to the path is the MODULE TREE these wrappers create, because it names the
tensors of the saved checkpoint: `<tower>.base_model.model.<original key>` (vlmc/formats.py strips exactly that on reload,
evaluate_new.py:229-231).
"""
from contextlib import contextmanager

import torch

from .tuners.lora import LoraModel
from .utils import PeftConfig, PeftType


class PeftModel(torch.nn.Module):
    def __init__(self, model, peft_config: PeftConfig):
        super().__init__()
        if peft_config.peft_type != PeftType.LORA:
            raise NotImplementedError("only LoRA adapters are built (the SparseLoRA path)")
        self.peft_config = peft_config
        self.config = getattr(model, "config", None)
        self.modules_to_save = getattr(peft_config, "modules_to_save", None)
        self.base_model = LoraModel(peft_config, model)
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.base_model_torch_dtype = getattr(model, "dtype", None)

    def __getattr__(self, name: str):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.base_model, name)                  # (LoraModel forwards to the wrapped model in turn)

    def get_base_model(self):
        return self.base_model.model

    def forward(self, *args, **kwargs):
        return self.get_base_model()(*args, **kwargs)

    @contextmanager
    def disable_adapter(self):
        self.base_model.disable_adapter_layers()
        try:
            yield
        finally:
            self.base_model.enable_adapter_layers()

    def print_trainable_parameters(self):
        trainable = total = 0
        for p in self.parameters():
            n = p.numel() or getattr(p, "ds_numel", 0)
            total += n
            trainable += n if p.requires_grad else 0
        print(f"trainable params: {trainable} || all params: {total} || trainable%: {100 * trainable / total}")


class PeftModelForCausalLM(PeftModel):
    """Language towers (`task_type="CAUSAL_LM"`, train.py:418,432,446): every keyword goes to the wrapped model, which is
    how the reference's forward ends for LoRA adapters (peft_model.py:537-549)."""

    def forward(self, *args, **kwargs):
        return self.base_model(*args, **kwargs)


class PeftModelForViT(PeftModel):
    def forward(self, image, sparse=False):
        return self.base_model(image, sparse)


class PeftModelForQformer(PeftModel):
    def forward(self, **kwargs):
        return self.base_model(**kwargs)
