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
    """(peft_model.py:663-705) the Q-Former's `BertLMHeadModel.forward` signature, positional `input_ids` included, with the
    defaults the reference hands on explicitly (`use_cache=True`, `is_decoder=True`, `reduction="mean"`, `return_logits=False`)."""

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, head_mask=None, query_embeds=None,
                encoder_hidden_states=None, encoder_attention_mask=None, labels=None, past_key_values=None, use_cache=True,
                output_attentions=None, output_hidden_states=None, return_dict=None, return_logits=False, is_decoder=True,
                reduction="mean"):
        return self.base_model(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids, head_mask=head_mask,
                               query_embeds=query_embeds, encoder_hidden_states=encoder_hidden_states,
                               encoder_attention_mask=encoder_attention_mask, labels=labels, past_key_values=past_key_values,
                               use_cache=use_cache, output_attentions=output_attentions, output_hidden_states=output_hidden_states,
                               return_dict=return_dict, return_logits=return_logits, is_decoder=is_decoder, reduction=reduction)
