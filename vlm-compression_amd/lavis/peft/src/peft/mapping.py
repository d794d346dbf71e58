"""`get_peft_model` for the task types the RESSA driver uses (reference lavis/peft/src/peft/mapping.py:29-36, :187-212;
train.py:413-486: "CAUSAL_LM" for the language tower, "ViT", "Qformer")."""
from .peft_model import PeftModel, PeftModelForCausalLM, PeftModelForQformer, PeftModelForViT
from .utils import PeftType

MODEL_TYPE_TO_PEFT_MODEL_MAPPING = {"CAUSAL_LM": PeftModelForCausalLM, "ViT": PeftModelForViT, "Qformer": PeftModelForQformer}


def get_peft_model(model, peft_config):
    if peft_config.peft_type != PeftType.LORA:
        raise NotImplementedError("only LoRA adapters are built (the SparseLoRA path)")
    if peft_config.target_modules is None:
        raise ValueError("Please specify `target_modules` in `peft_config`")
    peft_config.base_model_name_or_path = model.__dict__.get("name_or_path", None)
    if len(peft_config.target_modules) == 1:                       # mapping.py:159-161: one target = a fused projection (MergedLinear)
        peft_config.fan_in_fan_out = True
        peft_config.enable_lora = [True, False, True]
    if peft_config.inference_mode:
        peft_config.merge_weights = True
    task = getattr(peft_config.task_type, "value", peft_config.task_type)
    return MODEL_TYPE_TO_PEFT_MODEL_MAPPING.get(task, PeftModel)(model, peft_config)
