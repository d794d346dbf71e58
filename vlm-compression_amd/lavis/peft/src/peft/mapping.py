"""`get_peft_model` for the task types the RESSA driver uses (reference lavis/peft/src/peft/mapping.py:29-36, :187-212;
train.py:413-486: "CAUSAL_LM" for the language tower, "ViT", "Qformer")."""
from .peft_model import PeftModel, PeftModelForCausalLM, PeftModelForQformer, PeftModelForViT
from .utils import PeftType

# mapping.py:46-62: the linears LoRA targets when `target_modules` is not given, by `config.model_type`
TRANSFORMERS_MODELS_TO_LORA_TARGET_MODULES_MAPPING = {
    "t5": ["q", "v"], "mt5": ["q", "v"], "bart": ["q_proj", "v_proj"], "gpt2": ["c_attn"], "bloom": ["query_key_value"],
    "opt": ["q_proj", "v_proj"], "gptj": ["q_proj", "v_proj"], "gpt_neox": ["query_key_value"], "gpt_neo": ["q_proj", "v_proj"],
    "bert": ["query", "value"], "roberta": ["query", "value"], "xlm-roberta": ["query", "value"], "electra": ["query", "value"],
    "deberta-v2": ["query_proj", "value_proj"], "deberta": ["in_proj"], "layoutlm": ["query", "value"], "llama": ["q_proj", "v_proj"],
    "chatglm": ["query_key_value"], "vit": ["qkv"],
}
MODEL_TYPE_TO_PEFT_MODEL_MAPPING = {"CAUSAL_LM": PeftModelForCausalLM, "ViT": PeftModelForViT, "Qformer": PeftModelForQformer}


def get_peft_model(model, peft_config):
    if peft_config.peft_type != PeftType.LORA:
        raise NotImplementedError("only LoRA adapters are built (the SparseLoRA path)")
    if peft_config.target_modules is None:                         # mapping.py:152-158: the model type's default targets
        cfg = getattr(model, "config", None)
        cfg = cfg.to_dict() if hasattr(cfg, "to_dict") else (cfg if isinstance(cfg, dict) else getattr(cfg, "__dict__", {}))
        model_type = cfg.get("model_type")
        if model_type not in TRANSFORMERS_MODELS_TO_LORA_TARGET_MODULES_MAPPING:
            raise ValueError("Please specify `target_modules` in `peft_config`")
        peft_config.target_modules = list(TRANSFORMERS_MODELS_TO_LORA_TARGET_MODULES_MAPPING[model_type])
    peft_config.base_model_name_or_path = model.__dict__.get("name_or_path", None)
    if len(peft_config.target_modules) == 1:                       # mapping.py:159-161: one target = a fused projection (MergedLinear)
        peft_config.fan_in_fan_out = True
        peft_config.enable_lora = [True, False, True]
    if peft_config.inference_mode:
        peft_config.merge_weights = True
    task = getattr(peft_config.task_type, "value", peft_config.task_type)
    return MODEL_TYPE_TO_PEFT_MODEL_MAPPING.get(task, PeftModel)(model, peft_config)
