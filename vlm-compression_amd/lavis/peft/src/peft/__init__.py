from .mapping import MODEL_TYPE_TO_PEFT_MODEL_MAPPING, get_peft_model  # noqa: F401
from .peft_model import PeftModel, PeftModelForCausalLM, PeftModelForQformer, PeftModelForViT  # noqa: F401
from .tuners.lora import LoraConfig, LoraModel  # noqa: F401
