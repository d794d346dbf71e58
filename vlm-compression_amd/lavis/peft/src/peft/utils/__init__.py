"""The three names `tuners/lora.py` needs from `peft.utils` (reference:
lavis/peft/src/peft/utils/config.py:27-120, other.py:158-159).  In the reference tree the
full PEFT utils package (save/load, hub mixins, task types) stays in place -- it is host
glue outside the hot path; only `lora.py` is replaced."""
import enum
from dataclasses import asdict, dataclass, field
from typing import Optional, Union


class PeftType(str, enum.Enum):
    PROMPT_TUNING = "PROMPT_TUNING"
    P_TUNING = "P_TUNING"
    PREFIX_TUNING = "PREFIX_TUNING"
    LORA = "LORA"
    BOTTLENECK = "BOTTLENECK"


class TaskType(str, enum.Enum):
    SEQ_CLS = "SEQ_CLS"
    SEQ_2_SEQ_LM = "SEQ_2_SEQ_LM"
    CAUSAL_LM = "CAUSAL_LM"
    TOKEN_CLS = "TOKEN_CLS"


@dataclass
class PeftConfig:
    peft_type: Optional[Union[str, PeftType]] = field(default=None, metadata={"help": "Peft type"})
    base_model_name_or_path: Optional[str] = field(default=None, metadata={"help": "The name of the base model to use."})
    task_type: Optional[Union[str, TaskType]] = field(default=None, metadata={"help": "Task type"})
    inference_mode: bool = field(default=False, metadata={"help": "Whether to use inference mode"})

    def to_dict(self):
        return asdict(self)


def transpose(weight, fan_in_fan_out):
    return weight.T if fan_in_fan_out else weight
