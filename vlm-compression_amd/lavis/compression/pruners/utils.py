"""`print_time` and the calibration loss callbacks of lavis/compression/pruners/utils.py:6-44."""
import functools
from time import time


def print_time(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        start = time()
        ret = func(*args, **kwargs)
        print(f"{func.__name__} spent {time() - start:.3f} s")
        return ret
    return wrapper


def _prepare_sample(samples, cuda_enabled=True):
    import torch
    if not cuda_enabled:
        return samples
    return {k: (v.cuda(non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in samples.items()}


def loss_vision_language(model, samples, cuda_enabled):
    """(loss, batch size) of one calibration batch (utils.py:21-31)."""
    samples = _prepare_sample(samples, cuda_enabled=cuda_enabled)
    loss = model(samples)["loss"]
    return loss, len(samples["text_input"])


loss_language = loss_vision_language          # identical bodies in the reference (utils.py:34-44)
