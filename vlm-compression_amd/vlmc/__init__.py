"""vlmc -- host-side binding of the gfx950 pruning / SparseLoRA kernels.

`vlmc._lib` loads the C-ABI shared library (include/vlmc.h) with ctypes;
`vlmc.ops` wraps its entry points for torch tensors (device pointers, current HIP
stream, caller-owned workspace).  There is NO CPU fallback: every op raises if the
library is missing or the tensors are not on a GPU.
"""
from . import crosscheck as _crosscheck

_crosscheck.apply()        # VLMC_CROSSCHECK=name,... -> the individual switches, before anything reads them

from . import _lib, dsnot, forward, ops, phases, shard, sparse_lora, sparsegpt, wanda, workload  # noqa: F401,E402

__all__ = ["_lib", "dsnot", "forward", "ops", "phases", "shard", "sparse_lora", "sparsegpt", "wanda", "workload"]
