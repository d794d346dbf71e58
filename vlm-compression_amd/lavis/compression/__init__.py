"""`lavis.compression` drop-in: pruner registry side effects + `load_pruner`
(reference: lavis/compression/__init__.py:1-46)."""
from lavis.common.registry import registry
from lavis.compression.pruners.base_pruner import BasePruner
from lavis.compression.pruners.wanda_pruner import (  # noqa: F401  (registration)
    BLIPT5LayerWandaPruner, T5LayerWandaPruner, VITLayerWandaPruner,
)
from lavis.compression.pruners.sparsegpt_pruner import (  # noqa: F401  (registration)
    BLIPT5LayerSparseGPTPruner, T5LayerSparseGPTPruner, VITLayerSparseGPTPruner,
)
from lavis.compression.pruners.dsnot_pruner import (  # noqa: F401  (registration)
    BLIPT5LayerDSnoTPruner, T5LayerDSnoTPruner, VITLayerDSnoTPruner,
)
from lavis.compression.pruners.global_pruner import (  # noqa: F401  (registration)
    BLIPT5AMeZoPruner, BLIPT5AOBDPruner, BLIPT5MagPruner, BLIPT5RandPruner,
)

__all__ = ["BasePruner"]


def load_pruner(name, model, data_loader, cfg_path=None, cfg=None):
    """Same contract as the reference factory (:29-46): `cfg` is the driver's kwargs dict
    (train.py:488-512), `cfg_path` an OmegaConf yaml; an unknown name / bad kwargs prints the
    reference's message and exits with status 1."""
    if cfg_path is None and cfg is None:
        cfg = None
    elif cfg_path is not None:
        from omegaconf import OmegaConf
        cfg = OmegaConf.load(cfg_path)
    try:
        pruner = registry.get_pruner_class(name)(model=model, data_loader=data_loader, **cfg)
    except TypeError:
        print(f"Pruner {name} not found. Available pruners:\n" + ", ".join([str(k) for k in __all__]))
        exit(1)
    return pruner
