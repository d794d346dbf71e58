"""`lavis.tasks` subset: the RESSA retraining task (the caller of the SparseLoRA kernels)."""
from lavis.tasks.image_text_retrain import ImageTextRetrainTask  # noqa: F401
