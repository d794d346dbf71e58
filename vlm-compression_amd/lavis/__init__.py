"""Drop-in `lavis` subset for the compression hot path (MI355X build).

Only the packages on the pruning / SparseLoRA path exist here:
`lavis.common.registry` (pruner table), `lavis.compression` (pruner API) and
`lavis.peft.src.peft.tuners.lora` (SparseLoRA Linear).  In the reference tree these
files replace their namesakes; everything else of LAVIS (models, runners, tasks,
datasets) stays as it is there.
"""
