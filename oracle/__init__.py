"""CPU oracle for the lavis.compression / SparseLoRA hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement of the reference's
algorithms (Shwai-He/VLM-Compression, `lavis/compression/pruners/*` and
`lavis/peft/src/peft/tuners/lora.py`).  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it; the product path
(`vlm-compression_amd/`) never does and fails loudly when the HIP library is
missing.

Parity pinning: the reference's own tests hold no golden vectors for this path
(SURVEY.md F2), so the oracle is pinned against outputs of the reference itself,
generated in the build container by `tests/golden/make_golden.py` (which imports
the reference from /root/reference) and committed as `tests/golden/*.npz`.
`tests/test_oracle_golden.py` checks every oracle function against them.
"""
