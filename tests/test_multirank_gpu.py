"""Multi-rank paths on real GPUs.

* `bench.py --gpus 2` starts its own ranks (a fresh child process running torch.distributed.run; the parent never
  touches the GPU) -- rehearsed on one device over gloo (`VLMC_BENCH_ONE_DEVICE=1`), and it fails loudly without that
  switch when the box has fewer GPUs than ranks;
* with >= 2 GPUs visible: the sample-sharded pruner over RCCL (`backend="nccl"`), every rank bit-identical to the
  single-process run (skipped on a one-GPU box: RCCL refuses two ranks on one device)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_toy_attention():
    import toy_models
    yield
    toy_models.ToyAttention.use_sdpa = False


HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run(cmd, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("world", [2, 4])          # (8 ranks on one device would exceed the box's limit of 6 GPU processes:
def test_bench_launches_its_own_ranks_and_shards_the_prune(world):      # tests/test_dist_world48.py covers 8 over gloo on the CPU)
    r = _run([sys.executable, "bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "1", "--cpu-seconds", "0", "--kernel-pass", "0"],
             env={"VLMC_BENCH_ONE_DEVICE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE json line
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["config"]["world_size"] == world and out["config"]["backend"] == "gloo"
    assert out["config"]["parallelism"] == f"calib-dp{world}" and out["scaling"] == "strong"
    assert out["value"] > 0 and abs(out["value"] - 588 / (out["ms_per_step"] * 1e-3)) < 1.0
    assert out["config"]["pruned_fraction"] == pytest.approx(0.5, abs=1e-4)
    assert out["cpu_baseline"] is None and out["kernel_pass"] is None


@pytest.mark.timeout(300)
def test_bench_refuses_more_ranks_than_gpus():
    if torch.cuda.device_count() >= 8:
        pytest.skip("needs a box with fewer than 8 GPUs")
    r = _run([sys.executable, "bench.py", "--gpus", "8", "--steps", "1", "--warmup", "0"], env={"VLMC_BENCH_ONE_DEVICE": "0"}, timeout=200)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
    # under a launcher with a wrong world size it does not run one rank silently either
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"],
             env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, timeout=200)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path[:0] = [{root!r}, os.path.join({root!r}, "vlm-compression_amd"), os.path.join({root!r}, "tests")]
rank = int(os.environ["RANK"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
import pruner_helpers as H, toy_models
toy_models.ToyAttention.use_sdpa = {attention!r}
out = {{tag: {{k: v.cpu() for k, v in H.run_16bit_toy(tag, f"cuda:{{rank}}", ragged=(tag == "dsnot")).items()}} for tag in ("wanda", "dsnot")}}
torch.save(out, os.path.join({out!r}, f"rank{{rank}}.pt"))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.timeout(900)
def test_sample_sharded_pruners_over_rccl_equal_single_process(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank; this box has one")
    import pruner_helpers as H
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, out=str(tmp_path), attention=True))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), str(script)], env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    # the toy in the real model's dtypes: its linears run on the batch-invariant kernel, so sharding the samples (other
    # groups per rank) cannot change a bit
    import toy_models
    toy_models.ToyAttention.use_sdpa = True        # (restored by the fixture below)
    for tag in ("wanda", "dsnot"):
        single = {k: v.cpu() for k, v in H.run_16bit_toy(tag, "cuda:0", ragged=(tag == "dsnot")).items()}
        for rank in range(2):
            got = torch.load(tmp_path / f"rank{rank}.pt")[tag]
            assert got.keys() == single.keys()
            for k in single:
                assert torch.equal(got[k], single[k]), (rank, tag, k)              # bit-identical for any world size


_WORKER_GLOO = _WORKER.replace('torch.cuda.set_device(rank)', 'torch.cuda.set_device(0)') \
    .replace('dist.init_process_group("nccl", device_id=torch.device("cuda", rank))', 'dist.init_process_group("gloo")') \
    .replace('f"cuda:{{rank}}"', '"cuda:0"')


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,attention", [(2, True), (2, "matmul16"), (4, "matmul16")])
def test_sample_sharding_with_the_kernels_is_bit_identical_to_single_process(world, attention, tmp_path):
    """Two / four ranks sharing cuda:0 (collectives over gloo): each captures and replays its share of the calibration samples with
    the real kernels, one all-gather of statistics per block -- masks, weights and importance scores of every rank equal the
    single-process run bit for bit (Wanda; DSnoT on ragged text).  `attention`: SDPA, or the reference's own op sequence --
    batched 16-bit `torch.matmul`s (modeling_t5.py:590,638), which run on vlmc_attn_matmul during the replay: ranks form other
    groups than one process does, and the masks still do not depend on the world size."""
    import pruner_helpers as H
    script = tmp_path / "worker.py"
    script.write_text(_WORKER_GLOO.format(root=ROOT, out=str(tmp_path), attention=attention))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
              "--master-port", str(port), str(script)], timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    import toy_models
    toy_models.ToyAttention.use_sdpa = attention   # (restored by the fixture below)
    for tag in ("wanda", "dsnot"):
        single = {k: v.cpu() for k, v in H.run_16bit_toy(tag, "cuda:0", ragged=(tag == "dsnot")).items()}
        for rank in range(world):
            got = torch.load(tmp_path / f"rank{rank}.pt")[tag]
            assert got.keys() == single.keys()
            for k in single:
                assert torch.equal(got[k], single[k]), (rank, tag, k)
