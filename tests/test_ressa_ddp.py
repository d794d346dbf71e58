"""RESSA under data parallelism (BASELINE.json config 5: DP with gradient all-reduce): the drop-in SparseLoRA layer and
`ImageTextRetrainTask._train_inner_loop` inside `DistributedDataParallel`, world_size 2.

CPU (gloo, oracle stand-ins for the kernels): adapters stay identical on both ranks, and equal a single process that
averages the two ranks' gradients by hand.  GPU (gloo, both ranks on cuda:0, the real kernels): the custom autograd function
feeds DDP's gradient hooks -- adapters identical across ranks and different from purely local training."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _paths():
    for p in (ROOT, os.path.join(ROOT, "vlm-compression_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)


def _build(device, oracle):
    import pruner_helpers as H
    import toy_models
    from lavis.peft.src.peft.tuners.lora import mark_only_lora_as_trainable
    if oracle:
        import oracle_ops
        from oracle import sparse_lora as OL
        from vlmc import sparse_lora

        class MP:
            @staticmethod
            def setattr(obj, name, val):
                setattr(obj, name, val)
        oracle_ops.install(MP)
        sparse_lora.linear = lambda x, w, A, B, mask, bias, scaling, sparse: OL.forward(x, w, A, B, mask, bias, scaling, sparse)
    model = toy_models.init_toy(toy_models.ToyBlipT5(), seed=7)
    H.wrap_lora(model)
    g = torch.Generator().manual_seed(99)
    for m in model.modules():
        if hasattr(m, "lora_A"):
            m.merge_weights = False
            m.sparse = True
            m.mask = torch.rand(m.weight.shape, generator=g) > 0.5
    model.to(device)
    mark_only_lora_as_trainable(model)
    return model


def _train(model, batches, steps):
    from lavis.tasks.image_text_retrain import ImageTextRetrainTask
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.05)

    class Sched:
        def step(self, cur_epoch, cur_step):
            pass
    task = ImageTextRetrainTask()
    task.kl_weight = 0.1
    task._train_inner_loop(epoch=0, iters_per_epoch=steps, model=model, data_loader=iter(batches), optimizer=opt,
                           lr_scheduler=Sched(), scaler=None, log_freq=0, cuda_enabled=False)
    return task


def _worker(rank, world, port, out_dir, device, oracle):
    _paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import toy_models
    from torch.nn.parallel import DistributedDataParallel as DDP
    model = _build(device, oracle)
    ddp = DDP(model)                     # runner_base.py:104-108 wraps the model the same way
    batches = [{k: t.to(device) for k, t in b.items()} for b in toy_models.make_batches(6, seed=11)]
    _train(ddp, batches[rank::world], steps=3)            # each rank sees its own shard
    torch.save({k: v.cpu() for k, v in model.state_dict().items() if "lora_" in k}, os.path.join(out_dir, f"lora_{rank}.pt"))
    dist.destroy_process_group()


def _check(tmp_path, device, oracle):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), device, oracle), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "lora_0.pt"), torch.load(tmp_path / "lora_1.pt")
    assert a.keys() == b.keys() and len(a) > 20
    for k in a:
        assert torch.equal(a[k], b[k]), k                # DDP keeps the replicas in lock-step
    return a


@pytest.mark.timeout(600)
def test_ressa_ddp_world2_gloo_cpu(tmp_path):
    got = _check(tmp_path, "cpu", True)
    # single process, gradients of the two shards averaged by hand = what the all-reduce computes
    _paths()
    import toy_models
    from lavis.tasks.image_text_retrain import ImageTextRetrainTask  # noqa: F401
    from torch.nn import KLDivLoss
    import torch.nn.functional as F
    model = _build("cpu", True)
    batches = toy_models.make_batches(6, seed=11)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.05)
    kl = KLDivLoss(reduction="batchmean", log_target=True)
    for step in range(3):
        grads = None
        for rank in range(2):
            s = batches[rank::2][step]
            model.eval()
            with torch.no_grad():
                dd = model(s, vit_dense=True, llm_dense=True)["logits"]
            model.train()
            out = model(s, vit_dense=False, llm_dense=False)
            loss = 0.9 * out["loss"] + 0.1 * kl(F.log_softmax(out["logits"] / 2.0, -1), F.log_softmax(dd / 2.0, -1))
            g = torch.autograd.grad(loss, params)
            grads = [x / 2 for x in g] if grads is None else [a + x / 2 for a, x in zip(grads, g)]
        for p, g in zip(params, grads):
            p.grad = g
        opt.step()
        opt.zero_grad()
    ref = {k: v for k, v in model.state_dict().items() if "lora_" in k}
    for k in got:
        assert torch.allclose(got[k], ref[k], rtol=1e-5, atol=1e-7), k


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_ressa_ddp_world2_on_one_gpu_with_the_kernels(tmp_path):
    got = _check(tmp_path, "cuda:0", False)
    _paths()
    import toy_models
    model = _build("cuda:0", False)
    batches = [{k: t.to("cuda:0") for k, t in b.items()} for b in toy_models.make_batches(6, seed=11)]
    _train(model, batches[0::2], steps=3)                 # rank 0's shard without the all-reduce
    local = {k: v.cpu() for k, v in model.state_dict().items() if "lora_" in k}
    assert any(not torch.equal(local[k], got[k]) for k in got)
