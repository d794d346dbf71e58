"""RESSA retraining step: the caller of the SparseLoRA kernels.

Mirrors `ImageTextRetrainTask` of lavis/tasks/image_text_retrain.py:21-203 -- `train_step`,
`valid_step` and `_train_inner_loop` with the same signature and the same per-iteration
sequence:

    lr_scheduler.step(cur_epoch, cur_step)
    model.eval();  no_grad + autocast:  _, logits_DD = model(samples, vit_dense=True,  llm_dense=True)   (:152-155)
    model.train(); autocast:            loss, logits_SS = model(samples, vit_dense=False, llm_dense=False) (:160-161)
    kl = KLDivLoss(batchmean, log_target)(log_softmax(SS/T), log_softmax(DD/T))                           (:163)
    loss = (1 - kl_weight) * loss + kl_weight * kl                                                        (:169)
    [scaler.scale(loss)|loss].backward();  every accum_grad_iters: [scaler.]step, zero_grad              (:172-185)

The dense pass runs every LoRA-wrapped linear through the plain `F.linear` branch, the sparse
pass through `vlmc_lora_effective_weight` (forward) and `vlmc_lora_grad` (backward) -- see
lavis/peft/src/peft/tuners/lora.py here.  What is NOT mirrored is the LAVIS runner plumbing around
it (MetricLogger / SmoothedValue, `prepare_sample` from the dataset package, distributed metric
synchronisation): a plain running mean is returned in the reference's `{name: "%.3f"}` format.
"""
from __future__ import annotations

import logging

import torch
import torch.nn.functional as F
from torch.nn import KLDivLoss

from lavis.common.registry import registry


def prepare_sample(samples, cuda_enabled=True):
    """Move the tensors of a sample dict to the GPU (lavis/datasets/data_utils.py `prepare_sample`)."""
    if not cuda_enabled:
        return samples
    return {k: (v.cuda(non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in samples.items()}


@registry.register_task("image_text_retrain")
class ImageTextRetrainTask:
    def __init__(self):
        self.kl_weight = 0.01                                   # :25 (train.py overwrites it from --kl_weight)
        self.T = 2.

    def train_step(self, model, samples, vit_dense=False, llm_dense=False):
        outputs = model(samples, vit_dense=vit_dense, llm_dense=llm_dense)
        return outputs["loss"], outputs["logits"]

    def valid_step(self, model, samples):
        return model(samples, vit_dense=False)["loss"]

    def _train_inner_loop(self, epoch, iters_per_epoch, model, data_loader, optimizer, lr_scheduler, scaler=None,
                          start_iters=None, log_freq=50, cuda_enabled=False, accum_grad_iters=1):
        use_amp = scaler is not None
        if not hasattr(data_loader, "__next__"):
            data_loader = iter(data_loader)
        logging.info("Start training epoch {}, {} iters per inner epoch.".format(epoch, iters_per_epoch))
        inner_epoch = epoch if start_iters is None else start_iters // iters_per_epoch
        kl_fnt = KLDivLoss(reduction="batchmean", log_target=True)
        loss_sum, n_iters, last_lr = 0.0, 0, 0.0
        self.loss_history = []
        for i in range(iters_per_epoch):
            samples = next(data_loader)
            samples = prepare_sample(samples, cuda_enabled=cuda_enabled)
            samples.update({"epoch": inner_epoch, "num_iters_per_epoch": iters_per_epoch, "iters": i})
            lr_scheduler.step(cur_epoch=inner_epoch, cur_step=i)

            model.eval()
            with torch.no_grad():
                with torch.autocast("cuda", enabled=use_amp):
                    _, logits_DD = self.train_step(model=model, samples=samples, vit_dense=True, llm_dense=True)

            model.train()
            with torch.autocast("cuda", enabled=use_amp):
                loss, logits_SS = self.train_step(model=model, samples=samples, vit_dense=False, llm_dense=False)

            kl_loss = kl_fnt(F.log_softmax(logits_SS / self.T, -1), F.log_softmax(logits_DD / self.T, -1))
            loss = (1 - self.kl_weight) * loss + self.kl_weight * kl_loss
            if use_amp:
                scaler.scale(loss).backward()
            else:
                loss.backward()
            if (i + 1) % accum_grad_iters == 0:
                if use_amp:
                    scaler.step(optimizer)
                    scaler.update()
                else:
                    optimizer.step()
                optimizer.zero_grad()
            lv = loss.item()
            self.loss_history.append(lv)
            loss_sum += lv
            n_iters += 1
            last_lr = optimizer.param_groups[0]["lr"]
            if log_freq and i % log_freq == 0:
                logging.info("Train: data epoch: [{}]  [{}/{}]  lr: {:.6f}  loss: {:.4f}".format(epoch, i, iters_per_epoch,
                                                                                                 last_lr, lv))
        stats = {"lr": last_lr, "loss": loss_sum / max(1, n_iters)}
        logging.info("Averaged stats: " + str(stats))
        return {k: "{:.3f}".format(v) for k, v in stats.items()}
