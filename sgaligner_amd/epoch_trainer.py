"""Epoch loop around the hot path: the reference's `EpochBasedTrainer` + `Trainer`
(src/engine/base_trainer.py, src/engine/epoch_based_trainer.py:83-216, src/trainers/trainval_sgaligner.py:17-100) with
the same hook names, the same snapshot files and the same optimiser set-up, re-done for one-process-per-GPU on MI355X.

Same as the reference:
  * hooks `before/after_train_epoch`, `before/after_train_step`, `after_backward`, `before/after_val_epoch`,
    `before/after_val_step`, `train_step`, `val_step`, `set_train_mode`, `set_eval_mode`, `run()`;
  * Adam over model parameters (+ both CustomMultiLossLayer log_vars when more than one module), lr / weight_decay
    from `cfg.optim` (trainval_sgaligner.py:49-55);
  * snapshots: `epoch-<n>.pth.tar` = {'epoch','iteration','model'}, `snapshot.pth.tar` = that + 'optimizer'
    (+ 'scheduler'), `best_snapshot.pth.tar` when the validation loss improves (base_trainer.py:80-101,
    epoch_based_trainer.py:171-174); `load_snapshot` is non-strict and reports missing / unexpected keys (:103-138),
    so reference checkpoints load here and ours load there.

Different on purpose (the hygiene list of SURVEY.md 8(f) rank 3):
  * no `retain_graph=True`, no per-iteration `torch.cuda.empty_cache()`, no per-iteration `.item()` syncs: running loss
    sums stay on the device and are read back once per `log_steps`;
  * batches go to the GPU through `datasets.DevicePrefetcher` (pinned, non-blocking, one batch ahead on a second HIP stream:
    the upload of batch i+1 overlaps the step of batch i), index sets stay on the host;
  * the two `log_vars` vectors are saved too (key 'loss_layers'; the reference loses them on resume);
  * multi-GPU is real: under torchrun every rank owns a contiguous shard of each batch's pairs, the loss is the
    batch-GLOBAL loss (AlignerSteps._global_loss over RCCL) and parameter gradients are all-reduced;
  * gradient accumulation (`grad_acc_steps`) is honoured exactly as `optimizer_step` does (base_trainer.py:186-191).
"""
from __future__ import annotations

import logging
import os
import os.path as osp
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from . import dist as sdist
from . import ops
from .datasets import DeviceBatch, DevicePrefetcher
from .trainer import AlignerSteps


class EpochBasedTrainer:
    def __init__(self, steps: AlignerSteps, output_dir: str, max_epoch: int, lr: float = 1e-3, weight_decay: float = 0.0,
                 log_steps: int = 10, grad_acc_steps: int = 1, logger: logging.Logger | None = None, run_grad_check: bool = False):
        self.steps = steps
        self.model = steps.model
        self.loss_func = steps.loss_func
        self.device = steps.device
        self.max_epoch = max_epoch
        self.log_steps = log_steps
        self.grad_acc_steps = grad_acc_steps
        self.run_grad_check = run_grad_check          # epoch_based_trainer.py:65-73 (default off, as in the reference)
        self.snapshot_dir = osp.join(output_dir, 'snapshots')
        self.distributed = dist.is_initialized() and dist.get_world_size() > 1
        self.rank = dist.get_rank() if self.distributed else 0
        self.world = dist.get_world_size() if self.distributed else 1
        if self.rank == 0:
            os.makedirs(self.snapshot_dir, exist_ok=True)
        self.logger = logger or logging.getLogger('sgaligner_amd')
        # torch's fused Adam: one multi-tensor launch per step instead of the per-operation foreach chain (~3 ms of host time per step
        # at the reference's batch sizes, more than the whole forward+backward); same update rule (trainval_sgaligner.py:47-53)
        on_gpu = all(p.is_cuda for p in steps.params)
        try:
            self.optimizer = torch.optim.Adam([{'params': steps.params}], lr=lr, weight_decay=weight_decay, fused=on_gpu)
        except (TypeError, RuntimeError):
            self.optimizer = torch.optim.Adam([{'params': steps.params}], lr=lr, weight_decay=weight_decay)
        self.scheduler = None
        self.epoch = 0
        self.iteration = 0
        self.inner_iteration = 0
        self.best_val_loss = float('inf')
        self.training = True
        self.train_loader = None
        self.val_loader = None
        self.history = []                      # one dict per epoch: train/val means (host floats)

    # ---- registration (base_trainer.py:140-176) ------------------------------------------------------
    def register_loader(self, train_loader, val_loader):
        self.train_loader, self.val_loader = train_loader, val_loader

    def register_scheduler(self, scheduler):
        self.scheduler = scheduler

    # ---- hooks: no-ops by default, same names / arguments as the reference ---------------------------
    def before_train_epoch(self, epoch): pass
    def before_val_epoch(self, epoch): pass
    def after_train_epoch(self, epoch): pass
    def after_val_epoch(self, epoch): pass
    def before_train_step(self, epoch, iteration, data_dict): pass
    def before_val_step(self, epoch, iteration, data_dict): pass
    def after_train_step(self, epoch, iteration, data_dict, output_dict, result_dict): pass
    def after_val_step(self, epoch, iteration, data_dict, output_dict, result_dict): pass
    def after_backward(self, epoch, iteration, data_dict, output_dict, result_dict): pass

    def train_step(self, epoch, iteration, data_dict):
        return self.steps.train_step(epoch, iteration, data_dict)

    def val_step(self, epoch, iteration, data_dict):
        return self.steps.val_step(epoch, iteration, data_dict)

    def set_train_mode(self):
        self.training = True
        self.model.train()
        self.steps.multi_loss_layer_ial.train()
        self.steps.multi_loss_layer_icl.train()
        torch.set_grad_enabled(True)

    def set_eval_mode(self):
        self.training = False
        self.model.eval()
        self.steps.multi_loss_layer_ial.eval()
        self.steps.multi_loss_layer_icl.eval()
        torch.set_grad_enabled(False)

    def get_lr(self):
        return self.optimizer.param_groups[0]['lr']

    # ---- snapshots -------------------------------------------------------------------------------------
    def _loss_layer_state(self):
        return {'ial': self.steps.multi_loss_layer_ial.state_dict(), 'icl': self.steps.multi_loss_layer_icl.state_dict()}

    def save_snapshot(self, filename):
        if self.rank != 0:
            return
        state = {'epoch': self.epoch, 'iteration': self.iteration,
                 'model': OrderedDict(self.model.state_dict()), 'loss_layers': self._loss_layer_state()}
        torch.save(state, osp.join(self.snapshot_dir, filename))
        state = dict(state)
        state['optimizer'] = self.optimizer.state_dict()
        if self.scheduler is not None:
            state['scheduler'] = self.scheduler.state_dict()
        state['best_val_loss'] = self.best_val_loss
        torch.save(state, osp.join(self.snapshot_dir, 'snapshot.pth.tar'))
        self.logger.info('Snapshot saved to "%s" (+ snapshot.pth.tar)', osp.join(self.snapshot_dir, filename))

    def load_snapshot(self, snapshot):
        state = torch.load(snapshot, map_location='cpu', weights_only=False)
        model_dict = state['model']
        model_dict = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in model_dict.items())
        self.model.load_state_dict(model_dict, strict=False)
        have, want = set(model_dict), set(self.model.state_dict())
        if want - have:
            self.logger.warning('Missing keys: %s', sorted(want - have))
        if have - want:
            self.logger.warning('Unexpected keys: %s', sorted(have - want))
        if 'loss_layers' in state:
            self.steps.multi_loss_layer_ial.load_state_dict(state['loss_layers']['ial'])
            self.steps.multi_loss_layer_icl.load_state_dict(state['loss_layers']['icl'])
        self.epoch = state.get('epoch', self.epoch)
        self.iteration = state.get('iteration', self.iteration)
        self.best_val_loss = state.get('best_val_loss', self.best_val_loss)
        if 'optimizer' in state:
            self.optimizer.load_state_dict(state['optimizer'])
        if 'scheduler' in state and self.scheduler is not None:
            self.scheduler.load_state_dict(state['scheduler'])
        return {'missing': sorted(want - have), 'unexpected': sorted(have - want)}

    # ---- one process per GPU: this rank's pairs of a collated batch ----------------------------------------
    def _shard(self, data_dict):
        if self.distributed:
            lo, hi = sdist.shard_range(int(data_dict['batch_size']), self.rank, self.world)
            data_dict = sdist.shard_data_dict(data_dict, lo, hi)
        return data_dict

    def _to_device(self, data_dict):
        return DeviceBatch(self._shard(data_dict), self.device)

    def _device_batches(self, loader):
        """The loader's batches, sharded for this rank and uploaded one batch ahead on a second HIP stream."""
        return DevicePrefetcher(loader, self.device, prepare=self._shard)

    def check_gradients(self, epoch, iteration, data_dict, output_dict, result_dict):
        """NaN / Inf scan of every gradient (one device-side reduction; the reference dumps data/model and drops into ipdb,
        epoch_based_trainer.py:65-73 -- here: dump and raise)."""
        if not self.run_grad_check:
            return
        bad = torch.zeros((), device=self.device)
        for p in self.steps.params:
            if p.grad is not None:
                bad += (~torch.isfinite(p.grad)).sum()
        if float(bad) > 0:
            if self.rank == 0:
                torch.save({k: v for k, v in data_dict.items()}, osp.join(self.snapshot_dir, 'bad_grad_data.pth'))
                torch.save(self.model.state_dict(), osp.join(self.snapshot_dir, 'bad_grad_model.pth'))
            raise FloatingPointError(f'Epoch {epoch}, iter {iteration}: invalid gradients ({int(bad)} non-finite values); '
                                     f'data_dict and model saved under {self.snapshot_dir}')

    def _optimizer_step(self, iteration):
        if iteration % self.grad_acc_steps == 0:
            self.optimizer.step()
            self.optimizer.zero_grad(set_to_none=True)

    # ---- loops -----------------------------------------------------------------------------------------------
    def train_epoch(self):
        if self.distributed and hasattr(getattr(self.train_loader, 'sampler', None), 'set_epoch'):
            self.train_loader.sampler.set_epoch(self.epoch)
        if getattr(self, '_shuffle_gen', None) is not None:      # the epoch's shuffle depends on (seed, epoch) only: resumable
            self._shuffle_gen.manual_seed(self._shuffle_seed + self.epoch)
        self.before_train_epoch(self.epoch)
        self.optimizer.zero_grad(set_to_none=True)
        total = len(self.train_loader)
        keys, acc, n_acc = None, None, 0
        ep_sum, ep_n = None, 0
        t0 = time.time()
        for iteration, data_dict in enumerate(self._device_batches(self.train_loader)):
            self.inner_iteration = iteration + 1
            self.iteration += 1
            self.before_train_step(self.epoch, self.inner_iteration, data_dict)
            output_dict, result_dict = self.train_step(self.epoch, self.inner_iteration, data_dict)
            result_dict['loss'].backward()
            if self.distributed and self.inner_iteration % self.grad_acc_steps == 0:
                # Reduce ONCE per optimiser step, on the locally accumulated gradients of all its micro-steps (reducing
                # every micro-step in place would re-sum the already-reduced part: world*G1 + G2).  log_vars see the
                # full (replicated) loss on every rank, everything else only this rank's rows -> pre-divide log_vars.
                self.steps.reduce_grads()
            self.after_backward(self.epoch, self.inner_iteration, data_dict, output_dict, result_dict)
            self.check_gradients(self.epoch, self.inner_iteration, data_dict, output_dict, result_dict)
            self._optimizer_step(self.inner_iteration)
            self.after_train_step(self.epoch, self.inner_iteration, data_dict, output_dict, result_dict)
            # running sums stay on the device; one host read-back per log_steps
            if keys is None:
                keys = [k for k, v in result_dict.items() if isinstance(v, torch.Tensor) and v.numel() == 1]
                acc = torch.zeros(len(keys), device=self.device, dtype=torch.float64)
                ep_sum = torch.zeros_like(acc)
            vals = torch.stack([result_dict[k].detach().double().reshape(()) for k in keys])
            acc += vals
            ep_sum += vals
            n_acc += 1
            ep_n += 1
            if self.inner_iteration % self.log_steps == 0 or self.inner_iteration == total:
                means = (acc / n_acc).tolist()
                if self.rank == 0:
                    self.logger.info('Epoch %d/%d iter %d/%d lr %.3e  %s  (%.2f it/s)', self.epoch, self.max_epoch,
                                     self.inner_iteration, total, self.get_lr(),
                                     ' '.join(f'{k}: {m:.4f}' for k, m in zip(keys, means)),
                                     self.inner_iteration / max(1e-9, time.time() - t0))
                acc.zero_()
                n_acc = 0
        ops.DEFERRED_CHECKS.flush()                      # deferred range checks of the epoch's last batches
        self.after_train_epoch(self.epoch)
        if self.scheduler is not None:
            self.scheduler.step()
        summary = dict(zip(keys, (ep_sum / max(1, ep_n)).tolist())) if keys else {}
        self.save_snapshot(f'epoch-{self.epoch}.pth.tar')
        return summary

    def inference_epoch(self):
        self.set_eval_mode()
        self.before_val_epoch(self.epoch)
        keys, acc, n = None, None, 0
        last_loss = None
        for iteration, data_dict in enumerate(self._device_batches(self.val_loader)):
            self.inner_iteration = iteration + 1
            self.before_val_step(self.epoch, self.inner_iteration, data_dict)
            output_dict, result_dict = self.val_step(self.epoch, self.inner_iteration, data_dict)
            self.after_val_step(self.epoch, self.inner_iteration, data_dict, output_dict, result_dict)
            if keys is None:
                keys = [k for k, v in result_dict.items() if isinstance(v, torch.Tensor) and v.numel() == 1]
                acc = torch.zeros(len(keys), device=self.device, dtype=torch.float64)
            acc += torch.stack([result_dict[k].detach().double().reshape(()) for k in keys])
            last_loss = result_dict['loss'].detach()
            n += 1
        summary = dict(zip(keys, (acc / max(1, n)).tolist())) if keys else {}
        # the reference compares the LAST batch's loss with the best so far (epoch_based_trainer.py:171-174)
        if last_loss is not None and float(last_loss) < self.best_val_loss:
            self.best_val_loss = float(last_loss)
            self.save_snapshot('best_snapshot.pth.tar')
        if self.rank == 0:
            self.logger.info('[Val] Epoch %d  %s', self.epoch, ' '.join(f'{k}: {v:.4f}' for k, v in summary.items()))
        self.after_val_epoch(self.epoch)
        self.set_train_mode()
        return summary

    def run(self, resume: bool = False, snapshot: str | None = None):
        assert self.train_loader is not None and self.val_loader is not None
        if resume:
            self.load_snapshot(osp.join(self.snapshot_dir, 'snapshot.pth.tar'))
        elif snapshot is not None:
            self.load_snapshot(snapshot)
        self.set_train_mode()
        while self.epoch < self.max_epoch:
            self.epoch += 1
            tr = self.train_epoch()
            va = self.inference_epoch()
            self.history.append({'epoch': self.epoch, 'train': tr, 'val': va})
        return self.history


class Trainer(EpochBasedTrainer):
    """`trainers/trainval_sgaligner.py:Trainer` from the same cfg fields: cfg.modules, cfg.model.rel_dim/attr_dim,
    cfg.loss.zoom, cfg.optim.lr/weight_decay/max_epoch, cfg.train/val.batch_size, cfg.num_workers, cfg.output_dir
    (+ everything datasets.Scan3RDataset reads)."""

    def __init__(self, cfg, log_steps=10):
        from .datasets import Scan3RDataset
        steps = AlignerSteps(cfg.modules, rel_dim=cfg.model.rel_dim, attr_dim=cfg.model.attr_dim, zoom=cfg.loss.zoom,
                             device='cuda', seed=getattr(cfg, 'seed', 42))
        super().__init__(steps, cfg.output_dir, cfg.optim.max_epoch, lr=cfg.optim.lr, weight_decay=cfg.optim.weight_decay,
                         log_steps=log_steps, grad_acc_steps=getattr(cfg.optim, 'grad_acc_steps', 1))
        train_ds, val_ds = Scan3RDataset(cfg, 'train'), Scan3RDataset(cfg, 'val')
        nw = getattr(cfg, 'num_workers', 0)
        # every rank iterates the SAME global batches (same shuffle seed) and keeps its shard of the pairs of each
        g = torch.Generator().manual_seed(getattr(cfg, 'seed', 42))
        self._shuffle_gen, self._shuffle_seed = g, getattr(cfg, 'seed', 42)
        self.register_loader(
            torch.utils.data.DataLoader(train_ds, batch_size=cfg.train.batch_size, shuffle=True, generator=g,
                                        num_workers=nw, collate_fn=train_ds.collate_fn, drop_last=True,
                                        persistent_workers=nw > 0),        # workers (and their scan caches) live across epochs
            torch.utils.data.DataLoader(val_ds, batch_size=cfg.val.batch_size, shuffle=False, num_workers=nw,
                                        collate_fn=val_ds.collate_fn, drop_last=False, persistent_workers=nw > 0))
