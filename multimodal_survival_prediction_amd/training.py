"""train_epoch / validate of the three reference scripts, driving the fused HIP-graph step.

Signatures are the reference's: train_epoch(model, loader, optimizer, device) / validate(model, loader, device)
(final_multimodal.py:238-305, partial_modality_training.py:382-485, simple_fusion.py:242-333).  `optimizer` may be
  * a `FusedOptimizer` (below): the whole step -- zero-grad, forward, Cox loss, backward, clip_grad_norm_(1.0),
    Adam/AdamW -- is ONE replayed HIP graph per batch, losses accumulate on the device, one host sync per epoch;
  * any torch.optim optimizer: the reference's own loop body runs unchanged on the autograd-compatible path.
Batch-skipping rules, loss averaging and return values follow each script exactly.
"""
import torch

from . import losses
from .engine import engine_of


class FusedOptimizer:
    """Handle for the engine-resident Adam/AdamW state (drop-in where the scripts build `optim.Adam(...)`)."""

    def __init__(self, model, lr=1e-4, weight_decay=1e-4, adamw=False, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0,
                 gate_entropy_weight=0.01):
        self.engine = engine_of(model, lr=lr, weight_decay=weight_decay, adamw=adamw, betas=betas, eps=eps,
                                max_norm=max_norm, gate_entropy_weight=gate_entropy_weight)
        self.param_groups = [dict(lr=lr, weight_decay=weight_decay)]

    def set_lr(self, lr):
        self.param_groups[0]["lr"] = lr
        self.engine.set_lr(lr)

    def zero_grad(self, set_to_none=True):
        pass

    def step(self):
        raise RuntimeError("FusedOptimizer steps inside train_epoch's fused graph; call train_epoch(...)")


class ReduceLROnPlateau:
    """optim.lr_scheduler.ReduceLROnPlateau(mode='max', factor, patience) for FusedOptimizer
    (final_multimodal.py:351: default threshold 1e-4 rel, cooldown 0, min_lr 0)."""

    def __init__(self, optimizer, mode="max", factor=0.5, patience=5, threshold=1e-4):
        assert mode == "max"
        self.opt, self.factor, self.patience, self.threshold = optimizer, factor, patience, threshold
        self.best, self.bad = -float("inf"), 0

    def step(self, metric):
        if metric > self.best * (1 + self.threshold) if self.best > 0 else metric > self.best:
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            self.opt.set_lr(self.opt.param_groups[0]["lr"] * self.factor)
            self.bad = 0


class CosineAnnealingLR:
    """optim.lr_scheduler.CosineAnnealingLR(T_max) (simple_fusion.py:392), eta_min = 0."""

    def __init__(self, optimizer, T_max):
        import math
        self.opt, self.T, self.base, self.t, self.math = optimizer, T_max, optimizer.param_groups[0]["lr"], 0, math

    def step(self):
        self.t += 1
        self.opt.set_lr(self.base * (1 + self.math.cos(self.math.pi * self.t / self.T)) / 2)


def _t(x, dev):
    return x.to(dev, non_blocking=True) if isinstance(x, torch.Tensor) else torch.as_tensor(x).to(dev)


# ---- final_multimodal.py --------------------------------------------------------------------------------
def train_epoch_final(model, loader, optimizer, device):
    if not isinstance(optimizer, FusedOptimizer):
        return _train_epoch_final_autograd(model, loader, optimizer, device)
    model.train()
    eng = optimizer.engine
    eng.reset_epoch_stats()
    for batch in loader:
        label = batch['label']
        eng.train_step(batch['image'], batch['rnaseq'], batch['clinical'], time=label[:, 0], event=label[:, 1],
                       skip_if_unusable=True)      # degenerate batch: loss 0 without graph -> no update (:173-176)
    st = eng.epoch_stats()
    return st["sum_loss"] / st["n_batches"] if st["n_batches"] > 0 else 0


def _train_epoch_final_autograd(model, loader, optimizer, device):
    model.train()
    total, nb = 0.0, 0
    for batch in loader:
        ct, rna, clin = _t(batch['image'], device), _t(batch['rnaseq'], device), _t(batch['clinical'], device)
        label = _t(batch['label'], device)
        hazard = model(ct, rna, clin)
        loss = losses.cox_loss(hazard, label[:, 1], label[:, 0])
        optimizer.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        total += loss.item()
        nb += 1
    return total / nb if nb > 0 else 0


def validate_final(model, loader, device):
    model.eval()
    eng = engine_of(model)
    total, nb, hs, ts, es = 0.0, 0, [], [], []
    for batch in loader:
        label = _t(batch['label'], device)
        hz, _ = eng.forward_eval(batch['image'], batch['rnaseq'], batch['clinical'])
        hz = hz.clone()
        total += losses.cox_loss(hz, label[:, 1], label[:, 0]).item()
        nb += 1
        hs.append(hz); ts.append(label[:, 0]); es.append(label[:, 1])
    c = losses.calculate_cindex(torch.cat(hs), torch.cat(es), torch.cat(ts)) if hs else 0.5
    return (total / nb if nb > 0 else 0), c


# ---- partial_modality_training.py -----------------------------------------------------------------------------
def train_epoch_partial(model, loader, optimizer, device):
    if not isinstance(optimizer, FusedOptimizer):
        raise TypeError("train_epoch_partial drives the fused step; pass a FusedOptimizer")
    model.train()
    eng = optimizer.engine
    eng.reset_epoch_stats()
    for batch in loader:
        label = batch['label']
        valid = torch.as_tensor(batch['has_survival'], dtype=torch.float32)
        eng.train_step(batch['image'], batch['rnaseq'], batch['clinical'], mask=batch['mask'], time=label[:, 0],
                       event=label[:, 1], valid=valid, skip_if_unusable=False)   # entropy term: every batch steps (:418-428)
    st = eng.epoch_stats()
    avg_cox = st["sum_loss"] / st["n_usable"] if st["n_usable"] > 0 else 0
    avg_ent = st["sum_entropy"] / st["n_batches"] if st["n_batches"] > 0 else 0
    return avg_cox, avg_ent


def validate_partial(model, loader, device):
    model.eval()
    eng = engine_of(model)
    total, nb, hs, ts, es = 0.0, 0, [], [], []
    for batch in loader:
        label = _t(batch['label'], device)
        smask = torch.as_tensor(batch['has_survival'], dtype=torch.bool, device=device)
        hz, _ = eng.forward_eval(batch['image'], batch['rnaseq'], batch['clinical'], mask=batch['mask'])
        if int(smask.sum()) > 0:
            h, t, e = hz[smask].clone(), label[smask, 0], label[smask, 1]
            if h.shape[0] >= 2 and float(e.sum()) > 0:
                total += losses.cox_loss(h, e, t).item()
                nb += 1
                hs.append(h); ts.append(t); es.append(e)
    c = losses.calculate_cindex(torch.cat(hs), torch.cat(es), torch.cat(ts)) if hs else 0.5
    return (total / nb if nb > 0 else 0), c


# ---- simple_fusion.py -------------------------------------------------------------------------------------------
def train_epoch_simple(model, loader, optimizer, device):
    if not isinstance(optimizer, FusedOptimizer):
        raise TypeError("train_epoch_simple drives the fused step; pass a FusedOptimizer")
    model.train()
    eng = optimizer.engine
    eng.reset_epoch_stats()
    for batch in loader:
        hs = batch['has_survival']
        valid = torch.as_tensor(hs, dtype=torch.float32)
        if float(valid.sum()) < 2:
            continue                                   # :257-258, before the forward
        eng.train_step(batch['image'], batch['rnaseq'], time=batch['time'].reshape(-1), event=batch['event'].reshape(-1),
                       valid=valid, skip_if_unusable=True)   # no events: forward ran (BN stats moved), no update (:267-268)
    st = eng.epoch_stats()
    return st["sum_loss"] / st["n_usable"] if st["n_usable"] > 0 else 0.0


def validate_simple(model, loader, device):
    model.eval()
    eng = engine_of(model)
    total, nb, hs, ts, es = 0.0, 0, [], [], []
    for batch in loader:
        smask = torch.as_tensor(batch['has_survival'], dtype=torch.bool, device=device)
        if int(smask.sum()) < 2:
            continue
        time, event = _t(batch['time'].reshape(-1), device), _t(batch['event'].reshape(-1), device)
        hz, _ = eng.forward_eval(batch['image'], batch['rnaseq'])
        h, t, e = hz[smask].clone(), time[smask], event[smask].float()
        if float(e.sum()) == 0:
            continue
        total += losses.neg_partial_log_likelihood(h, e, t).item()
        nb += 1
        hs.append(h); ts.append(t); es.append(e)
    if not hs:
        return 0.0, 0.5
    c = losses.ConcordanceIndex()(torch.cat(hs), torch.cat(es), torch.cat(ts)).item()
    return (total / nb if nb > 0 else 0.0), c
