"""train_epoch / validate of the three reference scripts, driving the fused HIP-graph step.

Signatures are the reference's: train_epoch(model, loader, optimizer, device) / validate(model, loader, device)
(final_multimodal.py:238-305, partial_modality_training.py:382-485, simple_fusion.py:242-333).  `optimizer` may be
  * a `FusedOptimizer` (below): the whole step -- zero-grad, forward, Cox loss, backward, clip_grad_norm_(1.0),
    Adam/AdamW -- is ONE replayed HIP graph per batch, losses accumulate on the device, one host sync per epoch;
  * any torch.optim optimizer: the reference's own loop body runs unchanged on the autograd-compatible path.
Batch-skipping rules, loss averaging and return values follow each script exactly.
"""
import torch

from . import losses, ops
from .engine import engine_of


class FusedOptimizer:
    """Handle for the engine-resident Adam/AdamW state (drop-in where the scripts build `optim.Adam(...)`)."""

    def __init__(self, model, lr=1e-4, weight_decay=1e-4, adamw=False, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0,
                 gate_entropy_weight=0.01, cox_ties=None, dn_opts=None):
        # (a model that already belongs to a FoldGroupEngine keeps that engine: this object is then just the handle
        # the LR schedulers talk to -- pass the same hyper-parameters to the group's constructor)
        self.engine = engine_of(model, lr=lr, weight_decay=weight_decay, adamw=adamw, betas=betas, eps=eps,
                                max_norm=max_norm, gate_entropy_weight=gate_entropy_weight, cox_ties=cox_ties, dn_opts=dn_opts)
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
    eng.check_b4()
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
    eng.check_b4()
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
    eng.check_b4()
    return (total / nb if nb > 0 else 0.0), c


# ---- flexible_multimodal.py (rows of SURVEY section 8f: same kernels, learnable missing-modality bias) ---------------------
def train_epoch_flexible(model, loader, optimizer, device):
    """flexible_multimodal.py:262-303: the simple_fusion loop with the (B,2) modality mask as a third model input."""
    if not isinstance(optimizer, FusedOptimizer):
        raise TypeError("train_epoch_flexible drives the fused step; pass a FusedOptimizer")
    model.train()
    eng = optimizer.engine
    eng.reset_epoch_stats()
    for batch in loader:
        valid = torch.as_tensor(batch['has_survival'], dtype=torch.float32)
        if float(valid.sum()) < 2:
            continue                                   # :276-277
        eng.train_step(batch['image'], batch['rnaseq'], mask=batch['mask'][:, :2], time=batch['time'].reshape(-1),
                       event=batch['event'].reshape(-1), valid=valid, skip_if_unusable=True)     # :287-288
    st = eng.epoch_stats()
    return st["sum_loss"] / st["n_usable"] if st["n_usable"] > 0 else 0.0


def validate_flexible(model, loader, device):
    """flexible_multimodal.py:305-357."""
    model.eval()
    eng = engine_of(model)
    total, nb, hs, ts, es = 0.0, 0, [], [], []
    for batch in loader:
        smask = torch.as_tensor(batch['has_survival'], dtype=torch.bool, device=device)
        if int(smask.sum()) < 2:
            continue
        time, event = _t(batch['time'].reshape(-1), device), _t(batch['event'].reshape(-1), device)
        hz, _ = eng.forward_eval(batch['image'], batch['rnaseq'], mask=batch['mask'][:, :2])
        h, t, e = hz[smask].clone(), time[smask], event[smask].float()
        if float(e.sum()) == 0:
            continue
        total += losses.neg_partial_log_likelihood(h, e, t).item()
        nb += 1
        hs.append(h); ts.append(t); es.append(e)
    if not hs:
        return 0.0, 0.5
    c = losses.ConcordanceIndex()(torch.cat(hs), torch.cat(es), torch.cat(ts)).item()
    eng.check_b4()
    return (total / nb if nb > 0 else 0.0), c


# ---- train_rnaseq_only.py ------------------------------------------------------------------------------------------------
def train_epoch_rnaseq(model, loader, optimizer, device):
    """train_rnaseq_only.py:157-176: NPLL on the whole batch, every batch steps, NO gradient clipping (build the
    FusedOptimizer with max_norm=0), mean over len(loader)."""
    if not isinstance(optimizer, FusedOptimizer):
        raise TypeError("train_epoch_rnaseq drives the fused step; pass a FusedOptimizer(model, adamw=True, max_norm=0)")
    model.train()
    eng = optimizer.engine
    eng.reset_epoch_stats()
    for batch in loader:
        eng.train_step(None, batch['rnaseq'], time=batch['time'].reshape(-1), event=batch['event'].reshape(-1),
                       skip_if_unusable=False)       # a batch without events yields loss 0 / zero gradients, and still steps (:168-172)
    st = eng.epoch_stats()
    return st["sum_loss"] / st["n_batches"] if st["n_batches"] > 0 else 0.0


def validate_rnaseq(model, loader, device):
    """train_rnaseq_only.py:178-209."""
    model.eval()
    eng = engine_of(model)
    total, nb, hs, ts, es = 0.0, 0, [], [], []
    for batch in loader:
        time, event = _t(batch['time'].reshape(-1), device), _t(batch['event'].reshape(-1), device).float()
        hz, _ = eng.forward_eval(None, batch['rnaseq'])
        h = hz.clone()
        total += losses.neg_partial_log_likelihood(h, event, time).item()
        nb += 1
        hs.append(h); ts.append(time); es.append(event)
    c = losses.ConcordanceIndex()(torch.cat(hs), torch.cat(es), torch.cat(ts)).item()
    return total / nb, c


# ---- K folds in lock-step (fold groups) ------------------------------------------------------------------------------
# The reference trains its folds one after the other; they are independent, so a FoldGroupEngine advances all of them by
# one batch with ONE launch sequence (fold_group.py).  Per fold, batch order, skipping rules, loss averaging and return
# values are exactly those of train_epoch_<style> / validate_<style> above; only the interleaving of the folds differs.
def _train_kwargs(style, batch):
    """-> (keyword arguments of the fused step, skip_if_unusable) or None when the script skips the batch before the forward."""
    if style == "final":
        label = batch['label']
        return dict(ct=batch['image'], rna=batch['rnaseq'], clinical=batch['clinical'], time=label[:, 0], event=label[:, 1])
    if style == "rnaseq":
        return dict(rna=batch['rnaseq'], time=batch['time'].reshape(-1), event=batch['event'].reshape(-1))
    valid = torch.as_tensor(batch['has_survival'], dtype=torch.float32)
    if style == "partial":
        label = batch['label']
        return dict(ct=batch['image'], rna=batch['rnaseq'], clinical=batch['clinical'], mask=batch['mask'], time=label[:, 0],
                    event=label[:, 1], valid=valid)
    if style in ("simple", "flexible"):
        if float(valid.sum()) < 2:
            return None                                      # simple_fusion.py:257-258, flexible_multimodal.py:276-277
        kw = dict(ct=batch['image'], rna=batch['rnaseq'], time=batch['time'].reshape(-1), event=batch['event'].reshape(-1),
                  valid=valid)
        if style == "flexible":
            kw["mask"] = batch['mask'][:, :2]
        return kw
    raise ValueError(style)


_SKIP_UNUSABLE = {"final": True, "partial": False, "simple": True, "flexible": True, "rnaseq": False}


def _lockstep(loaders, members):
    """Yield, batch position by batch position, {member: batch} of the folds that still have a batch."""
    its = {g: iter(l) for g, l in zip(members, loaders)}
    while its:
        out = {}
        for g in list(its):
            b = next(its[g], None)
            if b is None:
                del its[g]
            else:
                out[g] = b
        if out:
            yield out


def train_epoch_lockstep(group, loaders, style, members=None, concurrent=1):
    """One epoch of the folds `members` (default: all) of a FoldGroupEngine, each on its own loader.
    concurrent = n >= 2: the members are split into n fixed sub-groups that step as n lock-step sub-groups on n HIP streams
    (one sub-group's latency-bound kernels overlap the others').  Every member's loader is iterated under its sub-group's
    stream, so batch tensors are allocated and consumed on the same stream.
    -> per member, what train_epoch_<style> returns for that fold."""
    members = tuple(range(len(group))) if members is None else tuple(members)
    for g in members:
        group.engines[g].model.train()
        group.engines[g].reset_epoch_stats()

    def advance(pos):
        lazy = [(g, b) for g, b in pos.items() if "gather" in b]
        if lazy:        # batches named by index (data.BatchLoader(lazy=True)): one gather launch per (sub-)group step
            by = {}
            for g, b in lazy:
                if style in ("simple", "flexible") and sum(bool(x) for x in b["has_survival"]) < 2:
                    continue                          # skipped before the forward (simple_fusion.py:257-258)
                by.setdefault((len(b["index"]), id(b["gather"])), []).append((g, b))
            for items in by.values():
                group.train_step_indexed(items[0][1]["gather"], torch.stack([torch.as_tensor(b["index"]) for _, b in items]),
                                         members=tuple(g for g, _ in items), skip_if_unusable=_SKIP_UNUSABLE[style])
            pos = {g: b for g, b in pos.items() if "gather" not in b}
        by_size = {}
        for g, batch in pos.items():
            kw = _train_kwargs(style, batch)
            if kw is not None:
                n = int(kw["rna"].shape[0])
                by_size.setdefault(n, []).append((g, kw))
        for items in by_size.values():           # a ragged last batch forms its own (sub-)group step
            group.train_step([kw for _, kw in items], members=tuple(g for g, _ in items),
                             skip_if_unusable=_SKIP_UNUSABLE[style])

    _run_subgroups(group, loaders, members, concurrent, advance)
    if concurrent > 1:
        torch.cuda.synchronize()
    out = []
    for g in members:
        st = group.engines[g].epoch_stats()
        if style in ("final", "rnaseq"):
            out.append(st["sum_loss"] / st["n_batches"] if st["n_batches"] > 0 else 0)
        elif style == "partial":
            out.append((st["sum_loss"] / st["n_usable"] if st["n_usable"] > 0 else 0,
                        st["sum_entropy"] / st["n_batches"] if st["n_batches"] > 0 else 0))
        else:
            out.append(st["sum_loss"] / st["n_usable"] if st["n_usable"] > 0 else 0.0)
    return out


def subgroup_sizes(n_members, concurrent):
    """Sizes of the fixed, contiguous sub-groups n_members lock-step fold models step as on `concurrent` HIP streams: near-equal, at
    most one sub-group per stream, ONE group when there is a single stream or a single member.  Round 3 measurements (K-fold epoch,
    patients/s): 5 members 2 + 2 + 1 on three streams 2740 (3 + 2 on two: 2640); 3 members 1 + 1 + 1: 2136, 2 + 1: 2056, one group of
    3: 1973; 2 members 1 + 1: 1635, one group of 2: 1588 -- what a rank of the K-fold job holds at N = 2 / 4 GPUs.
    MMS_MIN_SPLIT_MEMBERS (default 2): fewest members that are split at all."""
    import os
    min_split = int(os.environ.get("MMS_MIN_SPLIT_MEMBERS", "2"))
    if concurrent <= 1 or n_members < max(min_split, 2):
        return (n_members,)
    nsub = min(int(concurrent), n_members)
    base, extra = divmod(n_members, nsub)
    return tuple(base + (1 if h < extra else 0) for h in range(nsub))


def _run_subgroups(group, loaders, members, concurrent, advance):
    """Lock-step iteration of the members' loaders: as one group, or as the fixed, contiguous sub-groups of subgroup_sizes(), each
    stepping on its own HIP stream (5 folds on 3 streams: 2 + 2 + 1)."""
    sizes = subgroup_sizes(len(members), concurrent)
    if len(sizes) <= 1:
        for pos in _lockstep(loaders, members):
            advance(pos)
        return
    nsub = len(sizes)
    streams = ops.worker_streams(group.device, nsub)
    cuts, o = [], 0
    for n in sizes:
        cuts.append((members[o:o + n], loaders[o:o + n]))
        o += n
    cur = torch.cuda.current_stream()
    for s in streams[:nsub]:
        s.wait_stream(cur)
    its = [_lockstep(ld, mem) for mem, ld in cuts]
    live = [True] * nsub
    while any(live):
        for h in range(nsub):
            if not live[h]:
                continue
            with torch.cuda.stream(streams[h]):
                pos = next(its[h], None)         # the loaders gather their batches on this stream
                if pos is None:
                    live[h] = False
                else:
                    advance(pos)
    for s in streams[:nsub]:
        cur.wait_stream(s)


def _validate_lockstep_named(group, loaders, style, members, concurrent=1):
    """validate_final / validate_partial over lazily NAMED batches (data.BatchLoader(lazy=True), cohort in HBM or pinned host memory):
    per lock-step position one gather launch + one graph (eval-mode forwards, the batches' Cox values, device-side accumulators); the
    hazards stay on the device and the host is synchronised ONCE, at the end (the reference and the eager path sync per batch).
    Bookkeeping as the reference's loops: final -- every batch counts, loss 0 when it has no event (final_multimodal.py:268-305);
    partial -- a batch counts, and its labelled patients enter the C-index, only when >= 2 of them are labelled and one has an event
    (partial_modality_training.py:438-485); simple -- the same rule (simple_fusion.py:314-349 skips batches with < 2 labelled
    patients before the forward and those without an event after it: in eval mode the skipped forward has no side effect), the
    C-index through ConcordanceIndex as the eager path."""
    dev = group.device
    for g in members:
        group.engines[g].model.eval()
        group.engines[g].acc_eval.zero_()
    n_rows = {g: len(ld.idx) for g, ld in zip(members, loaders)}
    n_bat = {g: len(ld) for g, ld in zip(members, loaders)}
    hz = {g: torch.zeros(n_rows[g], device=dev) for g in members}
    flags = {g: torch.zeros(max(n_bat[g], 1), 2, device=dev) for g in members}
    order = {g: [] for g in members}          # (patient indices, labelled flags) of the batches in the order they were evaluated
    off = {g: 0 for g in members}
    def advance(pos):
        by = {}
        for g, b in pos.items():
            by.setdefault((len(b["index"]), id(b["gather"])), []).append((g, b))
        for items in by.values():
            outs = group.eval_loss_step_indexed(items[0][1]["gather"], torch.stack([torch.as_tensor(b["index"]) for _, b in items]),
                                                members=tuple(g for g, _ in items))
            for (g, b), (h, lf) in zip(items, outs):
                n = len(b["index"])
                hz[g][off[g]:off[g] + n].copy_(h)
                flags[g][len(order[g])].copy_(lf)
                order[g].append((torch.as_tensor(b["index"]), torch.as_tensor(b["has_survival"], dtype=torch.bool)))
                off[g] += n

    _run_subgroups(group, loaders, members, concurrent, advance)
    torch.cuda.synchronize()
    for g in members:
        group.engines[g].check_b4()          # the pass's single sync point: a timed-out block-4 hand-off must not yield a C-index
    out = []
    for g, ld in zip(members, loaders):
        a = group.engines[g].acc_eval.tolist()
        if not order[g]:
            out.append((0.0 if style == "simple" else 0, 0.5))
            continue
        lab = ld.c["label"]
        idx = torch.cat([i for i, _ in order[g]])
        if style == "final":
            keep = torch.ones(len(idx), dtype=torch.bool)
            avg = a[0] / a[3] if a[3] > 0 else 0
        else:
            usable = flags[g][:len(order[g]), 1].cpu() > 0
            keep = torch.cat([m & bool(u) for (_, m), u in zip(order[g], usable)])
            avg = a[0] / a[1] if a[1] > 0 else 0
        if not bool(keep.any()):
            out.append((0.0 if style == "simple" else 0, 0.5))
            continue
        sel = idx[keep].to(lab.device)
        H, T, E = hz[g][keep.to(dev)], lab[sel, 0].to(dev), lab[sel, 1].to(dev)
        out.append((avg, losses.ConcordanceIndex()(H, E.float(), T).item() if style == "simple" else losses.calculate_cindex(H, E, T)))
    return out


def validate_lockstep(group, loaders, style, device, members=None, concurrent=1):
    """validate_<style> of the folds `members`, their eval forwards issued as fold-group launches.
    -> per member (val_loss, c_index).  Loaders that NAME their batches (final / partial / simple styles) take _validate_lockstep_named;
    concurrent = n: as train_epoch_lockstep, n sub-groups on n HIP streams (that path only)."""
    members = tuple(range(len(group))) if members is None else tuple(members)
    if style in ("final", "partial", "simple") and all(getattr(ld, "lazy", False) for ld in loaders):
        return _validate_lockstep_named(group, loaders, style, members, concurrent)
    acc = {g: dict(total=0.0, nb=0, hs=[], ts=[], es=[]) for g in members}
    for g in members:
        group.engines[g].model.eval()
    for pos in _lockstep(loaders, members):
        by_size, meta = {}, {}
        for g, batch in pos.items():
            if style == "final":
                kw = dict(ct=batch['image'], rna=batch['rnaseq'], clinical=batch['clinical'])
            elif style == "partial":
                kw = dict(ct=batch['image'], rna=batch['rnaseq'], clinical=batch['clinical'], mask=batch['mask'])
            elif style == "rnaseq":
                kw = dict(rna=batch['rnaseq'])
            else:
                if int(torch.as_tensor(batch['has_survival']).sum()) < 2:
                    continue
                kw = dict(ct=batch['image'], rna=batch['rnaseq'])
                if style == "flexible":
                    kw["mask"] = batch['mask'][:, :2]
            meta[g] = batch
            by_size.setdefault(int(kw["rna"].shape[0]), []).append((g, kw))
        for items in by_size.values():
            outs = group.forward_eval([kw for _, kw in items], members=tuple(g for g, _ in items))
            for (g, _), (hz, _gate) in zip(items, outs):
                batch, a = meta[g], acc[g]
                if style == "final":
                    label = _t(batch['label'], device)
                    h, t, e = hz.clone(), label[:, 0], label[:, 1]
                    a["total"] += losses.cox_loss(h, e, t).item(); a["nb"] += 1
                elif style == "partial":
                    label = _t(batch['label'], device)
                    smask = torch.as_tensor(batch['has_survival'], dtype=torch.bool, device=device)
                    if int(smask.sum()) == 0:
                        continue
                    h, t, e = hz[smask].clone(), label[smask, 0], label[smask, 1]
                    if not (h.shape[0] >= 2 and float(e.sum()) > 0):
                        continue
                    a["total"] += losses.cox_loss(h, e, t).item(); a["nb"] += 1
                elif style == "rnaseq":
                    t, e = _t(batch['time'].reshape(-1), device), _t(batch['event'].reshape(-1), device).float()
                    h = hz.clone()
                    a["total"] += losses.neg_partial_log_likelihood(h, e, t).item(); a["nb"] += 1
                else:
                    smask = torch.as_tensor(batch['has_survival'], dtype=torch.bool, device=device)
                    time, event = _t(batch['time'].reshape(-1), device), _t(batch['event'].reshape(-1), device)
                    h, t, e = hz[smask].clone(), time[smask], event[smask].float()
                    if float(e.sum()) == 0:
                        continue
                    a["total"] += losses.neg_partial_log_likelihood(h, e, t).item(); a["nb"] += 1
                a["hs"].append(h); a["ts"].append(t); a["es"].append(e)
    out = []
    for g in members:
        group.engines[g].check_b4()
    for g in members:
        a = acc[g]
        if not a["hs"]:
            out.append((0.0 if style in ("simple", "flexible", "rnaseq") else 0, 0.5))
            continue
        H, E, T = torch.cat(a["hs"]), torch.cat(a["es"]), torch.cat(a["ts"])
        c = (losses.ConcordanceIndex()(H, E, T).item() if style in ("simple", "flexible", "rnaseq")
             else losses.calculate_cindex(H, E, T))
        out.append((a["total"] / a["nb"] if a["nb"] > 0 else 0, c))
    return out
