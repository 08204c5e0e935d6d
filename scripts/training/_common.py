"""Shared driver pieces of the three entry points (synthetic cohort, K-fold loop, result JSON)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def env_int(name, default):
    return int(os.environ.get(name, default))


def env_float(name, default):
    return float(os.environ.get(name, default))


def env_dims(default=(64, 64, 32)):
    """MMS_VOLUME="D,H,W": CT volume size of the synthetic cohort (default: the reference's target_size 64,64,32)."""
    v = os.environ.get("MMS_VOLUME")
    return tuple(int(x) for x in v.split(",")) if v else tuple(default)


def setup_device():
    from multimodal_survival_prediction_amd import distributed as D
    world, rank, local = D.init()
    if not torch.cuda.is_available():
        raise SystemExit("these entry points run the HIP hot path and need an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    return world, rank, torch.device("cuda", local)


def save_json(path, obj):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(obj, f, indent=2)


def load_or_make_cohort(device, **make_kw):
    """The reference reads data/processed/{full_matching_table.csv, rnaseq_normalized_mapped.csv} relative to the cwd.  If that
    contract is present (e.g. written by cohort_io.write_cohort, or real data converted to .npy volumes) it is loaded ONCE
    into an HBM-resident store with the volume preprocessing done on the GPU; otherwise a seeded synthetic cohort is built."""
    from multimodal_survival_prediction_amd import cohort_io, data
    root = os.environ.get("MMS_DATA_ROOT", ".")
    if os.path.exists(os.path.join(root, cohort_io.TABLE)):
        print(f"loading cohort from {os.path.join(root, cohort_io.TABLE)}", flush=True)
        return cohort_io.load_cohort(root, device, target_size=make_kw.get("dims", (64, 64, 32)))
    return data.cohort_to(data.make_cohort(**make_kw), device)


def data_mod():
    from multimodal_survival_prediction_amd import data
    return data


def lockstep_enabled(n_local_folds):
    """Folds of this rank train in lock-step as one fold group (MMS_LOCKSTEP=0 restores fold-after-fold order)."""
    return env_int("MMS_LOCKSTEP", 1) != 0 and 2 <= n_local_folds <= 10


def cv_lockstep(style, models, loaders, group_kw, num_epochs, patience, make_scheduler, ckpt_path, device, rank, fold_names,
                log_every=5):
    """The per-fold epoch loop of the three scripts (train, validate, scheduler, best-checkpoint, early stopping) run for
    all local folds at once: one FoldGroupEngine advances every still-active fold by one batch per launch sequence.
    loaders: [(train_loader, val_loader)] per fold; make_scheduler(optimizer) -> object with step(metric) or step();
    patience None = no early stopping (simple_fusion.py).  -> per fold dict(best_c_index, best_epoch, patients_per_sec)."""
    import inspect
    import time
    from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
    from multimodal_survival_prediction_amd.training import FusedOptimizer, train_epoch_lockstep, validate_lockstep
    group = FoldGroupEngine(models, **group_kw)
    for tl, vl in loaders:                     # cohort in HBM (or pinned host memory): name the batches, let the group gather them
        for ld in ((tl, vl) if style in ("final", "partial", "simple") else (tl,)):      # (validate_lockstep's named-batch path: final / partial / simple)
            if env_int("MMS_LAZY_BATCHES", 1) and (ld.c["image"].is_cuda or ld.c["image"].is_pinned()):
                ld.lazy = True
                ld.view = data_mod().gather_view(ld.c, with_valid=(style != "final"))
                ld.hs_cpu = ld.c["has_survival"].cpu().tolist()
    opts = [FusedOptimizer(m, lr=group_kw.get("lr", 1e-4), weight_decay=group_kw.get("weight_decay", 1e-4)) for m in models]
    scheds = [make_scheduler(o) for o in opts]
    takes_metric = [len(inspect.signature(s.step).parameters) > 0 for s in scheds]
    st = [dict(best=0.0, best_epoch=0, bad=0, done=False, t=0.0, n=0, epochs=0) for _ in models]
    for epoch in range(1, num_epochs + 1):
        active = [g for g in range(len(models)) if not st[g]["done"]]
        if not active:
            break
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr = train_epoch_lockstep(group, [loaders[g][0] for g in active], style, members=active,
                                  concurrent=env_int("MMS_LOCKSTEP_STREAMS", 3))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        n_ep = sum(len(loaders[g][0].idx) for g in active)
        va = validate_lockstep(group, [loaders[g][1] for g in active], style, device, members=active,
                               concurrent=env_int("MMS_VALIDATE_STREAMS", 2))
        for g, trg, (val_loss, c) in zip(active, tr, va):
            s = st[g]
            s["t"] += dt; s["n"] += n_ep                       # the group's aggregate rate while this fold was active
            s["epochs"] = epoch
            scheds[g].step(c) if takes_metric[g] else scheds[g].step()
            if c > s["best"]:
                s["best"], s["best_epoch"], s["bad"] = c, epoch, 0
                torch.save(models[g].state_dict(), ckpt_path(fold_names[g]))
            else:
                s["bad"] += 1
                if patience is not None and s["bad"] >= patience:
                    s["done"] = True
            if epoch % log_every == 0 or epoch == 1:
                print(f"[rank {rank}] fold {fold_names[g]} epoch {epoch:3d}: train={trg} val_loss={val_loss:.4f} "
                      f"C-index={c:.4f} best={s['best']:.4f}", flush=True)
    return [dict(best_c_index=s["best"], best_epoch=s["best_epoch"], epochs_run=s["epochs"], patients_per_sec=s["n"] / max(s["t"], 1e-9))
            for s in st]
