#!/usr/bin/env python3
"""Partial-modality training (608 patients, modality masks, softmax gate) -- MI355X-native drop-in for the
reference's scripts/training/partial_modality_training.py.

Same surface: PartialModalityNet(rna_dim=5005, clinical_dim=1) -> (hazard, gate_weights), cox_loss,
gate_entropy_loss, calculate_cindex, train_epoch -> (avg_cox, avg_entropy), validate -> (avg_loss, c_index); file
defaults BATCH_SIZE 8, LEARNING_RATE 1e-4, NUM_EPOCHS 50, N_FOLDS 3, PATIENCE 15, GATE_ENTROPY_WEIGHT 0.01
(reference :364-369), K-fold over the LABELLED patients with all unlabelled patients added to every training
split (:502-513), results/partial_modality/cv_results.json (:592-607), models/partial_modality/fold_{k}_best.pth.
Override with MMS_* environment variables (BASELINE config 3: MMS_BATCH_SIZE=4 MMS_FOLDS=5 on 4 ranks).
"""
import os
import time

import numpy as np
import torch

from _common import cv_lockstep, env_dims, env_float, env_int, load_or_make_cohort, lockstep_enabled, save_json, setup_device

from multimodal_survival_prediction_amd import data, distributed as D
from multimodal_survival_prediction_amd.losses import calculate_cindex, cox_loss, gate_entropy_loss  # noqa: F401
from multimodal_survival_prediction_amd.models import PartialModalityNet
from multimodal_survival_prediction_amd.training import FusedOptimizer, ReduceLROnPlateau
from multimodal_survival_prediction_amd.training import train_epoch_partial as train_epoch
from multimodal_survival_prediction_amd.training import validate_partial as validate

SEED = 42
BATCH_SIZE = env_int("MMS_BATCH_SIZE", 8)
LEARNING_RATE = env_float("MMS_LR", 1e-4)
NUM_EPOCHS = env_int("MMS_EPOCHS", 50)
N_FOLDS = env_int("MMS_FOLDS", 3)
PATIENCE = env_int("MMS_PATIENCE", 15)
GATE_ENTROPY_WEIGHT = env_float("MMS_GATE_ENTROPY_WEIGHT", 0.01)
N_PATIENTS = env_int("MMS_PATIENTS", 608)


def main():
    torch.manual_seed(SEED)
    np.random.seed(SEED)
    world, rank, device = setup_device()
    cohort = load_or_make_cohort(device, n=N_PATIENTS, dims=env_dims(), seed=608, complete=False)      # data/processed/* in the cwd, else synthetic
    has_surv = cohort["has_survival"].cpu().numpy()
    survival = np.nonzero(has_surv)[0]
    non_survival = np.nonzero(~has_surv)[0]
    folds = data.kfold_indices(len(survival), N_FOLDS, seed=SEED)
    os.makedirs("models/partial_modality", exist_ok=True)
    local = []
    my_folds = list(D.folds_of_rank(N_FOLDS, world, rank))
    if lockstep_enabled(len(my_folds)):
        splits = [(np.concatenate([survival[folds[f][0]], non_survival]), survival[folds[f][1]]) for f in my_folds]
        loaders = [(data.BatchLoader(cohort, tr_all, BATCH_SIZE, shuffle=True, seed=SEED + f),
                    data.BatchLoader(cohort, va_s, BATCH_SIZE, shuffle=False)) for f, (tr_all, va_s) in zip(my_folds, splits)]
        models = [PartialModalityNet().to(device) for _ in my_folds]
        res = cv_lockstep("partial", models, loaders,
                          dict(lr=LEARNING_RATE, weight_decay=1e-4, adamw=False, gate_entropy_weight=GATE_ENTROPY_WEIGHT),
                          NUM_EPOCHS, PATIENCE, lambda o: ReduceLROnPlateau(o, mode="max", factor=0.5, patience=5),
                          lambda name: f"models/partial_modality/fold_{name}_best.pth", device, rank, [f + 1 for f in my_folds])
        local = [{"fold": f + 1, "best_c_index": r["best_c_index"], "train_size": int(len(tr_all)),
                  "train_survival_size": int(len(folds[f][0])), "val_size": int(len(va_s)), "patients_per_sec": r["patients_per_sec"],
                  "epochs_run": r["epochs_run"]}
                 for f, r, (tr_all, va_s) in zip(my_folds, res, splits)]
        my_folds = []
    for fold in my_folds:
        tr, va = folds[fold]
        train_all = np.concatenate([survival[tr], non_survival])          # (:508-513)
        val_survival = survival[va]
        train_loader = data.BatchLoader(cohort, train_all, BATCH_SIZE, shuffle=True, seed=SEED + fold)
        val_loader = data.BatchLoader(cohort, val_survival, BATCH_SIZE, shuffle=False)
        model = PartialModalityNet().to(device)
        optimizer = FusedOptimizer(model, lr=LEARNING_RATE, weight_decay=1e-4, adamw=False,
                                   gate_entropy_weight=GATE_ENTROPY_WEIGHT)
        scheduler = ReduceLROnPlateau(optimizer, mode="max", factor=0.5, patience=5)
        best_c_index, patience_counter, t_train, n_train, epochs_run = 0, 0, 0.0, 0, 0
        for epoch in range(NUM_EPOCHS):
            epochs_run = epoch + 1
            torch.cuda.synchronize(); t0 = time.perf_counter()
            train_cox, train_entropy = train_epoch(model, train_loader, optimizer, device)
            torch.cuda.synchronize(); t_train += time.perf_counter() - t0; n_train += len(train_all)
            val_loss, val_c_index = validate(model, val_loader, device)
            scheduler.step(val_c_index)
            if (epoch + 1) % 5 == 0 or epoch == 0:
                print(f"[rank {rank}] fold {fold + 1} epoch {epoch + 1:3d}: Cox={train_cox:.4f}, Entropy={train_entropy:.4f}, "
                      f"Val Loss={val_loss:.4f}, C-index={val_c_index:.4f}", flush=True)
            if val_c_index > best_c_index:
                best_c_index, patience_counter = val_c_index, 0
                torch.save(model.state_dict(), f"models/partial_modality/fold_{fold + 1}_best.pth")
            else:
                patience_counter += 1
                if patience_counter >= PATIENCE:
                    break
        local.append({"fold": fold + 1, "best_c_index": best_c_index, "train_size": int(len(train_all)),
                      "train_survival_size": int(len(tr)), "val_size": int(len(val_survival)),
                      "patients_per_sec": n_train / t_train, "epochs_run": epochs_run})
    cv_results = D.gather_fold_results(local, world)
    if rank == 0:
        c = [r["best_c_index"] for r in cv_results]
        save_json("results/partial_modality/cv_results.json", {
            "model": "PartialModalityNet (Gating + Entropy Regularization)", "c_index_mean": float(np.mean(c)),
            "c_index_std": float(np.std(c)), "fold_results": cv_results,
            "hyperparameters": {"batch_size": BATCH_SIZE, "learning_rate": LEARNING_RATE, "epochs": NUM_EPOCHS,
                                "n_folds": N_FOLDS, "gate_entropy_weight": GATE_ENTROPY_WEIGHT}})
        print(f"C-index: {np.mean(c):.4f} +/- {np.std(c):.4f}; saved results/partial_modality/cv_results.json")


if __name__ == "__main__":
    main()
