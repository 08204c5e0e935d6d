#!/usr/bin/env python3
"""Final multimodal survival model -- MI355X-native drop-in for the reference's scripts/training/final_multimodal.py.

Same surface: MultiModalSurvivalNet(rna_dim=5005, clinical_dim=1), cox_loss, calculate_cindex,
train_epoch(model, loader, optimizer, device), validate(model, loader, device); same constants as defaults
(BATCH_SIZE 4, LEARNING_RATE 1e-4, NUM_EPOCHS 50, N_FOLDS 5, PATIENCE 15; reference :221-225), same
results/final/cv_results.json schema (:403-417) and models/final/fold_{k}_best.pth checkpoints (:370).
The reference edits constants in place; here they can also be overridden by MMS_* environment variables.
Data: the reference reads data/processed/multimodal_matching_table.csv + a generated dataset module (real TCGA-OV
data, not redistributable); this entry point trains on the seeded synthetic cohort of
multimodal_survival_prediction_amd.data (109 complete patients by default).  Folds are sharded over ranks when
launched with torch.distributed.run (fold k -> rank k mod world).
"""
import os
import time

import numpy as np
import torch

from _common import cv_lockstep, env_dims, env_float, env_int, lockstep_enabled, save_json, setup_device

from multimodal_survival_prediction_amd import data, distributed as D
from multimodal_survival_prediction_amd.losses import calculate_cindex, cox_loss  # noqa: F401  (reference surface)
from multimodal_survival_prediction_amd.models import MultiModalSurvivalNet
from multimodal_survival_prediction_amd.training import FusedOptimizer, ReduceLROnPlateau
from multimodal_survival_prediction_amd.training import train_epoch_final as train_epoch
from multimodal_survival_prediction_amd.training import validate_final as validate

SEED = 42
BATCH_SIZE = env_int("MMS_BATCH_SIZE", 4)
LEARNING_RATE = env_float("MMS_LR", 1e-4)
NUM_EPOCHS = env_int("MMS_EPOCHS", 50)
N_FOLDS = env_int("MMS_FOLDS", 5)
PATIENCE = env_int("MMS_PATIENCE", 15)
N_PATIENTS = env_int("MMS_PATIENTS", 109)


def main():
    torch.manual_seed(SEED)
    np.random.seed(SEED)
    world, rank, device = setup_device()
    cohort = data.cohort_to(data.make_cohort(n=N_PATIENTS, dims=env_dims(), seed=608, complete=True), device)
    folds = data.kfold_indices(cohort["n"], N_FOLDS, seed=SEED)
    os.makedirs("models/final", exist_ok=True)
    local = []
    my_folds = list(D.folds_of_rank(N_FOLDS, world, rank))
    if lockstep_enabled(len(my_folds)):      # all local folds advance together, one launch sequence per batch position
        loaders = [(data.BatchLoader(cohort, folds[f][0], BATCH_SIZE, shuffle=True, seed=SEED + f),
                    data.BatchLoader(cohort, folds[f][1], BATCH_SIZE, shuffle=False)) for f in my_folds]
        models = [MultiModalSurvivalNet().to(device) for _ in my_folds]
        res = cv_lockstep("final", models, loaders, dict(lr=LEARNING_RATE, weight_decay=1e-4, adamw=False), NUM_EPOCHS, PATIENCE,
                          lambda o: ReduceLROnPlateau(o, mode="max", factor=0.5, patience=5),
                          lambda name: f"models/final/fold_{name}_best.pth", device, rank, [f + 1 for f in my_folds])
        local = [{"fold": f + 1, "best_c_index": r["best_c_index"], "patients_per_sec": r["patients_per_sec"], "epochs_run": r["epochs_run"]}
                 for f, r in zip(my_folds, res)]
        my_folds = []
    for fold in my_folds:
        train_idx, val_idx = folds[fold]
        train_loader = data.BatchLoader(cohort, train_idx, BATCH_SIZE, shuffle=True, seed=SEED + fold)
        val_loader = data.BatchLoader(cohort, val_idx, BATCH_SIZE, shuffle=False)
        model = MultiModalSurvivalNet().to(device)
        optimizer = FusedOptimizer(model, lr=LEARNING_RATE, weight_decay=1e-4, adamw=False)      # optim.Adam (:350)
        scheduler = ReduceLROnPlateau(optimizer, mode="max", factor=0.5, patience=5)             # (:351)
        best_c_index, patience_counter, t_train, n_train, epochs_run = 0, 0, 0.0, 0, 0
        for epoch in range(NUM_EPOCHS):
            epochs_run = epoch + 1
            torch.cuda.synchronize(); t0 = time.perf_counter()
            train_loss = train_epoch(model, train_loader, optimizer, device)
            torch.cuda.synchronize(); t_train += time.perf_counter() - t0; n_train += len(train_idx)
            val_loss, val_c_index = validate(model, val_loader, device)
            scheduler.step(val_c_index)
            if (epoch + 1) % 5 == 0 or epoch == 0:
                print(f"[rank {rank}] fold {fold + 1} epoch {epoch + 1:3d}: Train Loss={train_loss:.4f}, "
                      f"Val Loss={val_loss:.4f}, C-index={val_c_index:.4f}", flush=True)
            if val_c_index > best_c_index:
                best_c_index, patience_counter = val_c_index, 0
                torch.save(model.state_dict(), f"models/final/fold_{fold + 1}_best.pth")
            else:
                patience_counter += 1
                if patience_counter >= PATIENCE:
                    break
        local.append({"fold": fold + 1, "best_c_index": best_c_index, "patients_per_sec": n_train / t_train, "epochs_run": epochs_run})
    cv_results = D.gather_fold_results(local, world)
    if rank == 0:
        c = [r["best_c_index"] for r in cv_results]
        save_json("results/final/cv_results.json", {
            "model": "MultiModalSurvivalNet (Late Fusion)", "c_index_mean": float(np.mean(c)), "c_index_std": float(np.std(c)),
            "fold_results": cv_results,
            "hyperparameters": {"batch_size": BATCH_SIZE, "learning_rate": LEARNING_RATE, "epochs": NUM_EPOCHS, "n_folds": N_FOLDS}})
        print(f"C-index: {np.mean(c):.4f} +/- {np.std(c):.4f}; saved results/final/cv_results.json")


if __name__ == "__main__":
    main()
