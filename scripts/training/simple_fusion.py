#!/usr/bin/env python3
"""Simple late fusion (RNA-seq + image) -- MI355X-native drop-in for the reference's scripts/training/simple_fusion.py.

Same surface: SimpleFusionModel(rna_dim=5005, img_feature_dim=128, rna_feature_dim=256), neg_partial_log_likelihood,
ConcordanceIndex, train_epoch / validate with the reference's two batch-skipping rules (:257-258 before the forward,
:267-268 after it); file defaults N_FOLDS 3, NUM_EPOCHS 50, BATCH_SIZE 8, LEARNING_RATE 1e-4, WEIGHT_DECAY 1e-3
(:86-91), AdamW + CosineAnnealingLR(T_max=NUM_EPOCHS) (:391-392), results/simple_fusion/cv_results.json (:444-451)
and results/simple_fusion/best_model_fold{k}.pth (:406-407).  BASELINE config 1: MMS_PATIENTS=88 MMS_BATCH_SIZE=4
MMS_FOLDS=... (the CT encoder always runs here; config 1's "CT encoder stubbed" CPU case is the oracle's).
"""
import os
import time

import numpy as np
import torch

from _common import cv_lockstep, env_dims, env_float, env_int, lockstep_enabled, save_json, setup_device

from multimodal_survival_prediction_amd import data, distributed as D
from multimodal_survival_prediction_amd.losses import ConcordanceIndex, neg_partial_log_likelihood  # noqa: F401
from multimodal_survival_prediction_amd.models import SimpleFusionModel
from multimodal_survival_prediction_amd.training import CosineAnnealingLR, FusedOptimizer
from multimodal_survival_prediction_amd.training import train_epoch_simple as train_epoch
from multimodal_survival_prediction_amd.training import validate_simple as validate

RESULTS_DIR = "results/simple_fusion"
N_FOLDS = env_int("MMS_FOLDS", 3)
NUM_EPOCHS = env_int("MMS_EPOCHS", 50)
BATCH_SIZE = env_int("MMS_BATCH_SIZE", 8)
LEARNING_RATE = env_float("MMS_LR", 1e-4)
WEIGHT_DECAY = env_float("MMS_WEIGHT_DECAY", 1e-3)
N_PATIENTS = env_int("MMS_PATIENTS", 88)


def main():
    world, rank, device = setup_device()
    os.makedirs(RESULTS_DIR, exist_ok=True)
    cohort = data.cohort_to(data.make_cohort(n=N_PATIENTS, dims=env_dims(), seed=88, complete=True), device)
    folds = data.kfold_indices(cohort["n"], N_FOLDS, seed=42)
    local = []
    my_folds = list(D.folds_of_rank(N_FOLDS, world, rank))
    if lockstep_enabled(len(my_folds)):
        loaders = [(data.BatchLoader(cohort, folds[f][0], BATCH_SIZE, shuffle=True, seed=f + 1, style="simple"),
                    data.BatchLoader(cohort, folds[f][1], BATCH_SIZE, shuffle=False, style="simple")) for f in my_folds]
        models = [SimpleFusionModel(rna_dim=cohort["rnaseq"].shape[1]).to(device) for _ in my_folds]
        res = cv_lockstep("simple", models, loaders, dict(lr=LEARNING_RATE, weight_decay=WEIGHT_DECAY, adamw=True), NUM_EPOCHS, None,
                          lambda o: CosineAnnealingLR(o, T_max=NUM_EPOCHS),
                          lambda name: os.path.join(RESULTS_DIR, f"best_model_fold{name}.pth"), device, rank,
                          [f + 1 for f in my_folds], log_every=10)
        local = [{"fold": f + 1, "best_c_index": r["best_c_index"], "best_epoch": r["best_epoch"], "train_size": int(len(folds[f][0])),
                  "val_size": int(len(folds[f][1])), "patients_per_sec": r["patients_per_sec"]} for f, r in zip(my_folds, res)]
        my_folds = []
    for fold0 in my_folds:
        fold = fold0 + 1
        train_ids, val_ids = folds[fold0]
        train_loader = data.BatchLoader(cohort, train_ids, BATCH_SIZE, shuffle=True, seed=fold, style="simple")
        val_loader = data.BatchLoader(cohort, val_ids, BATCH_SIZE, shuffle=False, style="simple")
        model = SimpleFusionModel(rna_dim=cohort["rnaseq"].shape[1]).to(device)
        optimizer = FusedOptimizer(model, lr=LEARNING_RATE, weight_decay=WEIGHT_DECAY, adamw=True)
        scheduler = CosineAnnealingLR(optimizer, T_max=NUM_EPOCHS)
        best_c_index, best_epoch, t_train, n_train = 0.0, 0, 0.0, 0
        for epoch in range(1, NUM_EPOCHS + 1):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            train_loss = train_epoch(model, train_loader, optimizer, device)
            torch.cuda.synchronize(); t_train += time.perf_counter() - t0; n_train += len(train_ids)
            val_loss, val_c_index = validate(model, val_loader, device)
            scheduler.step()
            if val_c_index > best_c_index:
                best_c_index, best_epoch = val_c_index, epoch
                torch.save(model.state_dict(), os.path.join(RESULTS_DIR, f"best_model_fold{fold}.pth"))
            if epoch % 10 == 0 or epoch == NUM_EPOCHS:
                print(f"[rank {rank}] fold {fold} epoch {epoch:3d} | Train Loss: {train_loss:.4f} | Val Loss: {val_loss:.4f} | "
                      f"Val C-index: {val_c_index:.4f} | Best: {best_c_index:.4f} (Epoch {best_epoch})", flush=True)
        local.append({"fold": fold, "best_c_index": best_c_index, "best_epoch": best_epoch, "train_size": int(len(train_ids)),
                      "val_size": int(len(val_ids)), "patients_per_sec": n_train / t_train})
    fold_results = D.gather_fold_results(local, world)
    if rank == 0:
        c = [r["best_c_index"] for r in fold_results]
        save_json(os.path.join(RESULTS_DIR, "cv_results.json"), {
            "model": "Simple-Fusion (RNA+Image)", "n_folds": N_FOLDS, "num_epochs": NUM_EPOCHS,
            "c_index_mean": float(np.mean(c)), "c_index_std": float(np.std(c)), "fold_results": fold_results})
        print(f"C-index: {np.mean(c):.4f} +/- {np.std(c):.4f}; saved {RESULTS_DIR}/cv_results.json")


if __name__ == "__main__":
    main()
