"""ONE FULL EPOCH of the CPU oracle on the metric's workload, on this box's host cores (an anchor for bench.py's bounded
cpu_baseline sample).  BASELINE config 3: 608 synthetic patients with modality masks, PartialModalityNet, fold 1 of the 5-fold CV (538 training
patients incl. the 260 unlabelled ones), batch 4, Adam(1e-4, wd 1e-4): oracle/loops.py train_epoch_partial over the whole training split, timed
after one untimed warm-up batch on a throw-away copy of the loop (thread pools, allocator).  Takes several minutes; prints one JSON line.
usage: python tools/cpu_oracle_epoch.py [--batches N]   (default: the whole epoch)"""
import argparse, json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from multimodal_survival_prediction_amd import data

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=0)
a = ap.parse_args()
B, K = 4, 5
cohort = data.make_cohort(n=608, dims=(64, 64, 32), rna_dim=5005, seed=608, complete=False)
has = cohort["has_survival"].numpy()
survival, non_survival = np.nonzero(has)[0], np.nonzero(~has)[0]
folds = data.kfold_indices(len(survival), K, seed=42)
train = np.concatenate([survival[folds[0][0]], non_survival])
nb = len(train) // B
steps = a.batches if a.batches > 0 else nb - 1
t0 = time.perf_counter()
r = bench.cpu_baseline_partial(cohort, torch.as_tensor(train), steps, B)
r["full_epoch"] = a.batches <= 0
r["batches_timed"] = steps
r["batches_in_epoch"] = nb
r["train_patients"] = int(len(train))
r["wall_s_incl_warmup"] = round(time.perf_counter() - t0, 1)
secs = steps * B / r["value"]
r["sample"] = ("%d of the %d batches (batch %d) of ONE EPOCH of the torch-fp32 CPU oracle's train_epoch_partial on fold 1's training split (%d patients) of "
               "bench.py's cohort -- all but the untimed warm-up batch --, %.1f s" % (steps, nb, B, len(train), secs)) if a.batches <= 0 else r["sample"]
print(json.dumps(r))
