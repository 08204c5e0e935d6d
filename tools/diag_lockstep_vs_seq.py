"""Diagnostic (GPU box): the lock-step K-fold driver against fold-after-fold training at lr = 0, per epoch and fold: train means, validation
loss, C-index.  usage: python tools/diag_lockstep_vs_seq.py [patients] [epochs]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from multimodal_survival_prediction_amd import data, models, training as T
from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine

n, epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 42, int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
cohort = data.cohort_to(data.make_cohort(n=n, dims=tuple(int(v) for v in os.environ.get("DIMS", "64,64,32").split(",")), seed=608, complete=False), dev)
has = cohort["has_survival"].cpu().numpy()
surv, non = np.nonzero(has)[0], np.nonzero(~has)[0]
folds = data.kfold_indices(len(surv), 3, seed=42)
splits = [(np.concatenate([surv[f[0]], non]), surv[f[1]]) for f in folds]
LR = float(os.environ.get("LR", "0"))

def mk_models():
    torch.manual_seed(42)
    ms = [models.PartialModalityNet().to(dev) for _ in range(3)]
    if os.environ.get("NODROP"):
        for m in ms:
            for q in m.modules():
                if isinstance(q, torch.nn.Dropout):
                    q.p = 0.0
    return ms

def loaders():
    return [(data.BatchLoader(cohort, tr, 4, shuffle=True, seed=42 + f), data.BatchLoader(cohort, va, 4, shuffle=False)) for f, (tr, va) in enumerate(splits)]

# sequential
seq = {}
for f, (m, (tl, vl)) in enumerate(zip(mk_models(), loaders())):
    opt = T.FusedOptimizer(m, lr=LR, weight_decay=1e-4, adamw=False, gate_entropy_weight=0.01)
    for ep in range(epochs):
        tr = T.train_epoch_partial(m, tl, opt, dev)
        va = T.validate_partial(m, vl, dev)
        seq[(f, ep)] = (tr, va)
# lock-step (as scripts/training/_common.py::cv_lockstep prepares its loaders)
ms = mk_models()
lds = loaders()
group = FoldGroupEngine(ms, lr=LR, weight_decay=1e-4, adamw=False, gate_entropy_weight=0.01)
for tl, vl in lds:
    for ld in (tl, vl):
        ld.lazy = True
        ld.view = data.gather_view(ld.c, with_valid=True)
        ld.hs_cpu = ld.c["has_survival"].cpu().tolist()
for ep in range(epochs):
    tr = T.train_epoch_lockstep(group, [l[0] for l in lds], "partial", concurrent=2)
    va = T.validate_lockstep(group, [l[1] for l in lds], "partial", dev, concurrent=2)
    for f in range(3):
        s = seq[(f, ep)]
        print("fold %d epoch %d: seq train (%.6f, %.6f) val (%.6f, %.4f) | lockstep train (%.6f, %.6f) val (%.6f, %.4f)" %
              (f, ep, s[0][0], s[0][1], s[1][0], s[1][1], tr[f][0], tr[f][1], va[f][0], va[f][1]))
