"""Where does the host-resident (H2D) epoch spend its time?  (GPU box)"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from multimodal_survival_prediction_amd import data, models
from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
from multimodal_survival_prediction_amd.training import train_epoch_lockstep
dev = torch.device("cuda", 0)
c = data.make_cohort(n=608, dims=(64, 64, 32), rna_dim=5005, seed=608, complete=False)
has = c["has_survival"].numpy(); surv, non = np.nonzero(has)[0], np.nonzero(~has)[0]
folds = data.kfold_indices(len(surv), 5, seed=42)
tr = [np.concatenate([surv[f[0]], non]) for f in folds]
pinned = {k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
loaders = [data.HostBatchLoader(pinned, t, 4, shuffle=True, seed=142 + k, device=dev) for k, t in enumerate(tr)]
t0 = time.perf_counter()
n = 0
for l in loaders:
    for b in l:
        n += 1
torch.cuda.synchronize()
print("loader iteration alone: %.3f s for %d batches" % (time.perf_counter() - t0, n))
ms = [models.PartialModalityNet(rna_dim=5005).to(dev).train() for _ in range(5)]
g = FoldGroupEngine(ms, lr=1e-4, weight_decay=1e-4, adamw=False, gate_entropy_weight=0.01)
train_epoch_lockstep(g, loaders, "partial", concurrent=2)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
train_epoch_lockstep(g, loaders, "partial", concurrent=2)
torch.cuda.synchronize()
pr.disable()
print("epoch: %.3f s" % (time.perf_counter() - t0))
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
