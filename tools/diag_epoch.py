"""Per-batch diagnostic of tests/test_gpu_epoch_parity.py (GPU box): oracle vs HIP loss of every batch of the short epoch."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import test_gpu_epoch_parity as T
from oracle import losses as OL
from multimodal_survival_prediction_amd import data
from multimodal_survival_prediction_amd.training import FusedOptimizer
style = sys.argv[1] if len(sys.argv) > 1 else "final"
cls, adamw, wd = T.STYLES[style]
cohort = T._cohort()
if style == "final":
    cohort["has_survival"][:] = True
    lab = cohort["label"]
    lab[lab[:, 0] == 0, 0] = torch.arange(1, 1 + int((lab[:, 0] == 0).sum()), dtype=torch.float32) * 7.5
dev = data.cohort_to(cohort, T.DEV)
ref, net = T._pair(cls, 11)
LR, EPS = float(os.environ.get('LR', 1e-3)), float(os.environ.get('EPS', 1e-3))
opt = torch.optim.Adam(ref.parameters(), lr=LR, weight_decay=wd, eps=EPS)
fo = FusedOptimizer(net, lr=LR, weight_decay=wd, adamw=adamw, eps=EPS)
eng = fo.engine
ref.train(); net.train()
prev = 0.0
for b in range(6):
    j = torch.arange(b * 4, min(b * 4 + 4, 22))
    lab = cohort["label"][j]
    hz = ref(cohort["image"][j], cohort["rnaseq"][j], cohort["clinical"][j])
    loss = OL.cox_loss(hz, lab[:, 1], lab[:, 0])
    opt.zero_grad(); loss.backward(); torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0); opt.step()
    eng.train_step(dev["image"][j], dev["rnaseq"][j], dev["clinical"][j], time=dev["label"][j, 0], event=dev["label"][j, 1], skip_if_unusable=True)
    st = eng.epoch_stats()
    P = eng.plans[(len(j),) + T.DIMS]
    print(b, "oracle loss %.6f hip %.6f" % (loss.item(), st["sum_loss"] - prev), "hz oracle", hz.detach().numpy().round(5), "hip", P.buf["hz"][:, 0].cpu().numpy().round(5),
          "t", lab[:, 0].numpy(), "e", lab[:, 1].numpy())
    prev = st["sum_loss"]
