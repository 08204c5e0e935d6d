import copy, sys, torch
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
from test_gpu_fold_group import _models, _kw
from test_gpu_models import _batch
from gpu_util import DEV
from multimodal_survival_prediction_amd.engine import SurvivalEngine
from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
cls, G, B, dims, rna_dim = "MultiModalSurvivalNet", 2, 4, (64, 64, 32), 1024
for p_drop in (0.0, 0.3):
    base = _models(cls, G, rna_dim, p_drop=p_drop)
    A = [copy.deepcopy(m).to(DEV).train() for m in base]
    Bm = [copy.deepcopy(m).to(DEV).train() for m in base]
    C = [copy.deepcopy(m).to(DEV).train() for m in base]
    ea = [SurvivalEngine(m) for m in A]; eb = [SurvivalEngine(m) for m in Bm]; gc = FoldGroupEngine(C)
    for it in range(3):
        bs = [_kw(cls, *_batch(B, dims, rna_dim, 50 + 10 * it + g), None) for g in range(G)]
        for g in range(G):
            ea[g].train_step(use_graph=it > 0, **bs[g]); eb[g].train_step(use_graph=it > 0, **bs[g])
        gc.train_step(bs, use_graph=it > 0)
        torch.cuda.synchronize()
        def frac(X, Y):
            tot = close = 0
            for p, q in zip(X.parameters(), Y.parameters()):
                d = (p.detach() - q.detach()).abs(); tot += d.numel(); close += int((d <= 2e-5).sum())
            return close / tot
        print(f"p_drop {p_drop} step {it}: solo-vs-solo {frac(A[0], Bm[0]):.4f}  solo-vs-group {frac(A[0], C[0]):.4f}  gradrel solo/solo {float((ea[0].gflat-eb[0].gflat).abs().max()/ea[0].gflat.abs().max()):.2e} solo/group {float((ea[0].gflat-gc.engines[0].gflat).abs().max()/ea[0].gflat.abs().max()):.2e}", flush=True)
