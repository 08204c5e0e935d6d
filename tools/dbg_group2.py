import copy, sys, torch
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
from test_gpu_fold_group import _models, _kw
from test_gpu_models import _batch
from gpu_util import DEV
from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
cls, G, B, dims, rna_dim = "MultiModalSurvivalNet", 2, 4, (64, 64, 32), 1024
for graph in (False, True):
    base = _models(cls, G, rna_dim, p_drop=0.0)
    C = [copy.deepcopy(m).to(DEV).train() for m in base]
    gc = FoldGroupEngine(C)
    for it in range(4):
        bs = [_kw(cls, *_batch(B, dims, rna_dim, 50 + 10 * it + g), None) for g in range(G)]
        gc.train_step(bs, use_graph=graph and it > 0)
        torch.cuda.synchronize()
        for g in range(G):
            e = gc.engines[g]
            bad = [k for k, p in C[g].named_parameters() if not torch.isfinite(p).all()]
            badg = [k for (k, p), gv in zip(C[g].named_parameters(), e.gviews) if not torch.isfinite(gv).all()]
            badb = [k for k, b in C[g].named_buffers() if not torch.isfinite(b.float()).all()]
            print(f"graph={graph} step {it} model {g}: acc {e.acc.tolist()} sumsq {float(e.sumsq):.3e} gmax {float(e.gflat.abs().max()):.3e} nan params {len(bad)} {bad[:3]} nan grads {len(badg)} {badg[:3]} nan bufs {len(badb)} {badb[:3]}", flush=True)
