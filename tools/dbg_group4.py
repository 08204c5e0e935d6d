import copy, sys, ctypes, torch
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
from test_gpu_fold_group import _models, _kw
from test_gpu_models import _batch
from gpu_util import DEV
from multimodal_survival_prediction_amd import _lib
from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
lib = _lib.load_library()
cls, G, B, dims, rna_dim = "MultiModalSurvivalNet", 2, 4, (64, 64, 32), 1024
base = _models(cls, G, rna_dim, p_drop=0.0)
E = FoldGroupEngine([copy.deepcopy(m).to(DEV).train() for m in base])
Gr = FoldGroupEngine([copy.deepcopy(m).to(DEV).train() for m in base])
def region(P, name, idx, dt=torch.float32):
    off, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert lib.mms_dn121_region(B, *dims, name.encode(), idx, ctypes.byref(off), ctypes.byref(nb)) == 0
    return P.ws[off.value:off.value + nb.value].view(dt).clone()
for it in range(4):
    bs = [_kw(cls, *_batch(B, dims, rna_dim, 50 + 10 * it + g), None) for g in range(G)]
    E.train_step(bs, use_graph=False); Gr.train_step(bs, use_graph=it > 0); torch.cuda.synchronize()
    PE, PG = E.plan(B, dims).Ps, Gr.plan(B, dims).Ps
    for g in range(G):
        names = [('y0', 0)] + [('slab', i) for i in range(4)] + [('dslab', i) for i in range(4)]
        d = {f"{n}{i}": float((region(PE[g], n, i) - region(PG[g], n, i)).abs().max()) for n, i in names}
        se, sg = region(PE[g], 'stats', 0, torch.float64), region(PG[g], 'stats', 0, torch.float64)
        d['stats'] = float((se - sg).abs().max()); d['stats_nan'] = int((~torch.isfinite(sg)).sum())
        d['feats'] = float((PE[g].buf['feats'] - PG[g].buf['feats']).abs().max())
        d['dfeats'] = float((PE[g].dbuf['feats'] - PG[g].dbuf['feats']).abs().max())
        d['gflat'] = float((E.engines[g].gflat - Gr.engines[g].gflat).abs().max())
        d['flat'] = float((E.engines[g].flat - Gr.engines[g].flat).abs().max())
        print(f"it {it} model {g}: " + " ".join(f"{k}={v:.2e}" for k, v in d.items()), flush=True)
for tag, X in (("eager", E), ("graph", Gr)):
    for g in range(G):
        e = X.engines[g]
        for (k, p), gv in zip(e.model.named_parameters(), e.gviews):
            m = float(gv.abs().max())
            if not (m < 1e3):
                bad = (gv.abs() > 1e3) | ~torch.isfinite(gv)
                print(tag, g, k, tuple(gv.shape), "max", m, "count", int(bad.sum()), "first idx", bad.nonzero()[:4].tolist(), flush=True)
