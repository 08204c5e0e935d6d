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
C = [copy.deepcopy(m).to(DEV).eval() for m in base]
gc = FoldGroupEngine(C)
def region(P, name, idx):
    off, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert lib.mms_dn121_region(B, *dims, name.encode(), idx, ctypes.byref(off), ctypes.byref(nb)) == 0
    return P.ws[off.value:off.value + nb.value].view(torch.float32).clone()
def ev(it):
    return [dict(ct=b[0], rna=b[1], clinical=b[2]) for b in [_batch(B, dims, rna_dim, 50 + 10 * it + g) for g in range(G)]]
for it in range(3):
    out = gc.forward_eval(ev(it), use_graph=True); torch.cuda.synchronize()
    hz_g = [o[0].clone() for o in out]
    GP = gc.plan(B, dims)
    regs_g = [[region(P, 'y0', 0)] + [region(P, 'slab', i) for i in range(4)] for P in GP.Ps]
    out = gc.forward_eval(ev(it), use_graph=False); torch.cuda.synchronize()
    hz_e = [o[0].clone() for o in out]
    regs_e = [[region(P, 'y0', 0)] + [region(P, 'slab', i) for i in range(4)] for P in GP.Ps]
    for g in range(G):
        d = [float((a - b).abs().max()) for a, b in zip(regs_g[g], regs_e[g])]
        print(f"it {it} model {g}: hz graph {hz_g[g].tolist()} eager {hz_e[g].tolist()} region diffs y0/slab0..3 {d}", flush=True)
