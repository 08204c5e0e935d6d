"""the rest of the reference's scripts/training family on the same kernels --
RNASeqSurvivalModel (train_rnaseq_only.py) and FlexibleMultimodalModel (flexible_multimodal.py) -- against the CPU oracle
(itself pinned to the reference's classes by tests/golden/g6_extra_models.npz)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV, assert_close, rel_err
from test_gpu_densenet import structured_volumes
from test_gpu_models import _grad_stats


def _pair(cls, seed, **kw):
    from oracle import models as OM
    from multimodal_survival_prediction_amd import models as HM
    torch.manual_seed(seed)
    ref = getattr(OM, cls)(**kw) if cls == "RNASeqSurvivalModel" else getattr(OM, cls)(use_monai=True, **kw)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, (torch.nn.BatchNorm3d, torch.nn.BatchNorm1d)):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1)
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    net = getattr(HM, cls)(**kw)
    net.load_state_dict(ref.state_dict())
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return ref, net.to(DEV)


def _surv(B, seed):
    rng = np.random.default_rng(seed)
    t = torch.tensor((rng.exponential(1000, B) + 1 + np.arange(B) * 1e-3).astype(np.float32))
    e = torch.tensor((rng.random(B) < 0.6).astype(np.float32)); e[0] = 1
    return t, e


def test_missing_mix_op():
    from multimodal_survival_prediction_amd import _lib, ops
    lib, S = _lib.load_library(), _lib.structs()
    torch.manual_seed(0)
    B = 7
    f = torch.randn(B, 384, device=DEV); f0 = f.clone()
    mask = (torch.rand(B, 2, device=DEV) < 0.5).float()
    b0, b1 = torch.randn(128, device=DEV), torch.randn(256, device=DEV)
    d = torch.randn(B, 384, device=DEV); d0 = d.clone()
    g0, g1 = torch.zeros(128, device=DEV), torch.zeros(256, device=DEV)
    M = S["MixP"]()
    M.feats, M.ld, M.M, M.mask, M.ldm, M.nseg = f.data_ptr(), 384, B, mask.data_ptr(), 2, 2
    M.seg_begin[0], M.seg_width[0], M.seg_begin[1], M.seg_width[1] = 0, 128, 128, 256
    M.bias[0], M.bias[1], M.dfeats, M.ldd, M.dbias[0], M.dbias[1] = b0.data_ptr(), b1.data_ptr(), d.data_ptr(), 384, g0.data_ptr(), g1.data_ptr()
    import ctypes
    _lib.check(lib.mms_missing_mix_fwd(ctypes.byref(M), ops.stream()), "mix fwd")
    _lib.check(lib.mms_missing_mix_bwd(ctypes.byref(M), ops.stream()), "mix bwd")
    torch.cuda.synchronize()
    mi, mr = mask[:, 0:1], mask[:, 1:2]
    want = torch.cat([f0[:, :128] * mi + b0 * (1 - mi), f0[:, 128:] * mr + b1 * (1 - mr)], 1)
    assert_close(f, want, 1e-6, "mix fwd")
    assert_close(g0, (d0[:, :128] * (1 - mi)).sum(0), 1e-6, "dbias img"); assert_close(g1, (d0[:, 128:] * (1 - mr)).sum(0), 1e-6, "dbias rna")
    assert_close(d, torch.cat([d0[:, :128] * mi, d0[:, 128:] * mr], 1), 1e-6, "dfeats")


@pytest.mark.parametrize("B", [16, 5])
def test_rnaseq_model_parity_and_fused_step(B):
    from oracle import losses as OL
    from multimodal_survival_prediction_amd.training import FusedOptimizer
    ref, net = _pair("RNASeqSurvivalModel", 3, input_dim=5005)
    rng = np.random.default_rng(B)
    rna = torch.tensor(rng.normal(0, 1, (B, 5005)).astype(np.float32))
    t, e = _surv(B, 9)
    ref.eval(); net.eval()
    with torch.no_grad():
        assert_close(net(rna.to(DEV)), ref(rna), 1e-4, "eval log-hazard")
    ref.train(); net.train()
    hz = ref(rna).squeeze(); loss = OL.neg_partial_log_likelihood(hz, e.bool(), t); loss.backward()
    from multimodal_survival_prediction_amd import losses as HL
    hz2 = net(rna.to(DEV)).squeeze(); loss2 = HL.neg_partial_log_likelihood(hz2, e.to(DEV).bool(), t.to(DEV)); loss2.backward()
    torch.cuda.synchronize()
    assert_close(hz2, hz, 1e-4, "train log-hazard"); assert abs(loss2.item() - loss.item()) <= 1e-4 * max(1, abs(loss.item()))
    p10, mx, l2, hmax = _grad_stats(ref, net)
    assert hmax <= 1e-4, hmax
    # fused step == reference loop body (zero_grad, backward, AdamW step; NO clipping)
    ref, net = _pair("RNASeqSurvivalModel", 4, input_dim=5005)
    ref0 = copy.deepcopy(ref)
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=1e-3)
    fo = FusedOptimizer(net, lr=1e-4, weight_decay=1e-3, adamw=True, max_norm=0.0)
    ref.train(); net.train()
    for it in range(2):
        rna = torch.tensor(np.random.default_rng(40 + it).normal(0, 1, (B, 5005)).astype(np.float32))
        t, e = _surv(B, 50 + it)
        opt.zero_grad(); OL.neg_partial_log_likelihood(ref(rna).squeeze(), e.bool(), t).backward(); opt.step()
        fo.engine.train_step(None, rna, time=t, event=e, skip_if_unusable=False)
    torch.cuda.synchronize()
    worst = 0.0
    for (k, p), (_, q), (_, p0) in zip(ref.named_parameters(), net.named_parameters(), ref0.named_parameters()):
        worst = max(worst, float(((p.detach() - p0.detach()) - (q.detach().cpu() - p0.detach())).abs().max()))
    assert worst <= 4.2e-4, worst
    st = fo.engine.epoch_stats()
    assert st["n_batches"] == 2


def test_flexible_model_parity():
    from oracle import losses as OL
    from multimodal_survival_prediction_amd import losses as HL
    B, dims, rna_dim = 4, (64, 64, 32), 1024
    ref, net = _pair("FlexibleMultimodalModel", 6, rna_dim=rna_dim)
    ct = structured_volumes(B, dims, 3)
    rna = torch.tensor(np.random.default_rng(1).normal(0, 1, (B, rna_dim)).astype(np.float32))
    mask = torch.tensor([[1, 1], [0, 1], [1, 0], [0, 0]], dtype=torch.float32)
    t, e = _surv(B, 2)
    ref.eval(); net.eval()
    with torch.no_grad():
        assert_close(net(ct.to(DEV), rna.to(DEV), mask.to(DEV)), ref(ct, rna, mask), 1e-4, "eval log-hazard")
    ref.train(); net.train()
    hz = ref(ct, rna, mask); loss = OL.neg_partial_log_likelihood(hz, e.bool(), t); loss.backward()
    hz2 = net(ct.to(DEV), rna.to(DEV), mask.to(DEV)); loss2 = HL.neg_partial_log_likelihood(hz2, e.to(DEV).bool(), t.to(DEV)); loss2.backward()
    torch.cuda.synchronize()
    assert_close(hz2, hz, 1e-4, "train log-hazard")
    assert abs(loss2.item() - loss.item()) <= 1e-4 * max(1, abs(loss.item()))
    p10, mx, l2, hmax = _grad_stats(ref, net)
    assert hmax <= 1e-4, hmax                     # heads incl. the two missing-modality biases
    assert p10 <= 5e-5 and mx <= 0.15 and l2 <= 2e-2
    for k in ("missing_image_bias", "missing_rna_bias"):
        assert_close(dict(net.named_parameters())[k].grad, dict(ref.named_parameters())[k].grad, 1e-4, k)


@pytest.mark.parametrize("idim", [128, 64])
@pytest.mark.parametrize("cls", ["SimpleFusionModel", "FlexibleMultimodalModel"])
def test_rna_feature_dim_other_than_256(cls, idim):
    """simple_fusion.py:163 / flexible_multimodal.py: rna_feature_dim AND img_feature_dim are constructor arguments; the heads' feature
    buffer is rna_feature_dim + img_feature_dim wide (multiples of 4: 16-byte aligned columns), the DenseNet121 class_layers.out is
    1024 -> img_feature_dim (MmsDnOpts.out_features, an argument of every driver call)."""
    from oracle import losses as OL
    from multimodal_survival_prediction_amd import losses as HL, models as HM
    B, dims, rna_dim = 4, (64, 64, 32), 96          # (the headline volume: on 32^3 block 4 has ONE voxel per sample and BatchNorm over 4 values is ill-conditioned)
    ref, net = _pair(cls, 8, rna_dim=rna_dim, rna_feature_dim=64, img_feature_dim=idim)
    ct = structured_volumes(B, dims, 5)
    rna = torch.tensor(np.random.default_rng(2).normal(0, 1, (B, rna_dim)).astype(np.float32))
    mask = torch.tensor([[1, 1], [0, 1], [1, 0], [1, 1]], dtype=torch.float32)
    args = (ct, rna, mask) if cls == "FlexibleMultimodalModel" else (ct, rna)
    t, e = _surv(B, 3)
    ref.eval(); net.eval()
    with torch.no_grad():
        assert_close(net(*[a.to(DEV) for a in args]), ref(*args), 1e-4, "eval log-hazard")
    ref.train(); net.train()
    hz = ref(*args); loss = OL.neg_partial_log_likelihood(hz, e.bool(), t); loss.backward()
    hz2 = net(*[a.to(DEV) for a in args]); loss2 = HL.neg_partial_log_likelihood(hz2, e.to(DEV).bool(), t.to(DEV)); loss2.backward()
    torch.cuda.synchronize()
    assert_close(hz2, hz, 1e-4, "train log-hazard")
    assert abs(loss2.item() - loss.item()) <= 1e-4 * max(1, abs(loss.item()))
    p10, mx, l2, hmax = _grad_stats(ref, net)
    assert hmax <= 1e-4, hmax
    with pytest.raises(ValueError):
        getattr(HM, cls)(rna_dim=rna_dim, rna_feature_dim=66)          # not a multiple of 4: rejected, not mis-read


def test_lockstep_rnaseq_and_flexible_epochs():
    """train_epoch_lockstep / validate_lockstep styles 'rnaseq' and 'flexible' == the per-fold loops: frozen weights (lr = 0, dropout
    on; group-size independent launch options), so the order of the folds cannot matter -- returned means and validation losses at
    1e-4, the C-index from identical pair counts."""
    from gpu_util import GROUP_INDEPENDENT_OPTS as GI
    from multimodal_survival_prediction_amd import data, models as HM, training as T
    from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
    for style, cls, kw, okw in (("rnaseq", "RNASeqSurvivalModel", dict(input_dim=48), dict(lr=0.0, weight_decay=1e-3, adamw=True, max_norm=0.0, dn_opts=GI)),
                                ("flexible", "FlexibleMultimodalModel", dict(rna_dim=48), dict(lr=0.0, weight_decay=1e-3, adamw=True, dn_opts=GI))):
        cohort = data.cohort_to(data.make_cohort(n=30, dims=(32, 32, 32), rna_dim=48, seed=5, complete=True), DEV)
        cohort["mask"][::3, 0] = 0; cohort["mask"][1::4, 1] = 0
        folds = data.kfold_indices(30, 2, seed=1)
        ld = lambda f: (data.BatchLoader(cohort, folds[f][0], 4, shuffle=True, seed=10 + f, style="simple"),
                        data.BatchLoader(cohort, folds[f][1], 4, shuffle=False, style="simple"))
        base = []
        for f in range(2):
            torch.manual_seed(f); base.append(getattr(HM, cls)(**kw))
        seq = []
        for f in range(2):
            m = copy.deepcopy(base[f]).to(DEV)
            opt = T.FusedOptimizer(m, **okw)
            tl, vl = ld(f)
            seq.append((getattr(T, "train_epoch_" + style)(m, tl, opt, DEV), getattr(T, "validate_" + style)(m, vl, DEV)))
        ge = FoldGroupEngine([copy.deepcopy(b).to(DEV) for b in base], **okw)
        ls = [ld(f) for f in range(2)]
        tr = T.train_epoch_lockstep(ge, [l[0] for l in ls], style)
        va = T.validate_lockstep(ge, [l[1] for l in ls], style, DEV)
        for f in range(2):
            assert abs(seq[f][0] - tr[f]) <= 1e-4 * max(1.0, abs(seq[f][0])), (style, f, seq[f][0], tr[f])
            assert abs(seq[f][1][0] - va[f][0]) <= 1e-4 * max(1.0, abs(seq[f][1][0])), (style, f, seq[f][1], va[f])
            assert abs(seq[f][1][1] - va[f][1]) <= 1e-6, (style, f, seq[f][1], va[f])          # identical pair counts
