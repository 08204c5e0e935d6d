"""Fold groups (G models advanced by ONE launch sequence) must follow the trajectories of the same models trained
one at a time: same kernels, same per-model arithmetic -- only fp32/fp64 atomic ordering may differ."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV, GROUP_INDEPENDENT_OPTS as GI, rel_err
from test_gpu_models import _batch


def _models(cls, G, rna_dim, p_drop=0.0):
    from multimodal_survival_prediction_amd import models as HM
    out = []
    for g in range(G):
        torch.manual_seed(100 + g)
        m = getattr(HM, cls)(rna_dim=rna_dim)
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, (torch.nn.BatchNorm3d, torch.nn.BatchNorm1d)):
                    mod.weight.uniform_(0.5, 1.5); mod.bias.normal_(0, 0.1)
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = p_drop
        out.append(m)
    return out


def _kw(cls, ct, rna, clin, t, e, mask, valid):
    if cls == "MultiModalSurvivalNet":
        return dict(ct=ct, rna=rna, clinical=clin, time=t, event=e)
    if cls == "PartialModalityNet":
        return dict(ct=ct, rna=rna, clinical=clin, mask=mask, time=t, event=e, valid=valid)
    return dict(ct=ct, rna=rna, time=t, event=e, valid=valid)


@pytest.mark.parametrize("cls,G", [("MultiModalSurvivalNet", 3), ("PartialModalityNet", 2), ("SimpleFusionModel", 5)])
def test_group_step_equals_single_steps(cls, G):
    # group-size independent launch options (gpu_util.GROUP_INDEPENDENT_OPTS) on both sides: identical per-model arithmetic, so the
    # comparison is tight; with the default, group-size dependent kernel forms the fp32 summation order differs and parity becomes
    # statistical (ReLU-mask flips, see test_gpu_densenet.py) -- covered by test_group_default_options below
    from multimodal_survival_prediction_amd.engine import SurvivalEngine
    from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
    B, dims, rna_dim = 4, (64, 64, 32), 1024
    # dropout stays ON (p = 0.3 everywhere): the hash-RNG streams are per model and must advance identically
    base = _models(cls, G, rna_dim, p_drop=0.3)
    solo = [copy.deepcopy(m).to(DEV).train() for m in base]
    grp = [copy.deepcopy(m).to(DEV).train() for m in base]
    skip = cls != "PartialModalityNet"
    kw = dict(lr=1e-4, weight_decay=1e-3 if cls == "SimpleFusionModel" else 1e-4, dn_opts=GI)
    se = [SurvivalEngine(m, **kw) for m in solo]
    ge = FoldGroupEngine(grp, **kw)
    valid = torch.tensor([1, 1, 0, 1], dtype=torch.float32)
    for it in range(3):
        batches = []
        for g in range(G):
            ct, rna, clin, t, e, mask = _batch(B, dims, rna_dim, 50 + 10 * it + g)
            if it == 1 and g == 1:
                e = torch.zeros_like(e)       # a batch without events: unusable -> skipped (or entropy-only) step for that member
            batches.append(_kw(cls, ct, rna, clin, t, e, mask, valid))
        for g in range(G):
            se[g].train_step(skip_if_unusable=skip, use_graph=it > 0, **batches[g])
        ge.train_step(batches, skip_if_unusable=skip, use_graph=it > 0)
        torch.cuda.synchronize()
        if it == 0:     # gradients of the very first step: identical inputs and weights on both sides
            for g in range(G):
                a, b = se[g].gflat, ge.engines[g].gflat
                assert rel_err(b, a) <= 2e-5, (g, rel_err(b, a))
    stats_s = [e.epoch_stats() for e in se]
    stats_g = ge.epoch_stats()
    for g in range(G):
        assert stats_g[g]["n_batches"] == 3 and stats_g[g]["n_usable"] == stats_s[g]["n_usable"]
        assert abs(stats_g[g]["sum_loss"] - stats_s[g]["sum_loss"]) <= 5e-2 * max(1.0, abs(stats_s[g]["sum_loss"])), (g, stats_g[g], stats_s[g])
        assert abs(stats_g[g]["sum_entropy"] - stats_s[g]["sum_entropy"]) <= 1e-3 * max(1.0, abs(stats_s[g]["sum_entropy"]))
        assert int(ge.engines[g].rng[1]) == int(se[g].rng[1]) == 3
        assert float(ge.engines[g].step_count) == float(se[g].step_count)
        # Adam turns rounding-noise gradients (exact-zero-gradient parameters) into +-lr moves, so compare in bulk
        tot = close = 0
        worst = 0.0
        for p, q in zip(solo[g].parameters(), grp[g].parameters()):
            d = (p.detach() - q.detach()).abs()
            tot += d.numel(); close += int((d <= 2e-5).sum()); worst = max(worst, float(d.max()))
        assert worst <= 6.5e-4, worst
        assert close / tot >= 0.9, close / tot
        for (k, b), (_, c) in zip(solo[g].named_buffers(), grp[g].named_buffers()):
            if "num_batches" in k:
                assert int(b) == int(c)
            else:
                assert rel_err(c, b) <= 1e-2, k          # 3 chaotic steps; block-4 statistics are over 16 rows only
    print(f"{cls} x{G}: group == single steps")


@pytest.mark.parametrize("G", [2, 4])
def test_group_default_options(G):
    """DEFAULT launch options (what bench.py and the entry points run; G = 2 is the sub-group size of the K = 5 epoch): a group of G
    models against the same models stepped alone.  The kernel forms then differ between the two sides (group-size dependent tap
    split, tile shapes, small-launch kernels), so sums run in another order: everything that does not pass through a ReLU-mask flip
    agrees at the north_star 1e-4 -- training-mode losses of two steps at lr = 0 (the second a graph replay), BatchNorm running
    statistics, head gradients -- and the encoder gradients statistically (same bound as the network-level parity tests)."""
    from multimodal_survival_prediction_amd.engine import SurvivalEngine
    from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
    cls, B, dims, rna_dim = "MultiModalSurvivalNet", 4, (64, 64, 32), 512
    base = _models(cls, G, rna_dim)
    solo = [copy.deepcopy(m).to(DEV).train() for m in base]
    grp = [copy.deepcopy(m).to(DEV).train() for m in base]
    se = [SurvivalEngine(m, lr=0.0) for m in solo]
    ge = FoldGroupEngine(grp, lr=0.0)
    for it in range(2):
        batches = []
        for g in range(G):
            ct, rna, clin, t, e, mask = _batch(B, dims, rna_dim, 90 + 7 * it + g)
            batches.append(dict(ct=ct, rna=rna, clinical=clin, time=t, event=e))
            se[g].reset_epoch_stats()
            se[g].train_step(use_graph=it > 0, **batches[g])
        ge.reset_epoch_stats()
        ge.train_step(batches, use_graph=it > 0)
        torch.cuda.synchronize()
        for g in range(G):
            a, b = se[g].gflat.double(), ge.engines[g].gflat.double()
            l2 = float(((a - b) ** 2).sum().sqrt() / (a ** 2).sum().sqrt())
            assert l2 <= 1e-2, (it, g, l2)                      # same bound as the network-level gradient parity tests
            sa, sb = se[g].epoch_stats(), ge.engines[g].epoch_stats()
            assert abs(sa["sum_loss"] - sb["sum_loss"]) <= 1e-4 * max(1.0, abs(sa["sum_loss"])), (it, g, sa, sb)
            heads = [(p, q) for (k, p), (_, q) in zip(solo[g].named_parameters(), grp[g].named_parameters()) if not k.startswith("ct_encoder")]
            gs = dict(zip((id(p) for p in se[g].params), se[g].gviews)); gg = dict(zip((id(p) for p in ge.engines[g].params), ge.engines[g].gviews))
            top = max(float(gs[id(p)].abs().max()) for p, _ in heads)
            for p, q in heads:
                if float(gs[id(p)].abs().max()) > 1e-5 * top:          # (a bias in front of a BatchNorm has a gradient of rounding noise only)
                    assert rel_err(gg[id(q)], gs[id(p)]) <= 2e-4, (it, g)
    for g in range(G):
        for (k, b), (_, c) in zip(solo[g].named_buffers(), grp[g].named_buffers()):
            if "num_batches" in k:
                assert int(b) == int(c) == 2
            else:
                assert rel_err(c, b) <= 1e-4, (g, k)


def test_group_subset_and_eval():
    """A sub-group (ragged tail: only some folds still have a batch) and the grouped eval forward."""
    from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
    cls, G, B, dims, rna_dim = "MultiModalSurvivalNet", 3, 3, (32, 32, 32), 256
    base = _models(cls, G, rna_dim)
    grp = [copy.deepcopy(m).to(DEV).train() for m in base]
    ge = FoldGroupEngine(grp)
    batches = []
    for g in range(G):
        ct, rna, clin, t, e, mask = _batch(B, dims, rna_dim, 7 + g)
        batches.append(dict(ct=ct, rna=rna, clinical=clin, time=t, event=e))
    ge.train_step([batches[0], batches[2]], members=(0, 2))
    torch.cuda.synchronize()
    st = ge.epoch_stats()
    assert [s["n_batches"] for s in st] == [1, 0, 1]
    for m in grp:
        m.eval()
    ev = [dict(ct=b["ct"], rna=b["rna"], clinical=b["clinical"]) for b in batches]
    outs = ge.forward_eval(ev)
    torch.cuda.synchronize()
    for g in range(G):
        hz_single = grp[g](ev[g]["ct"].to(DEV), ev[g]["rna"].to(DEV), ev[g]["clinical"].to(DEV))
        assert rel_err(outs[g][0], hz_single.reshape(-1)) <= 1e-5


def test_group_rejects_mismatched_shapes():
    from multimodal_survival_prediction_amd import _lib, ops
    lib = _lib.load_library()
    S = _lib.structs()
    x = torch.zeros(64, 64, device=DEV)
    a = ops.adam_params(x.view(-1), x.view(-1), x.view(-1), x.view(-1), torch.zeros(6, device=DEV),
                        torch.zeros(1, dtype=torch.float64, device=DEV), torch.zeros(1, device=DEV))
    y = torch.zeros(32, device=DEV)
    b = ops.adam_params(y, y, y, y, torch.zeros(6, device=DEV), torch.zeros(1, dtype=torch.float64, device=DEV),
                        torch.zeros(1, device=DEV))
    arr = (S["AdamP"] * 2)(a, b)
    assert lib.mms_grad_sumsq_group(arr, 2, ops.stream()) == -1       # MMS_ERR_ARG: sizes differ
    assert lib.mms_grad_sumsq_group(arr, 11, ops.stream()) == -1      # > MMS_MAX_GROUP


def test_indexed_step_equals_batch_step():
    """train_step_indexed (one gather launch from the device-resident cohort) == train_step on the same batches."""
    from multimodal_survival_prediction_amd import data
    from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
    cls, G, B, dims, rna_dim = "PartialModalityNet", 2, 4, (32, 32, 32), 64
    cohort = data.cohort_to(data.make_cohort(n=24, dims=dims, rna_dim=rna_dim, seed=3, complete=False), DEV)
    cohort["valid"] = cohort["has_survival"].float()
    base = _models(cls, G, rna_dim)
    A = FoldGroupEngine([copy.deepcopy(m).to(DEV).train() for m in base], dn_opts=GI)
    Bg = FoldGroupEngine([copy.deepcopy(m).to(DEV).train() for m in base], dn_opts=GI)
    rng = np.random.default_rng(0)
    for it in range(3):
        idx = np.stack([rng.permutation(24)[:B] for _ in range(G)])
        batches = []
        for g in range(G):
            j = torch.as_tensor(idx[g], device=DEV)
            lab = cohort["label"][j]
            batches.append(dict(ct=cohort["image"][j], rna=cohort["rnaseq"][j], clinical=cohort["clinical"][j],
                                mask=cohort["mask"][j], time=lab[:, 0], event=lab[:, 1], valid=cohort["valid"][j]))
        A.train_step(batches, skip_if_unusable=False, use_graph=it > 0)
        Bg.train_step_indexed(cohort, idx, skip_if_unusable=False, use_graph=it > 0)
        torch.cuda.synchronize()
        if it == 0:
            for g in range(G):
                assert rel_err(Bg.engines[g].gflat, A.engines[g].gflat) <= 2e-5
    for a, b in zip(A.epoch_stats(), Bg.epoch_stats()):
        assert a["n_batches"] == b["n_batches"] == 3 and a["n_usable"] == b["n_usable"]
        # three chaotic fp32 steps apart (atomic ordering -> Adam sign flips on noise-level gradients): loose on the loss sum
        assert abs(a["sum_loss"] - b["sum_loss"]) <= 5e-2 * max(1.0, abs(a["sum_loss"]))


@pytest.mark.parametrize("style", ["final", "partial", "simple"])
@pytest.mark.parametrize("lr", [0.0, 1e-4])
def test_lockstep_epoch_matches_sequential(style, lr):
    """train_epoch_lockstep / validate_lockstep == train_epoch_<style> / validate_<style> fold by fold (ragged fold sizes:
    the last batch positions run as sub-groups).  lr = 0 (frozen weights, dropout on): the order of the folds cannot matter --
    returned means and validation losses at 1e-4, C-index from identical pair counts.  lr = 1e-4: two correct fp32 runs of a
    chaotic trajectory (atomic summation order) -- loose, liveness + bookkeeping only."""
    from multimodal_survival_prediction_amd import data, models as HM, training as T
    from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
    dims, rna_dim, K, B = (32, 32, 32), 48, 3, 4
    cohort = data.cohort_to(data.make_cohort(n=29, dims=dims, rna_dim=rna_dim, seed=5, complete=(style != "partial")), DEV)
    folds = data.kfold_indices(29, K, seed=1)
    cls = {"final": "MultiModalSurvivalNet", "partial": "PartialModalityNet", "simple": "SimpleFusionModel"}[style]
    bstyle = "simple" if style == "simple" else "final"
    def loaders(f):
        return (data.BatchLoader(cohort, folds[f][0], B, shuffle=True, seed=10 + f, style=bstyle),
                data.BatchLoader(cohort, folds[f][1], B, shuffle=False, style=bstyle))
    base = []
    for f in range(K):
        torch.manual_seed(f)
        base.append(getattr(HM, cls)(rna_dim=rna_dim))
    kw = dict(lr=lr, weight_decay=1e-4, adamw=(style == "simple"), dn_opts=GI)
    # sequential: the reference's order
    seq = []
    for f in range(K):
        m = copy.deepcopy(base[f]).to(DEV)
        opt = T.FusedOptimizer(m, **kw)
        tl, vl = loaders(f)
        tr = getattr(T, "train_epoch_" + style)(m, tl, opt, DEV)
        va = getattr(T, "validate_" + style)(m, vl, DEV)
        seq.append((tr, va))
    grp_models = [copy.deepcopy(b).to(DEV) for b in base]
    ge = FoldGroupEngine(grp_models, **kw)
    ls = [loaders(f) for f in range(K)]
    tr = T.train_epoch_lockstep(ge, [l[0] for l in ls], style)
    va = T.validate_lockstep(ge, [l[1] for l in ls], style, DEV)
    for f in range(K):
        a, b = np.atleast_1d(np.asarray(seq[f][0], dtype=float)), np.atleast_1d(np.asarray(tr[f], dtype=float))
        if lr == 0:
            assert np.allclose(a, b, rtol=1e-4, atol=1e-6), (f, a, b)
            assert abs(seq[f][1][0] - va[f][0]) <= 1e-4 * max(1.0, abs(seq[f][1][0])), (f, seq[f][1], va[f])
            assert abs(seq[f][1][1] - va[f][1]) <= 1e-6, (f, seq[f][1], va[f])          # identical concordant / discordant pair counts
            continue
        assert np.allclose(a, b, rtol=8e-2, atol=5e-3), (f, a, b)        # one epoch of chaotic fp32 training: loose on the mean loss
        assert abs(seq[f][1][0] - va[f][0]) <= 5e-2 * max(1.0, abs(seq[f][1][0])), (f, seq[f][1], va[f])


def test_lockstep_two_streams_matches_one():
    """concurrent=2 (two sub-groups on two streams) trains the same folds as concurrent=1: at lr = 0 (frozen weights, dropout on) the
    returned means agree at 1e-4 whatever the stream layout."""
    from multimodal_survival_prediction_amd import data, models as HM, training as T
    from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
    dims, rna_dim, K, B = (32, 32, 32), 48, 5, 4
    cohort = data.cohort_to(data.make_cohort(n=45, dims=dims, rna_dim=rna_dim, seed=5, complete=True), DEV)
    folds = data.kfold_indices(45, K, seed=1)
    ld = lambda f: data.BatchLoader(cohort, folds[f][0], B, shuffle=True, seed=10 + f)
    base = []
    for f in range(K):
        torch.manual_seed(f); base.append(HM.MultiModalSurvivalNet(rna_dim=rna_dim))
    res = []
    lazy = lambda f: data.BatchLoader(cohort, folds[f][0], B, shuffle=True, seed=10 + f, lazy=True, with_valid=False)
    for conc, mk in ((1, ld), (2, ld), (2, lazy)):      # lazy: batches named by index, assembled by the group's one gather launch
        ge = FoldGroupEngine([copy.deepcopy(b).to(DEV) for b in base], lr=0.0, weight_decay=1e-4, dn_opts=GI)
        res.append((T.train_epoch_lockstep(ge, [mk(f) for f in range(K)], "final", concurrent=conc), ge.epoch_stats()))
    for f in range(K):
        assert res[0][1][f]["n_batches"] == res[1][1][f]["n_batches"] == res[2][1][f]["n_batches"] == 9
        for other in (1, 2):
            assert abs(res[0][0][f] - res[other][0][f]) <= 1e-4 * max(1.0, abs(res[0][0][f])), (f, other, res[0][0][f], res[other][0][f])


def test_training_batch_of_one_raises_like_torch():
    """A training step on ONE patient: the reference's loops die in torch's BatchNorm1d ("Expected more than 1 value per channel when
    training"); the engines raise the same ValueError instead of a kernel argument error."""
    from multimodal_survival_prediction_amd.engine import SurvivalEngine
    from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
    ms = [m.to(DEV).train() for m in _models("SimpleFusionModel", 3, 32)]
    ct, rna = torch.zeros(1, 1, 32, 32, 32, device=DEV), torch.zeros(1, 32, device=DEV)
    kw = dict(ct=ct, rna=rna, time=torch.ones(1, device=DEV), event=torch.ones(1, device=DEV), valid=torch.ones(1, device=DEV))
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        SurvivalEngine(ms[0]).train_step(**kw)
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        FoldGroupEngine(ms[1:]).train_step([kw, kw])
