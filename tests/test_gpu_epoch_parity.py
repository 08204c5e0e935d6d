"""Epoch-level parity: the HIP `train_epoch_*` / `validate_*` functions -- fused-graph steps,
device-side loss accumulation, skip rules, denominators -- against the CPU oracle's restatement of the reference loops
(oracle/loops.py: final_multimodal.py:238-305, partial_modality_training.py:382-485, simple_fusion.py:242-333) on ONE short epoch of
a cohort that contains every special case of those loops:
  batch 0  all labelled, events present                       -> ordinary step
  batch 1  all labelled, NO event                             -> final: zero loss without graph, no update; partial: entropy-only
                                                                 step, not counted in the Cox mean; simple: forward runs (BatchNorm
                                                                 running statistics move), then `continue`
  batch 2  ONE labelled patient + 3 unlabelled                -> partial: entropy-only step; simple: skipped BEFORE the forward
  batch 3  2 labelled (1 event) + 2 unlabelled                -> Cox on the labelled subset only
  batch 4  all labelled
  batch 5  ragged tail of 2 patients
Dropout is off (its RNG cannot be matched); DenseNet121-3D encoder on the headline 64x64x32 volumes.

Two variants per style.
  lr = 0 ("frozen weights"): everything but the weight update runs -- per-batch losses, skip rules, denominators, BatchNorm running
      statistics, validation loss, C-index, held-out hazards -- and must agree at the north_star tolerance 1e-4.
  lr = 1e-4 (the scripts' setting): after the first Adam step the per-batch losses of ANY two arithmetics part ways at the 1e-3 level --
      measured here for the CPU oracle against its own fp64 run (2.2e-3 on the fourth batch, 5.2e-3 on the fifth) -- so agreement at 1e-4
      cannot be asked of an fp32 implementation there.  How far a correct one may drift is measured, not assumed: the oracle loop runs a
      second time in fp64 (exact for this purpose); every per-batch loss of the HIP epoch, its returned means and its BatchNorm running
      statistics must lie within ENV_FACTOR x the fp32 oracle's own distance from the fp64 run (running maximum over the batches so far:
      the drift grows along the epoch) + 1e-5.  The factor is not 1 because the distances are single draws of a chaotic quantity; it is
      set to twice the worst HIP / oracle drift ratio recorded over the fixtures (profiles/r04_envelope_ratios.txt), not fitted higher.  (Raising Adam's eps to
      1e-3 -- which turns lr * sign(g) moves of noise-level gradient entries into negligible ones -- did NOT remove the drift: 1.5e-3 on the
      fourth batch; it is the update of the real gradient entries through 121 training-mode BatchNorm layers that is this sensitive.)
      Net effect: the returned means are bounded at ~1e-2 absolute by evidence, where the previous round accepted 3e-2 relative without.
The exact counts (denominators, skip rules) are asserted in every variant; the update arithmetic is also pinned at step level in
tests/test_gpu_models.py and tests/test_gpu_heads.py.
Config 1 of BASELINE.json (simple_fusion, 88 complete patients, RNA-seq 5005-d, CT encoder input stubbed to zeros) runs at the end."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV, assert_close
from test_gpu_densenet import structured_volumes

import os as _os
DIMS, RNA = tuple(int(v) for v in _os.environ.get("MMS_TEST_DIMS", "64,64,32").split(",")), 96
# How far the HIP path may sit from the fp64 run, in units of the fp32 CPU oracle's own distance from it: 2 x the worst ratio recorded on
# the MI355X over every envelope test of the suite (profiles/r04_envelope_ratios.txt: worst 2.05 in two full runs -> 4.5; rounds 2-3 used 8).
ENV_FACTOR = 4.5


def _record_ratio(test, what, ratio):
    """Worst HIP / oracle drift ratios go to stdout and, on the GPU box, to gpurun_out/envelope_ratios.txt (-> profiles/)."""
    line = "%s | %s | %.3f" % (test, what, ratio)
    print("  envelope ratio:", line)
    d = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gpurun_out")
    if _os.path.isdir(d):
        with open(_os.path.join(d, "envelope_ratios.txt"), "a") as fh:
            fh.write(line + "\n")


def _cohort(n_extra_val=10, seed=5):
    """22 training patients laid out as the six batches above + a validation split with an unlabelled patient, a batch without
    events and a single-labelled batch."""
    rng = np.random.default_rng(seed)
    n = 22 + n_extra_val
    img = structured_volumes(n, DIMS, seed).reshape(n, 1, *DIMS)
    rna = rng.normal(0, 1, (n, RNA)).astype(np.float32)
    age = (np.clip(rng.normal(60, 11, n), 30, 90) / 100).astype(np.float32)
    time = (rng.exponential(1000, n) + 1 + np.arange(n) * 1e-2).astype(np.float32)
    event = (rng.random(n) < 0.6).astype(np.float32)
    has = np.ones(n, bool)
    event[0], event[3] = 1, 0                     # batch 0: events and a censored patient
    event[4:8] = 0                                # batch 1: no event
    has[9:12] = False; event[8] = 1               # batch 2: one labelled patient
    has[14:16] = False; event[12], event[13] = 1, 0   # batch 3
    event[16] = 1                                 # batch 4
    event[20], event[21] = 1, 0                   # tail
    v = 22                                        # validation: batch A ordinary, batch B no event, batch C one labelled + tail
    event[v] = 1
    event[v + 4:v + 8] = 0
    has[v + 8] = False
    event[v + 9] = 1
    mask = np.ones((n, 3), np.float32)
    mask[1, 0] = 0; mask[5, 1] = 0; mask[10, 0] = 0; mask[13, 2] = 0; mask[17, 0] = 0; mask[v + 1, 0] = 0
    img[mask[:, 0] == 0] = 0.0
    rna[mask[:, 1] == 0] = 0.0
    clin = age * mask[:, 2]
    time = np.where(has, time, 0.0).astype(np.float32)
    event = np.where(has, event, 0.0).astype(np.float32)
    return dict(image=img.contiguous(), rnaseq=torch.tensor(rna), clinical=torch.tensor(clin).view(n, 1),
                label=torch.tensor(np.stack([time, event], 1)), mask=torch.tensor(mask), has_survival=torch.tensor(has), n=n, dims=DIMS)


def _ref(cls, seed, rna_dim=None):
    """The CPU oracle model of a (class, seed) pair -- also what tests/golden/generate_fp64_envelope.py builds (no GPU needed)."""
    from oracle import models as OM
    torch.manual_seed(seed)
    ref = getattr(OM, cls)(rna_dim=rna_dim or globals()["RNA"], use_monai=True)
    for m in ref.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return ref


def _pair(cls, seed, rna_dim=None):
    from multimodal_survival_prediction_amd import models as HM
    ref = _ref(cls, seed, rna_dim)
    RNA = rna_dim or globals()["RNA"]
    net = getattr(HM, cls)(rna_dim=RNA)
    net.load_state_dict(ref.state_dict())
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return ref, net.to(DEV)


def _loaders(cohort, style, dev_cohort):
    from multimodal_survival_prediction_amd import data
    tr, va = np.arange(22), np.arange(22, cohort["n"])
    mk = lambda c, idx: data.BatchLoader(c, idx, 4, shuffle=False, style=style)
    return mk(cohort, tr), mk(cohort, va), mk(dev_cohort, tr), mk(dev_cohort, va)


# ---- fp64 legs of the envelope tests as fixtures ------------------------------------------------------------------------------
# The fp64 run of the oracle loops is a pure CPU computation of the oracle (no HIP code in it) and, in fp64, reproducible to ~1e-12
# on any machine: tests/golden/generate_fp64_envelope.py runs it once and commits the per-batch losses; the GPU tests read them
# instead of spending a CPU epoch of DenseNet121 in fp64 each (the slowest part of the GPU suite).  A fixture is used only when its
# fingerprint (initial weights, cohort, hyper-parameters, the oracle's own source text, torch version) matches what the test has in
# hand; otherwise the leg is computed live.
_FX_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "fp64_envelope.json")


def _fingerprint(model64, tensors, hyper=""):
    """sha256 over everything the fp64 leg depends on: the initial weights and the input tensors BYTE by byte (order-sensitive), the
    hyper-parameter string, the text of oracle/*.py (an edited oracle loop invalidates the fixture) and the torch version."""
    import hashlib
    h = hashlib.sha256()
    for q in model64.parameters():
        h.update(q.detach().double().contiguous().numpy().tobytes())
    for t in tensors:
        h.update(torch.as_tensor(t).detach().double().contiguous().numpy().tobytes())
    h.update(hyper.encode())
    odir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "oracle")
    for name in sorted(n for n in _os.listdir(odir) if n.endswith(".py")):
        with open(_os.path.join(odir, name), "rb") as fh:
            h.update(name.encode()); h.update(fh.read())
    return "%s|torch %s" % (h.hexdigest()[:32], torch.__version__.split("+")[0])


def _hyper(style, lr, eps=1e-8, B=4, extra=""):
    cls, adamw, wd = STYLES[style]
    return "style=%s cls=%s adamw=%s lr=%g wd=%g eps=%g B=%d dropout=0 %s" % (style, cls, adamw, lr, wd, eps, B, extra)


def _fp64_leg(key, fingerprint, compute):
    import json
    fx = {}
    if _os.path.exists(_FX_PATH):
        with open(_FX_PATH) as f:
            fx = json.load(f)
    e = fx.get(key)
    if e is not None and e["fingerprint"] == fingerprint and not _os.environ.get("MMS_FP64_LIVE"):
        print("fp64 leg", key, "from", _os.path.basename(_FX_PATH))
        return e["value"]
    print("fp64 leg", key, "computed live" + ("" if e is None else " (fixture fingerprint %s != %s)" % (e["fingerprint"], fingerprint)))
    value = compute()
    if _os.environ.get("MMS_WRITE_FP64_FIXTURES") == "1":
        fx[key] = dict(fingerprint=fingerprint, value=value)
        with open(_FX_PATH, "w") as f:
            json.dump(fx, f, indent=1, sort_keys=True)
    return value


def _fp64_epoch_leg(style, lr, ref64, cohort, eps=1e-8, split=None, buffers=False):
    """fp64 run of oracle/loops.train_epoch_<style> over the 22-patient training split (or `split`): -> dict(want64=[...], pb64=[[...], ...]
    [, buf64 = {BatchNorm running statistic: values} of the heads and of the encoder's first and last BatchNorm])."""
    from oracle import loops as OLP
    from multimodal_survival_prediction_amd import data
    cls, adamw, wd = STYLES[style]
    opt64 = (torch.optim.AdamW(ref64.parameters(), lr=lr, weight_decay=wd, eps=eps) if adamw
             else torch.optim.Adam(ref64.parameters(), lr=lr, weight_decay=wd, eps=eps))
    pb64 = []
    tr_c = data.BatchLoader(cohort, np.arange(22) if split is None else split, 4, shuffle=False, style=style)
    want64 = getattr(OLP, "train_epoch_" + style)(ref64, _cast_loader(tr_c, torch.float64), opt64, torch.device("cpu"),
                                                  on_batch=lambda *v: pb64.append([float(x) for x in v]))
    out = dict(want64=[float(x) for x in (want64 if isinstance(want64, tuple) else (want64,))], pb64=pb64)
    if buffers:
        out["buf64"] = {k: [float(x) for x in b.reshape(-1)] for k, b in ref64.named_buffers() if _kept_buffer(k)}
    return out


def _kept_buffer(k):
    """The running statistics a lock-step envelope leg stores (the fixture stays small): every head BatchNorm1d, the stem's norm0 and the
    encoder's last BatchNorm (norm5: downstream of all 120 others)."""
    return "num_batches" not in k and (not k.startswith("ct_encoder") or ".norm0." in k or ".norm5." in k)


LOCKSTEP_SPLITS = [np.arange(22), np.concatenate([np.arange(4, 22), np.arange(0, 2)])]       # second fold: other batches, no ragged tail


def _config1_inputs(lr):
    from multimodal_survival_prediction_amd import data
    cohort = data.make_cohort(n=88, dims=(32, 32, 32), rna_dim=5005, seed=88, complete=True)      # (the volumes are zeros: small grid)
    cohort["image"].zero_()
    tr, va = data.kfold_indices(88, 3, seed=42)[0]
    if lr != 0:
        tr = tr[:32]          # the envelope variant needs three epochs (fp32 oracle, fp64 oracle, HIP): the first 8 batches of the fold keep the GPU suite short
    return cohort, tr, va


def _fp64_config1_leg(lr, ref64, cohort, tr, va):
    from oracle import loops as OLP
    from multimodal_survival_prediction_amd import data
    cpu = torch.device("cpu")
    mk = lambda c, idx: data.BatchLoader(c, idx, 4, shuffle=False, style="simple")
    opt64 = torch.optim.AdamW(ref64.parameters(), lr=lr, weight_decay=1e-3)
    want64 = OLP.train_epoch_simple(ref64, _cast_loader(mk(cohort, tr), torch.float64), opt64, cpu)
    vw64 = OLP.validate_simple(ref64, _cast_loader(mk(cohort, va), torch.float64), cpu)
    return dict(want64=float(want64), vw64=[float(vw64[0]), float(vw64[1])])


def _cast_loader(loader, dtype):
    """The same batches with every floating tensor in `dtype` (the fp64 run of the oracle loops)."""
    for batch in loader:
        yield {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}


def _check_buffers(ref, net, tol=2e-3):
    for (k, b), (_, c) in zip(ref.named_buffers(), net.named_buffers()):
        if "num_batches" in k:
            assert int(b) == int(c), (k, int(b), int(c))       # counts the training-mode forwards (simple_fusion's second skip keeps one)
        else:
            assert_close(c, b, tol, k)


STYLES = {
    "final": ("MultiModalSurvivalNet", False, 1e-4),
    "partial": ("PartialModalityNet", False, 1e-4),
    "simple": ("SimpleFusionModel", True, 1e-3),
}


@pytest.mark.parametrize("style,lr", [("final", 0.0), ("partial", 0.0), ("simple", 0.0), ("partial", 1e-4), ("simple", 1e-4)])
def test_epoch_and_validate_match_oracle_loops(style, lr, eps=1e-8):
    from oracle import loops as OLP
    from multimodal_survival_prediction_amd import data, training
    from multimodal_survival_prediction_amd.training import FusedOptimizer
    cls, adamw, wd = STYLES[style]
    cohort = _cohort()
    if style == "final":                          # final_multimodal.py trains on complete, fully labelled patients
        cohort["has_survival"][:] = True
        lab = cohort["label"]
        lab[lab[:, 0] == 0, 0] = torch.arange(1, 1 + int((lab[:, 0] == 0).sum()), dtype=torch.float32) * 7.5
    dev_cohort = data.cohort_to(cohort, DEV)
    ref, net = _pair(cls, 11)
    tr_c, va_c, tr_d, va_d = _loaders(cohort, style, dev_cohort)
    opt_ref = (torch.optim.AdamW(ref.parameters(), lr=lr, weight_decay=wd, eps=eps) if adamw
               else torch.optim.Adam(ref.parameters(), lr=lr, weight_decay=wd, eps=eps))
    fo = FusedOptimizer(net, lr=lr, weight_decay=wd, adamw=adamw, eps=eps)
    cpu = torch.device("cpu")
    tol = 1e-4
    if lr != 0:
        import copy
        ref64 = copy.deepcopy(ref).double()                      # BEFORE the fp32 oracle steps: same initial weights
        _, net2 = _pair(cls, 11)                                 # a second HIP model, stepped batch by batch for the per-batch losses
    pb32 = []
    want = getattr(OLP, "train_epoch_" + style)(ref, tr_c, opt_ref, cpu, on_batch=lambda *v: pb32.append(v))
    got = getattr(training, "train_epoch_" + style)(net, tr_d, fo, DEV)
    st = fo.engine.epoch_stats()
    print(style, "train_epoch oracle", want, "hip", got, st)
    if lr != 0:
        fo2 = FusedOptimizer(net2, lr=lr, weight_decay=wd, adamw=adamw, eps=eps)
        pbh = []
        for batch in tr_d:                                       # one batch per call: the engine's state carries over, the call returns that batch's losses
            if style == "simple" and sum(bool(x) for x in batch["has_survival"]) < 2:
                continue                                         # (simple_fusion.py:257-258: never reaches the model)
            r = getattr(training, "train_epoch_" + style)(net2, [batch], fo2, DEV)
            if style == "simple" and fo2.engine.epoch_stats()["n_usable"] == 0:
                continue                                         # forward only (:267-268): no loss term
            pbh.append(r if isinstance(r, tuple) else (r,))
        gm, wm = [(v if isinstance(v, tuple) else (v,)) for v in (got, want)]
    if lr != 0:
        leg = _fp64_leg("epoch-%s-%g" % (style, lr), _fingerprint(ref64, [cohort["image"], cohort["rnaseq"], cohort["label"]], _hyper(style, lr, eps)),
                        lambda: _fp64_epoch_leg(style, lr, ref64, cohort, eps))
        pb64 = [tuple(v) for v in leg["pb64"]]
        want64 = tuple(leg["want64"]) if len(leg["want64"]) > 1 else leg["want64"][0]
        assert len(pbh) == len(pb32) == len(pb64), (len(pbh), len(pb32), len(pb64))
        env, worst = 0.0, 0.0
        for i, (h, a, x) in enumerate(zip(pbh, pb32, pb64)):
            env = max(env, max(abs(float(u) - float(v)) for u, v in zip(a, x)))
            for u, v, w_ in zip(h, x, a):
                print("  batch %d: fp64 %.7f | fp32 oracle %+.2e | HIP %+.2e | running max of the oracle's distance %.2e" % (i, v, w_ - v, u - v, env))
                worst = max(worst, abs(float(u) - float(v)) / (env + 1e-30) if env > 1e-5 else 0.0)
                assert abs(float(u) - float(v)) <= ENV_FACTOR * env + 1e-5 * max(1.0, abs(float(v))), (style, i, h, a, x)
        _record_ratio("test_epoch_and_validate_match_oracle_loops[%s]" % style, "per-batch losses, worst over the epoch", worst)
        xm = want64 if isinstance(want64, tuple) else (want64,)
        for u, w_, v in zip(gm, wm, xm):                         # the returned epoch means, same criterion
            assert abs(u - v) <= ENV_FACTOR * max(env, abs(w_ - v)) + 1e-5 * max(1.0, abs(v)), (style, got, want, want64)
    if style == "final":
        # :249-262: every batch counts in the mean, the no-event batch with loss 0 (and no update)
        assert st["n_batches"] == 6 and st["n_usable"] == 5
        assert lr != 0 or got == pytest.approx(want, rel=tol)
    elif style == "partial":
        # :401-428: 4 of 6 batches are Cox-usable (denominator n_usable); the entropy mean runs over all 6
        assert st["n_batches"] == 6 and st["n_usable"] == 4
        assert lr != 0 or (got[0] == pytest.approx(want[0], rel=tol) and got[1] == pytest.approx(want[1], rel=tol))
    else:
        # :257-268: batch 2 never reaches the engine, batch 1 runs the forward only
        assert st["n_batches"] == 5 and st["n_usable"] == 4
        assert lr != 0 or got == pytest.approx(want, rel=tol)
    if lr != 0:
        # BatchNorm running statistics after the epoch: within ENV_FACTOR x the fp32 oracle's own distance from the fp64 run (+ 1e-4 of the buffer's scale)
        b64 = dict(ref64.named_buffers())
        wb = 0.0
        for (k, b), (_, c) in zip(ref.named_buffers(), net.named_buffers()):
            if "num_batches" in k:
                assert int(b) == int(c), (k, int(b), int(c))
                continue
            x = b64[k]
            sc_ = float(x.abs().max()) + 1e-30
            e_hip, e_f32 = float((c.cpu().double() - x).abs().max()) / sc_, float((b.double() - x).abs().max()) / sc_
            assert e_hip <= ENV_FACTOR * e_f32 + 1e-4, (k, e_hip, e_f32)
            if e_f32 > 1e-4:
                wb = max(wb, e_hip / e_f32)
        _record_ratio("test_epoch_and_validate_match_oracle_loops[%s]" % style, "running statistics, worst buffer (oracle drift > 1e-4)", wb)
        return
    _check_buffers(ref, net, 1e-4)
    # validate: (avg_loss, c_index) with the reference's inclusion rules
    vw = getattr(OLP, "validate_" + style)(ref, va_c, cpu)
    vg = getattr(training, "validate_" + style)(net, va_d, DEV)
    print(style, "validate oracle", vw, "hip", vg)
    assert vg[0] == pytest.approx(vw[0], rel=1e-4)
    assert abs(vg[1] - vw[1]) <= 1e-6                 # identical pair counts (ConcordanceIndex returns an fp32 tensor)
    # hazards of the held-out patients after the epoch (running statistics of six training forwards)
    ref.eval(); net.eval()
    j = torch.arange(22, 26)
    with torch.no_grad():
        if style == "final":
            w = ref(cohort["image"][j], cohort["rnaseq"][j], cohort["clinical"][j])
            g = net(dev_cohort["image"][j], dev_cohort["rnaseq"][j], dev_cohort["clinical"][j])
        elif style == "partial":
            w = ref(cohort["image"][j], cohort["rnaseq"][j], cohort["clinical"][j], cohort["mask"][j])[0]
            g = net(dev_cohort["image"][j], dev_cohort["rnaseq"][j], dev_cohort["clinical"][j], dev_cohort["mask"][j])[0]
        else:
            w = ref(cohort["image"][j], cohort["rnaseq"][j])
            g = net(dev_cohort["image"][j], dev_cohort["rnaseq"][j])
    assert_close(g, w, 1e-4, "held-out hazards after one epoch")


@pytest.mark.parametrize("lr", [0.0, 1e-4])
def test_lockstep_epoch_matches_oracle_loops(lr):
    """The path the entry points and bench.py run: train_epoch_lockstep / validate_lockstep of a FoldGroupEngine (two fold models on
    the DEFAULT launch options, lazily named batches gathered on the GPU) -- each fold against the oracle's train_epoch_partial /
    validate_partial.  lr = 0: 1e-4 and identical pair counts.  lr = 1e-4 (the scripts' value): the returned epoch means and the
    BatchNorm running statistics within ENV_FACTOR x the fp32 oracle's own distance from its fp64 run (module docstring)."""
    import copy
    from oracle import loops as OLP
    from multimodal_survival_prediction_amd import data, training
    from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
    cohort = _cohort()
    dev_cohort = data.cohort_to(cohort, DEV)
    pairs = [_pair("PartialModalityNet", 21), _pair("PartialModalityNet", 22)]
    refs64 = [copy.deepcopy(p[0]).double() for p in pairs] if lr != 0 else None
    splits = LOCKSTEP_SPLITS
    group = FoldGroupEngine([p[1] for p in pairs], lr=lr, weight_decay=1e-4, adamw=False, gate_entropy_weight=0.01)
    tl = [data.BatchLoader(dev_cohort, s, 4, shuffle=False, lazy=True, with_valid=True) for s in splits]
    vl = [data.BatchLoader(dev_cohort, np.arange(22, cohort["n"]), 4, shuffle=False) for _ in splits]
    got = training.train_epoch_lockstep(group, tl, "partial")
    vgot = training.validate_lockstep(group, vl, "partial", DEV)
    for f, ((ref, net), s) in enumerate(zip(pairs, splits)):
        opt = torch.optim.Adam(ref.parameters(), lr=lr, weight_decay=1e-4)
        want = OLP.train_epoch_partial(ref, data.BatchLoader(cohort, s, 4, shuffle=False), opt, torch.device("cpu"))
        print("fold", f, "oracle", want, "hip", got[f], vgot[f])
        if lr == 0:
            vwant = OLP.validate_partial(ref, data.BatchLoader(cohort, np.arange(22, cohort["n"]), 4, shuffle=False), torch.device("cpu"))
            assert got[f][0] == pytest.approx(want[0], rel=1e-4) and got[f][1] == pytest.approx(want[1], rel=1e-4)
            _check_buffers(ref, net, 1e-4)
            assert vgot[f][0] == pytest.approx(vwant[0], rel=1e-4) and abs(vgot[f][1] - vwant[1]) <= 1e-6
            continue
        leg = _fp64_leg("lockstep-partial-f%d-%g" % (f, lr),
                        _fingerprint(refs64[f], [cohort["image"], cohort["rnaseq"], cohort["label"], torch.as_tensor(s)], _hyper("partial", lr)),
                        lambda: _fp64_epoch_leg("partial", lr, refs64[f], cohort, split=s, buffers=True))
        worst = 0.0
        for u, w_, v in zip(got[f], want, leg["want64"]):
            print("  fold %d mean: fp64 %.7f | fp32 oracle %+.2e | HIP %+.2e" % (f, v, w_ - v, u - v))
            assert abs(u - v) <= ENV_FACTOR * abs(w_ - v) + 1e-5 * max(1.0, abs(v)), (f, got[f], want, leg["want64"])
            if abs(w_ - v) > 1e-5:
                worst = max(worst, abs(u - v) / abs(w_ - v))
        _record_ratio("test_lockstep_epoch_matches_oracle_loops", "fold %d epoch means" % f, worst)
        wb = 0.0
        for (k, b), (_, c) in zip(ref.named_buffers(), net.named_buffers()):
            if "num_batches" in k:
                assert int(b) == int(c), (k, int(b), int(c))
            elif k in leg["buf64"]:
                x = torch.tensor(leg["buf64"][k], dtype=torch.float64).reshape(b.shape)
                sc_ = float(x.abs().max()) + 1e-30
                e_hip, e_f32 = float((c.cpu().double() - x).abs().max()) / sc_, float((b.double() - x).abs().max()) / sc_
                assert e_hip <= ENV_FACTOR * e_f32 + 1e-4, (k, e_hip, e_f32)
                if e_f32 > 1e-4:
                    wb = max(wb, e_hip / e_f32)
        _record_ratio("test_lockstep_epoch_matches_oracle_loops", "fold %d running statistics (heads, norm0, norm5)" % f, wb)


def test_lockstep_epoch_from_pinned_host_cohort_matches_device_cohort():
    """The host-resident data path (data.cohort_pin: the gather launch reads the batch rows over PCIe and zero-fills the rows of
    missing modalities instead of reading them) must assemble exactly the batches of the device-resident path: same models, same
    splits (ragged tail, unlabelled patients, missing CT / RNA-seq), one lock-step epoch at lr = 0 (so that run-to-run summation
    order of the weight-gradient atomics cannot enter): returned losses and BatchNorm running statistics agree to 1e-6."""
    from multimodal_survival_prediction_amd import data, training
    from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
    cohort = _cohort()
    assert float((cohort["mask"][:, 0] == 0).sum()) > 0 and float((cohort["mask"][:, 1] == 0).sum()) > 0      # the zero-fill branch is exercised
    splits = LOCKSTEP_SPLITS
    outs = []
    for place in (lambda c: data.cohort_to(c, DEV), data.cohort_pin):
        src = place(dict(cohort))
        pairs = [_pair("PartialModalityNet", 21), _pair("PartialModalityNet", 22)]
        group = FoldGroupEngine([p[1] for p in pairs], lr=0.0, weight_decay=1e-4, adamw=False, gate_entropy_weight=0.01)
        tl = [data.BatchLoader(src, s, 4, shuffle=False, lazy=True, with_valid=True) for s in splits]
        got = training.train_epoch_lockstep(group, tl, "partial")
        torch.cuda.synchronize()
        outs.append((got, [[q.detach().clone() for q in p[1].buffers() if q.dtype == torch.float32] for p in pairs]))
    (la, ba), (lb, bb) = outs
    for x, y in zip(la, lb):
        assert x[0] == pytest.approx(y[0], rel=1e-6) and x[1] == pytest.approx(y[1], rel=1e-6), (la, lb)
    for ma, mb in zip(ba, bb):
        for a, b in zip(ma, mb):
            assert_close(b, a, 1e-6, "running statistics")


@pytest.mark.parametrize("style,cls", [("partial", "PartialModalityNet"), ("final", "MultiModalSurvivalNet"), ("simple", "SimpleFusionModel")])
def test_validate_lockstep_named_batches_match_eager_path(style, cls):
    """validate_lockstep over lazily NAMED batches (one gather + one graph per lock-step position, Cox values and accumulators on the
    device, one host sync at the end) against its eager path (batches materialised, per-batch losses through losses.cox_loss with a
    host sync each -- the path compared with oracle/loops.py above): (avg_loss, c_index) of two fold models on a validation split with
    an unlabelled patient, a batch without events, a single-labelled batch and a ragged tail."""
    from multimodal_survival_prediction_amd import data, training
    from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
    cohort = _cohort()
    dev_cohort = data.cohort_to(cohort, DEV)
    pairs = [_pair(cls, 41), _pair(cls, 42)]
    group = FoldGroupEngine([p[1] for p in pairs], lr=0.0, weight_decay=1e-4, adamw=False, gate_entropy_weight=0.01)
    splits = [np.arange(22, cohort["n"]), np.arange(20, cohort["n"] - 1)]
    wv = style != "final"
    eager = training.validate_lockstep(group, [data.BatchLoader(dev_cohort, s, 4, shuffle=False, style=style) for s in splits], style, DEV)
    named = training.validate_lockstep(group, [data.BatchLoader(dev_cohort, s, 4, shuffle=False, lazy=True, with_valid=wv) for s in splits],
                                       style, DEV)
    again = training.validate_lockstep(group, [data.BatchLoader(dev_cohort, s, 4, shuffle=False, lazy=True, with_valid=wv) for s in splits],
                                       style, DEV)          # replayed graphs, accumulators reset
    print(style, "eager", eager, "named", named)
    for e, n, a in zip(eager, named, again):
        assert n[0] == pytest.approx(e[0], rel=1e-5, abs=1e-7) and abs(n[1] - e[1]) <= 1e-6
        assert a[0] == pytest.approx(n[0], rel=1e-6, abs=1e-7) and abs(a[1] - n[1]) <= 1e-6


def test_validate_lockstep_sub_group_streams():
    """Four fold models: validate_lockstep over named batches as one group and as 2 + 2 sub-groups on two HIP streams give the same
    (avg_loss, c_index) -- different ragged tails per fold, so the sub-groups' positions and plans differ."""
    from multimodal_survival_prediction_amd import data, training
    from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
    cohort = _cohort()
    dev_cohort = data.cohort_to(cohort, DEV)
    pairs = [_pair("PartialModalityNet", 50 + i) for i in range(4)]
    group = FoldGroupEngine([p[1] for p in pairs], lr=0.0, weight_decay=1e-4, adamw=False, gate_entropy_weight=0.01)
    n = cohort["n"]
    splits = [np.arange(22, n), np.arange(20, n - 1), np.arange(18, n), np.arange(21, n - 2)]
    mk = lambda: [data.BatchLoader(dev_cohort, s, 4, shuffle=False, lazy=True, with_valid=True) for s in splits]
    one = training.validate_lockstep(group, mk(), "partial", DEV)
    two = training.validate_lockstep(group, mk(), "partial", DEV, concurrent=2)
    for a, b in zip(one, two):
        assert b[0] == pytest.approx(a[0], rel=1e-6, abs=1e-7) and abs(b[1] - a[1]) <= 1e-6, (one, two)


@pytest.mark.parametrize("lr", [0.0, 1e-4])
def test_config1_simple_fusion_ct_stubbed(lr):
    """BASELINE config 1 at its own width: simple_fusion.py, 88 synthetic complete patients, RNA-seq 5005-d, CT encoder input stubbed to
    zero volumes (the RNA-seq heads do the work), batch 4, fold 1 of 3 -- one epoch of the HIP train_epoch + validate against the oracle
    loops (dropout off).  lr = 0: train mean and validation loss at 1e-4, the C-index from IDENTICAL pair counts.  lr = 1e-4 (the
    script's value): within ENV_FACTOR x the fp32 oracle's own distance from its fp64 run + 1e-4 (see the module docstring)."""
    import copy
    from oracle import loops as OLP
    from multimodal_survival_prediction_amd import data, training
    from multimodal_survival_prediction_amd.training import FusedOptimizer
    cohort, tr, va = _config1_inputs(lr)
    dev_cohort = data.cohort_to(cohort, DEV)
    ref, net = _pair("SimpleFusionModel", 31, rna_dim=5005)
    ref64 = copy.deepcopy(ref).double()
    cpu = torch.device("cpu")
    mk = lambda c, idx: data.BatchLoader(c, idx, 4, shuffle=False, style="simple")
    opt = torch.optim.AdamW(ref.parameters(), lr=lr, weight_decay=1e-3)
    fo = FusedOptimizer(net, lr=lr, weight_decay=1e-3, adamw=True)
    want = OLP.train_epoch_simple(ref, mk(cohort, tr), opt, cpu)
    got = training.train_epoch_simple(net, mk(dev_cohort, tr), fo, DEV)
    vw = OLP.validate_simple(ref, mk(cohort, va), cpu)
    vg = training.validate_simple(net, mk(dev_cohort, va), DEV)
    print("config 1 lr", lr, ": train", want, got, "validate", vw, vg)
    if lr == 0:
        assert got == pytest.approx(want, rel=1e-4)
        assert vg[0] == pytest.approx(vw[0], rel=1e-4) and abs(vg[1] - vw[1]) <= 1e-6          # identical concordant / discordant pair counts
        return
    leg = _fp64_leg("config1-%g" % lr, _fingerprint(ref64, [cohort["rnaseq"], cohort["label"], torch.as_tensor(tr)], _hyper("simple", lr, extra="config1")),
                    lambda: _fp64_config1_leg(lr, ref64, cohort, tr, va))
    want64, vw64 = leg["want64"], tuple(leg["vw64"])
    print("config 1 fp64: train", want64, "validate", vw64)
    for u, w_, v in ((got, want, want64), (vg[0], vw[0], vw64[0])):
        print("  fp64 %.7f | fp32 oracle %+.2e | HIP %+.2e" % (v, w_ - v, u - v))
        assert abs(u - v) <= ENV_FACTOR * abs(w_ - v) + 1e-4 * max(1.0, abs(v)), (u, w_, v)
        if abs(w_ - v) > 1e-5:
            _record_ratio("test_config1_simple_fusion_ct_stubbed", "train mean / validation loss", abs(u - v) / abs(w_ - v))
    n_pairs = len(va) * (len(va) - 1) / 2
    assert abs(vg[1] - vw64[1]) <= 2.0 * abs(vw[1] - vw64[1]) + 4.0 / n_pairs, (vg, vw, vw64)      # <= 4 pairs beyond the oracle's own flips
