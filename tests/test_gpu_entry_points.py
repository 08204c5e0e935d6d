"""The three entry points the reference's users run (SURVEY.md section 8b "Entry points to keep"): scripts/training/final_multimodal.py,
partial_modality_training.py and simple_fusion.py executed as `python <script>` from a scratch cwd on a small synthetic cohort (42 patients: no
training split ends in a batch of ONE patient, which BatchNorm1d rejects in the reference and here alike)
(R/scripts/training/final_multimodal.py:316-417, partial_modality_training.py:496-607, simple_fusion.py:369-451):
  * results/<name>/cv_results.json carries the reference's keys (final_multimodal.py:403-417, simple_fusion.py:444-451);
  * the lock-step K-fold driver (scripts/training/_common.py::cv_lockstep: scheduler per fold, best-checkpoint save, early stopping,
    the group shrinking as folds stop) -- with patience 1 at least one fold stops before the last epoch;
  * the saved best checkpoints load into the ORACLE class (same state_dict keys as the reference's modules) and give the hazards
    the HIP model gives, eval mode, 1e-4;
  * MMS_LOCKSTEP=0 (fold after fold, as the reference trains) gives the same per-fold best C-index and stopping epoch (compared at lr = 0)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The reference's 64x64x32 volumes: on 32^3 the last dense block has ONE voxel per sample and its training-mode BatchNorm runs over the
# batch's 4 values -- so ill-conditioned that two arithmetically equivalent launch orders (lock-step group vs fold after fold) drift apart.
ENV = dict(MMS_PATIENTS="42", MMS_EPOCHS="4", MMS_FOLDS="3", MMS_PATIENCE="1", MMS_BATCH_SIZE="4", MMS_VOLUME="64,64,32")


def _run(script, cwd, **extra):
    env = dict(os.environ, **ENV, **extra)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "training", script)], cwd=cwd, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


def _hazards_match(ckpt, cls, cohort_kw, forward):
    """checkpoint -> oracle class on the CPU and HIP class on the GPU, eval mode: same hazards (1e-4)."""
    from oracle import models as OM
    from multimodal_survival_prediction_amd import data, models as HM
    sd = torch.load(ckpt, map_location="cpu")
    ref = getattr(OM, cls)(rna_dim=5005, use_monai=True)
    missing = ref.load_state_dict(sd, strict=True)
    net = getattr(HM, cls)(rna_dim=5005)
    net.load_state_dict(sd)
    ref.eval(); net.to("cuda:0").eval()
    c = data.make_cohort(**cohort_kw)
    j = torch.arange(0, 6)
    with torch.no_grad():
        w = forward(ref, {k: (v[j] if torch.is_tensor(v) and v.shape[:1] == (c["n"],) else v) for k, v in c.items()}, "cpu")
        g = forward(net, {k: (v[j] if torch.is_tensor(v) and v.shape[:1] == (c["n"],) else v) for k, v in c.items()}, "cuda:0").cpu()
    err = float((g - w).abs().max() / (w.abs().max() + 1e-30))
    assert err <= 1e-4, (ckpt, err)


def test_partial_modality_training_entry_point(tmp_path):
    out = _run("partial_modality_training.py", tmp_path)
    res = json.load(open(tmp_path / "results" / "partial_modality" / "cv_results.json"))
    assert set(res) >= {"model", "c_index_mean", "c_index_std", "fold_results", "hyperparameters"}          # partial_modality_training.py:592-607
    assert set(res["hyperparameters"]) >= {"batch_size", "learning_rate", "epochs", "n_folds", "gate_entropy_weight"}
    folds = res["fold_results"]
    assert [r["fold"] for r in folds] == [1, 2, 3] and all({"best_c_index", "train_size", "val_size"} <= set(r) for r in folds)
    assert all(0.0 <= r["best_c_index"] <= 1.0 for r in folds)
    assert res["c_index_mean"] == pytest.approx(sum(r["best_c_index"] for r in folds) / 3, abs=1e-9)
    # early stopping (patience 1): some fold stopped before the last epoch, i.e. the lock-step group went on without it
    assert min(r["epochs_run"] for r in folds) < 4 <= max(4, max(r["epochs_run"] for r in folds)), folds
    for k in (1, 2, 3):
        assert os.path.exists(tmp_path / "models" / "partial_modality" / f"fold_{k}_best.pth")
    _hazards_match(tmp_path / "models" / "partial_modality" / "fold_1_best.pth", "PartialModalityNet",
                   dict(n=42, dims=(64, 64, 32), seed=608, complete=False),
                   lambda m, c, d: m(c["image"].to(d), c["rnaseq"].to(d), c["clinical"].to(d), c["mask"].to(d))[0])
    # Fold after fold (the reference's order, MMS_LOCKSTEP=0) against the lock-step group.  Compared at lr = 0: with the scripts' lr the
    # two runs' weights part ways at the 1e-3 level within a few Adam steps (lr * sign(g) on gradient entries that are rounding noise,
    # tests/test_gpu_epoch_parity.py) and a C-index over 8 validation patients then differs by whole pairs; with frozen weights the
    # driver logic under test -- per-fold loaders and shuffles, BatchNorm running statistics moving with every training forward, the
    # scheduler, best-checkpoint rule, early stopping and the group shrinking -- must give the same per-fold numbers in both orders.
    runs = []
    for mode in ("1", "0"):
        d = tmp_path / ("lr0_lockstep" + mode)
        d.mkdir()
        _run("partial_modality_training.py", d, MMS_LOCKSTEP=mode, MMS_LR="0")
        runs.append(json.load(open(d / "results" / "partial_modality" / "cv_results.json"))["fold_results"])
    for a, b in zip(*runs):
        assert a["best_c_index"] == pytest.approx(b["best_c_index"], abs=1e-6) and a["epochs_run"] == b["epochs_run"], runs


def test_final_multimodal_entry_point(tmp_path):
    _run("final_multimodal.py", tmp_path)
    res = json.load(open(tmp_path / "results" / "final" / "cv_results.json"))
    assert set(res) >= {"model", "c_index_mean", "c_index_std", "fold_results", "hyperparameters"}          # final_multimodal.py:403-417
    assert set(res["hyperparameters"]) >= {"batch_size", "learning_rate", "epochs", "n_folds"}
    assert [r["fold"] for r in res["fold_results"]] == [1, 2, 3]
    assert all(0.0 <= r["best_c_index"] <= 1.0 for r in res["fold_results"])
    _hazards_match(tmp_path / "models" / "final" / "fold_2_best.pth", "MultiModalSurvivalNet", dict(n=42, dims=(64, 64, 32), seed=608, complete=True),
                   lambda m, c, d: m(c["image"].to(d), c["rnaseq"].to(d), c["clinical"].to(d)))


def test_simple_fusion_entry_point(tmp_path):
    _run("simple_fusion.py", tmp_path)
    res = json.load(open(tmp_path / "results" / "simple_fusion" / "cv_results.json"))
    assert set(res) >= {"model", "n_folds", "num_epochs", "c_index_mean", "c_index_std", "fold_results"}       # simple_fusion.py:444-451
    assert res["n_folds"] == 3 and res["num_epochs"] == 4
    for r in res["fold_results"]:
        assert {"fold", "best_c_index", "best_epoch", "train_size", "val_size"} <= set(r) and 1 <= r["best_epoch"] <= 4
    _hazards_match(tmp_path / "results" / "simple_fusion" / "best_model_fold1.pth", "SimpleFusionModel", dict(n=42, dims=(64, 64, 32), seed=88, complete=True),
                   lambda m, c, d: m(c["image"].to(d), c["rnaseq"].to(d)))
