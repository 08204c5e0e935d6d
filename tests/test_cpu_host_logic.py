"""CPU tests of the host logic: synthetic cohort contract, fold sharding / result gathering / flat-gradient averaging
over a 2-rank gloo group, scheduler restatements, and the no-fallback rule."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cohort_contract():
    from multimodal_survival_prediction_amd import data
    c = data.make_cohort(n=40, dims=(32, 32, 32), rna_dim=64, seed=1, complete=False)
    assert c["image"].shape == (40, 1, 32, 32, 32) and c["image"].dtype == torch.float32
    assert float(c["image"].min()) >= 0 and float(c["image"].max()) <= 1
    assert c["label"].shape == (40, 2) and c["mask"].shape == (40, 3)
    m = c["mask"].bool()
    assert float(c["image"][~m[:, 0]].abs().max()) == 0            # missing imaging -> zeros (partial...:90)
    assert float(c["rnaseq"][~m[:, 1]].abs().max()) == 0
    t = c["label"][c["has_survival"], 0].numpy()
    assert len(np.unique(t)) == len(t)                              # distinct times: no Cox ties
    c2 = data.make_cohort(n=40, dims=(32, 32, 32), rna_dim=64, seed=1, complete=False)
    assert torch.equal(c["image"], c2["image"]) and torch.equal(c["label"], c2["label"])
    full = data.make_cohort(n=20, dims=(32, 32, 32), rna_dim=8, seed=2)
    b = next(iter(data.BatchLoader(full, np.arange(10), 4, style="simple")))
    assert b["time"].shape == (4, 1) and b["event"].dtype == torch.int64 and len(b["has_survival"]) == 4
    folds = data.kfold_indices(109, 5)
    assert sorted(np.concatenate([v for _, v in folds]).tolist()) == list(range(109))
    assert [len(tr) for tr, _ in folds] == [87, 87, 87, 87, 88]


def test_fold_assignment():
    from multimodal_survival_prediction_amd.distributed import folds_of_rank
    assert [folds_of_rank(5, 4, r) for r in range(4)] == [[0, 4], [1], [2], [3]]
    assert folds_of_rank(5, 1, 0) == [0, 1, 2, 3, 4]
    assert sorted(sum((folds_of_rank(5, 8, r) for r in range(8)), [])) == [0, 1, 2, 3, 4]


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ["MMS_ROOT"])
from multimodal_survival_prediction_amd import distributed as D
world, rank, local = D.init("gloo")
assert world == 2
res = D.gather_fold_results([{"fold": k + 1, "best_c_index": 0.5 + 0.01 * k, "rank": rank} for k in D.folds_of_rank(5, world, rank)], world)
assert [r["fold"] for r in res] == [1, 2, 3, 4, 5] and [r["rank"] for r in res] == [0, 1, 0, 1, 0]
g = torch.full((1000,), float(rank + 1))
D.allreduce_mean_(g, world)
assert torch.allclose(g, torch.full((1000,), 1.5))
assert D.max_over_ranks(float(rank), "cpu") == 1.0
# the exchange of the global Cox risk set: rank r's B values land at [r*B, (r+1)*B) on every rank; gradients are summed
out = torch.zeros(2 * 3)
D.all_gather_into(out, torch.arange(3.0) + 10 * rank, world)
assert out.tolist() == [0, 1, 2, 10, 11, 12]
s = torch.full((8,), float(rank + 1))
D.allreduce_sum_(s, world)
assert torch.allclose(s, torch.full((8,), 3.0))
# the staged data-parallel step's exchange: buckets launched asynchronously (async_op=True), waited for later -- CPU tensors take the
# same branch RCCL takes on the GPU (no host detour); every bucket's ranges are summed, nothing outside them is touched
flat = torch.arange(100.0) * (rank + 1)
w1 = D.allreduce_ranges_async(flat, [(0, 10), (50, 20)], world)
w2 = D.allreduce_ranges_async(flat, [(90, 10)], world)
w2(); w1()
want = torch.arange(100.0) * (rank + 1)
for a, n in [(0, 10), (50, 20), (90, 10)]:
    want[a:a + n] = torch.arange(100.0)[a:a + n] * 3
assert torch.equal(flat, want), (flat, want)
D.barrier()
print("ok", rank)
'''


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MMS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"ok {r}" in o, o


def test_gradient_buckets_cover_flat_buffer_once():
    """engine._buckets / distributed.gradient_buckets: heads first, then the DenseNet121 backward stages 3..0; contiguous ranges that
    cover the parameter-ordered flat buffer exactly once, each stage = the parameter tensors of its dense block (+ transition / stem / norm5)."""
    from multimodal_survival_prediction_amd import distributed as D, models
    m = models.MultiModalSurvivalNet(rna_dim=16)
    params = list(m.parameters())
    enc = list(m.ct_encoder.parameters())
    assert len(enc) == D.ENC_STAGE_CUTS[-1]
    buckets = D.gradient_buckets(params, enc, True)
    assert len(buckets) == 5
    n = sum(p.numel() for p in params)
    cover = np.zeros(n, dtype=np.int32)
    for b in buckets:
        for a, k in b:
            cover[a:a + k] += 1
    assert cover.min() == 1 and cover.max() == 1
    # finalisation order: stage 3 = denseblock4 + norm5 + class_layers (+ transition3), ..., stage 0 = stem + denseblock1
    names = [k for k, _ in m.ct_encoder.named_parameters()]
    offs = np.cumsum([0] + [p.numel() for p in params])
    start = {id(p): offs[i] for i, p in enumerate(params)}
    c = D.ENC_STAGE_CUTS
    for k, blk in enumerate((3, 2, 1, 0)):
        lo = start[id(enc[c[blk]])]
        assert buckets[1 + k][0][0] == lo
        assert sum(x for _, x in buckets[1 + k]) == sum(p.numel() for p in enc[c[blk]:c[blk + 1]])
    assert names[c[3]].startswith("features.denseblock4") or names[c[3]].startswith("features.transition3")
    assert names[0] == "features.conv0.weight" and names[c[1]].startswith(("features.denseblock2", "features.transition1"))
    # single bucket for an encoder-less / fallback model
    one = D.gradient_buckets(params, [], False)
    assert one == [[(0, n)]]


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher starts N ranks itself (before touching a GPU) and never reports fewer GPUs than asked."""
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=120)
    cmd = __import__("json").loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert r.returncode == 0 and "torch.distributed.run" in cmd and "--nproc-per-node=2" in cmd and cmd[-2:] == ["--gpus", "2"]
    env = dict(os.environ, MMS_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--rendezvous-check"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert __import__("json").loads(line) == {"rendezvous": 2, "ranks": [0, 1], "backend": "gloo"}
    if torch.cuda.device_count() < 8:      # a request that cannot be met is refused, not downgraded to one GPU
        r = subprocess.run([sys.executable, bench, "--gpus", "8"], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode != 0 and "n_gpus" not in r.stdout


def test_schedulers_match_torch():
    from multimodal_survival_prediction_amd.training import CosineAnnealingLR, ReduceLROnPlateau

    class Opt:
        def __init__(self):
            self.param_groups = [dict(lr=1e-4)]

        def set_lr(self, lr):
            self.param_groups[0]["lr"] = lr
    p = torch.nn.Parameter(torch.zeros(1))
    ref = torch.optim.SGD([p], lr=1e-4)
    rs, mine_o = torch.optim.lr_scheduler.CosineAnnealingLR(ref, T_max=50), Opt()
    ms = CosineAnnealingLR(mine_o, T_max=50)
    for _ in range(50):
        ref.step(); rs.step(); ms.step()
        assert mine_o.param_groups[0]["lr"] == pytest.approx(ref.param_groups[0]["lr"], rel=1e-9, abs=1e-15)
    ref = torch.optim.SGD([p], lr=1e-4)
    rs, mine_o = torch.optim.lr_scheduler.ReduceLROnPlateau(ref, mode="max", factor=0.5, patience=5), Opt()
    ms = ReduceLROnPlateau(mine_o, mode="max", factor=0.5, patience=5)
    for v in [0.5, 0.55, 0.54, 0.53, 0.55, 0.551, 0.52, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.6, 0.59] + [0.58] * 14:
        rs.step(v); ms.step(v)
        assert mine_o.param_groups[0]["lr"] == pytest.approx(ref.param_groups[0]["lr"], rel=1e-12)


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU rather than fall back to torch/oracle math."""
    from multimodal_survival_prediction_amd import losses, models
    m = models.MultiModalSurvivalNet(rna_dim=16)
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 1, 32, 32, 32), torch.zeros(2, 16), torch.zeros(2, 1))
    with pytest.raises(RuntimeError):
        losses.cox_loss(torch.zeros(4), torch.ones(4), torch.arange(4.0))
    import multimodal_survival_prediction_amd as pkg
    src = "".join(open(os.path.join(os.path.dirname(pkg.__file__), f)).read()
                  for f in os.listdir(os.path.dirname(pkg.__file__)) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_cohort_contract_on_disk(tmp_path):
    """write_cohort lays down the reference's data/processed contract (create_full_matching_table.py:124-134 columns;
    patient-indexed rnaseq CSV); read_tables validates it.  Host side only (the volume preprocessing needs the GPU)."""
    import numpy as np
    from multimodal_survival_prediction_amd import cohort_io, data
    c = data.make_cohort(n=9, dims=(32, 32, 32), rna_dim=12, seed=2, complete=False, counts=dict(n=9, imaging=5, rnaseq=6, clinical=8, survival=4))
    cohort_io.write_cohort(str(tmp_path), c, native_dims=(10, 12, 9))
    mt, rn = cohort_io.read_tables(str(tmp_path))
    assert list(mt.columns) == cohort_io.COLUMNS and len(mt) == 9
    assert int(mt["has_imaging"].sum()) == 5 and int(mt["has_rnaseq"].sum()) == 6 and int(mt["has_survival"].sum()) == 4
    assert rn.shape == (6, 12) and set(rn.index) == set(mt.loc[mt["has_rnaseq"], "patient_id"])
    for flag, path in zip(mt["has_imaging"], mt["nifti_path"]):
        assert isinstance(path, str) == bool(flag)
        if flag:
            assert np.load(path).shape == (10, 12, 9)
    assert mt.loc[~mt["has_survival"], "survival_time"].isna().all()
    ages = mt.loc[mt["has_clinical"], "age"].to_numpy()
    assert ((ages >= 30) & (ages <= 90)).all()


def test_final_comparison_collects_and_tests(tmp_path):
    """scripts/training/final_comparison.py: cv_results.json collection, best model, paired t-test, results.json schema."""
    import importlib.util, json, os
    spec = importlib.util.spec_from_file_location("final_comparison", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                   "scripts", "training", "final_comparison.py"))
    fc = importlib.util.module_from_spec(spec); spec.loader.exec_module(fc)
    vals = {"results/rnaseq_only": [0.70, 0.66, 0.72], "results/simple_fusion": [0.61, 0.60, 0.65], "results/final": [0.5, 0.6, 0.55, 0.58, 0.52]}
    for d, v in vals.items():
        os.makedirs(tmp_path / d)
        json.dump({"c_index_mean": float(sum(v) / len(v)), "c_index_std": 0.02,
                   "fold_results": [{"fold": i + 1, "best_c_index": x, "val_size": 10} for i, x in enumerate(v)]}, open(tmp_path / d / "cv_results.json", "w"))
    out = fc.main(str(tmp_path))
    assert out["best_model"]["name"] == "RNA-Only"
    assert set(out["paired_t_tests"]) == {"Simple Fusion"}            # the 5-fold model has a different fold count: skipped (:74)
    assert out["paired_t_tests"]["Simple Fusion"]["p"] < 0.05
    assert json.load(open(tmp_path / "results" / "final_comparison" / "results.json"))["model_results"]["Final Multimodal"]["n_patients"] == 50


def test_lockstep_iteration_and_batch_rules():
    """Host logic of the lock-step K-fold driver: folds drop out as their loaders run dry; per-style batch keyword rules."""
    from multimodal_survival_prediction_amd import training as T
    loaders = [["a0", "a1", "a2"], ["b0"], ["c0", "c1"]]
    seen = list(T._lockstep(loaders, (4, 7, 9)))
    assert seen == [{4: "a0", 7: "b0", 9: "c0"}, {4: "a1", 9: "c1"}, {4: "a2"}]
    b = dict(image=torch.zeros(3, 1, 32, 32, 32), rnaseq=torch.zeros(3, 8), clinical=torch.zeros(3, 1), label=torch.ones(3, 2),
             mask=torch.ones(3, 3), has_survival=[True, False, True], time=torch.ones(3, 1), event=torch.ones(3, 1, dtype=torch.long))
    assert set(T._train_kwargs("final", b)) == {"ct", "rna", "clinical", "time", "event"}
    kw = T._train_kwargs("partial", b)
    assert kw["valid"].tolist() == [1.0, 0.0, 1.0] and kw["mask"].shape == (3, 3)
    assert T._train_kwargs("flexible", b)["mask"].shape == (3, 2)
    assert set(T._train_kwargs("rnaseq", b)) == {"rna", "time", "event"}
    b["has_survival"] = [True, False, False]
    assert T._train_kwargs("simple", b) is None and T._train_kwargs("flexible", b) is None     # < 2 labelled: skipped before the forward
    assert T._SKIP_UNUSABLE == {"final": True, "partial": False, "simple": True, "flexible": True, "rnaseq": False}


def test_subgroup_sizes_rule(monkeypatch):
    """training.subgroup_sizes: near-equal contiguous sub-groups, one per stream at most, a single group on one stream / for one member
    (the rule bench.py's labels and the roofline leg share with train_epoch_lockstep)."""
    from multimodal_survival_prediction_amd.training import subgroup_sizes as s
    monkeypatch.delenv("MMS_MIN_SPLIT_MEMBERS", raising=False)
    assert s(5, 3) == (2, 2, 1) and s(5, 2) == (3, 2) and s(5, 1) == (5,)
    assert s(3, 3) == (1, 1, 1) and s(3, 2) == (2, 1) and s(2, 3) == (1, 1) and s(1, 3) == (1,)
    assert s(10, 2) == (5, 5) and s(4, 3) == (2, 1, 1)
    for n in range(1, 11):
        for c in range(1, 5):
            assert sum(s(n, c)) == n and len(s(n, c)) <= c and max(s(n, c)) - min(s(n, c)) <= 1
    monkeypatch.setenv("MMS_MIN_SPLIT_MEMBERS", "4")          # rounds 1-2: fewer than four members stay one group
    assert s(3, 3) == (3,) and s(4, 3) == (2, 1, 1)


def test_batch_of_one_rule_and_launch_options():
    """The batch-of-one rule (torch's BatchNorm1d error) and the host side of the launch-shape options: ops.dn_opts builds the MmsDnOpts
    block of include/mmsurv.h (unknown field names are refused, not silently dropped); nothing in the package reads a launch-shape
    environment variable any more."""
    from multimodal_survival_prediction_amd import ops
    from multimodal_survival_prediction_amd.engine import check_train_batch
    o = ops.dn_opts(dict(split_wgs=512), persist_b4=-1, out_features=64)
    assert (o.split_wgs, o.persist_b4, o.out_features, o.conv3_small) == (512, -1, 64, 0)
    o2 = ops.dn_opts(o, conv3_small=2)
    assert (o2.split_wgs, o2.persist_b4, o2.conv3_small) == (512, -1, 2) and o.conv3_small == 0
    with pytest.raises(AttributeError):
        ops.dn_opts(no_such_option=1)
    assert ops.opts_ref(None) is None
    import glob
    src = "".join(open(p).read() for p in glob.glob(os.path.join(ROOT, "multimodal_survival_prediction_amd", "csrc", "*.h*")))
    assert "getenv(\"MMS_DEBUG_SKIP\")" in src and src.count("getenv(") == 1        # the one read sits behind #ifdef MMS_ABLATE_STEP (ablation builds)
    check_train_batch(2); check_train_batch(1, bn_world=2)
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        check_train_batch(1)
