#!/usr/bin/env python3
"""Collect the cv_results.json files of the entry points into one comparison -- the tables/t-tests part of the reference's
scripts/training/final_comparison.py (:31-60 result collection, :66-84 paired t-tests against the best model, :90-112 dataset
info, :353-373 results/final_comparison/results.json schema).  The figures (matplotlib/seaborn) are out of scope.
CPU only; no kernels.  Run from the directory that holds results/ (and, optionally, data/processed/full_matching_table.csv)."""
import json
import os

import numpy as np

RESULTS_FILES = {          # display name -> file written by scripts/training/<entry point>.py
    "RNA-Only": "results/rnaseq_only/cv_results.json",
    "Partial Modality": "results/partial_modality/cv_results.json",
    "Simple Fusion": "results/simple_fusion/cv_results.json",
    "Flexible Multimodal": "results/flexible_multimodal/cv_results.json",
    "Final Multimodal": "results/final/cv_results.json",
}


def collect(root="."):
    out = {}
    for name, rel in RESULTS_FILES.items():
        p = os.path.join(root, rel)
        if not os.path.exists(p):
            continue
        d = json.load(open(p))
        folds = [f["best_c_index"] for f in d["fold_results"]]
        out[name] = dict(mean=float(d.get("c_index_mean", np.mean(folds))), std=float(d.get("c_index_std", np.std(folds))),
                         fold_values=[float(x) for x in folds], n_patients=sum(f.get("val_size", 0) for f in d["fold_results"]) or None)
    return out


def compare(all_results):
    """-> (best model name, {other: dict(delta, t, p, sig)}) with scipy.stats.ttest_rel on equal fold counts (:74-84)."""
    from scipy import stats
    best = max(all_results.items(), key=lambda kv: kv[1]["mean"])
    tests = {}
    for name, r in all_results.items():
        if name == best[0] or len(r["fold_values"]) != len(best[1]["fold_values"]):
            continue
        t, p = stats.ttest_rel(best[1]["fold_values"], r["fold_values"])
        sig = "***" if p < 0.001 else "**" if p < 0.01 else "*" if p < 0.05 else "ns"
        tests[name] = dict(delta=best[1]["mean"] - r["mean"], t=float(t), p=float(p), sig=sig)
    return best[0], tests


def dataset_info(root="."):
    p = os.path.join(root, "data", "processed", "full_matching_table.csv")
    if not os.path.exists(p):
        return None
    import pandas as pd
    mt = pd.read_csv(p)
    complete = mt["has_imaging"] & mt["has_rnaseq"] & mt["has_clinical"] & mt["has_survival"]
    return {"Total patients": int(len(mt)), "With imaging": int(mt["has_imaging"].sum()), "With RNA-seq": int(mt["has_rnaseq"].sum()),
            "With clinical": int(mt["has_clinical"].sum()), "With survival": int(mt["has_survival"].sum()), "Complete (all 4)": int(complete.sum())}


def main(root="."):
    res = collect(root)
    if not res:
        raise SystemExit("no results/*/cv_results.json found under %s" % os.path.abspath(root))
    for name, r in sorted(res.items(), key=lambda kv: -kv[1]["mean"]):
        print(f"  {name:22s} C-index {r['mean']:.4f} +/- {r['std']:.4f}  folds {['%.3f' % v for v in r['fold_values']]}")
    best, tests = compare(res)
    print(f"best: {best} ({res[best]['mean']:.4f})")
    for name, t in tests.items():
        print(f"    vs {name}: delta={t['delta']:.4f}, p={t['p']:.4f} {t['sig']}")
    info = dataset_info(root)
    export = {"dataset_info": info, "model_results": {k: dict(c_index_mean=v["mean"], c_index_std=v["std"], fold_values=v["fold_values"],
                                                               n_patients=v["n_patients"]) for k, v in res.items()},
              "best_model": {"name": best, "c_index": res[best]["mean"], "std": res[best]["std"]}, "paired_t_tests": tests}
    os.makedirs(os.path.join(root, "results", "final_comparison"), exist_ok=True)
    with open(os.path.join(root, "results", "final_comparison", "results.json"), "w") as f:
        json.dump(export, f, indent=2)
    print("saved results/final_comparison/results.json")
    return export


if __name__ == "__main__":
    main()
