#!/usr/bin/env python3
"""Model evaluation -- counterpart of the reference's scripts/analysis/evaluate_model.py (and of the prediction step that
scripts/analysis/generate_km_curves.py leaves unimplemented: "load each model's best-fold checkpoint and predict").

Reference behaviour kept (evaluate_model.py:27-45,57-60,191-225): read results/test_predictions.csv (columns
survival_time, event, risk_score), C-index = concordance_index(survival_time, -risk_score, event), median split into
'High Risk' (score > median) / 'Low Risk', Kaplan-Meier curves per group, results/evaluation_summary.json with the same
keys; the figures results/{kaplan_meier_curves,risk_score_distribution,survival_vs_risk}.png are written when matplotlib
is importable.  Added: the log-rank test between the two risk groups (what generate_km_curves.py imports logrank_test for)
and the Kaplan-Meier tables as CSV.

    python scripts/analysis/evaluate_model.py                            # evaluates an existing results/test_predictions.csv
    python scripts/analysis/evaluate_model.py --predict models/final/fold_1_best.pth --model final --fold 1

--predict runs the checkpoint's eval-mode forward on the MI355X (the HIP path; no CPU fallback) over the validation split
of that fold of the cohort the training entry points use (data/processed/* under MMS_DATA_ROOT when present, else the seeded
synthetic cohort) and writes results/test_predictions.csv first.  lifelines/seaborn are not needed: the statistics are the
numpy restatements in multimodal_survival_prediction_amd/survival_stats.py.
"""
import argparse
import json
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts", "training"))

from multimodal_survival_prediction_amd import survival_stats as SS  # noqa: E402

# --model -> (class, constructor kwargs, the cohort its training entry point builds: make_cohort kwargs, default patients / folds / batch)
MODELS = {
    "final": ("MultiModalSurvivalNet", lambda c: dict(rna_dim=c["rnaseq"].shape[1]), dict(seed=608, complete=True), 109, 5, 4),
    "partial": ("PartialModalityNet", lambda c: dict(rna_dim=c["rnaseq"].shape[1]), dict(seed=608, complete=False), 608, 3, 8),
    "simple": ("SimpleFusionModel", lambda c: dict(rna_dim=c["rnaseq"].shape[1]), dict(seed=88, complete=True), 88, 3, 8),
    "flexible": ("FlexibleMultimodalModel", lambda c: dict(rna_dim=c["rnaseq"].shape[1]), dict(seed=608, complete=False), 608, 3, 16),
    "rnaseq": ("RNASeqSurvivalModel", lambda c: dict(input_dim=c["rnaseq"].shape[1]), dict(seed=427, complete=True, dims=(32, 32, 32)), 240, 3, 16),
}


def predict(checkpoint, kind, fold, n_folds, batch_size, out_csv):
    """Eval-mode log-hazards of the fold's validation patients -> results/test_predictions.csv.  The cohort and the fold split
    are rebuilt exactly as the model's training entry point builds them (scripts/training/<name>.py: same seeds, K-fold over the
    labelled patients with random_state 42), so fold k here is the held-out split of models/<name>/fold_k_best.pth."""
    import torch
    from _common import env_int, load_or_make_cohort, setup_device
    from multimodal_survival_prediction_amd import data, models
    from multimodal_survival_prediction_amd.engine import engine_of
    _, _, device = setup_device()
    cls, ctor, ckw, n_default, folds_default, batch_default = MODELS[kind]
    n_folds, batch_size = n_folds or env_int("MMS_FOLDS", folds_default), batch_size or env_int("MMS_BATCH_SIZE", batch_default)
    n = env_int("MMS_PATIENTS", n_default)
    if kind == "partial":
        cohort = load_or_make_cohort(device, n=n, **ckw)                 # data/processed/* under MMS_DATA_ROOT when present
    else:
        cohort = data.cohort_to(data.make_cohort(n=n, **ckw), device)
    labelled = torch.nonzero(cohort["has_survival"].cpu()).reshape(-1).numpy()
    _, val = data.kfold_indices(len(labelled), n_folds, seed=42)[fold - 1]
    idx = labelled[val]
    model = getattr(models, cls)(**ctor(cohort))
    model.load_state_dict(torch.load(checkpoint, map_location="cpu"))
    model.to(device).eval()
    eng = engine_of(model)
    risks = []
    for s in range(0, len(idx), batch_size):
        j = torch.as_tensor(idx[s:s + batch_size], device=device)
        ct, rna, clin, mask = cohort["image"][j], cohort["rnaseq"][j], cohort["clinical"][j], cohort["mask"][j]
        if kind == "final":
            hz, _ = eng.forward_eval(ct, rna, clin)
        elif kind == "partial":
            hz, _ = eng.forward_eval(ct, rna, clin, mask=mask)
        elif kind == "simple":
            hz, _ = eng.forward_eval(ct, rna)
        elif kind == "flexible":
            hz, _ = eng.forward_eval(ct, rna, mask=mask[:, :2])
        else:
            hz, _ = eng.forward_eval(None, rna)
        risks.append(hz.clone().cpu())
    eng.check_b4()                                     # a timed-out block-4 hand-off must not end up in the predictions file
    lab = cohort["label"].cpu().numpy()[idx]
    ids = cohort.get("patient_id")
    df = pd.DataFrame({"patient_id": [ids[i] for i in idx] if ids is not None else [f"SYN-{i:04d}" for i in idx],
                       "survival_time": lab[:, 0], "event": lab[:, 1].astype(int), "risk_score": torch.cat(risks).numpy()})
    os.makedirs(os.path.dirname(out_csv) or ".", exist_ok=True)
    df.to_csv(out_csv, index=False)
    print(f"wrote {out_csv}: {len(df)} patients of fold {fold}/{n_folds} ({cls})")
    return df


def evaluate(df, outdir="results", plots=True):
    """-> the evaluation_summary.json dictionary (reference keys + log-rank + per-group statistics)."""
    t, e, r = df["survival_time"].to_numpy(float), df["event"].to_numpy(int), df["risk_score"].to_numpy(float)
    c_index = SS.concordance_index(t, -r, e)
    group = SS.risk_groups(r)
    lo, hi = group == "Low Risk", group == "High Risk"
    chi2, p = SS.logrank_test(t[hi], t[lo], e[hi], e[lo]) if hi.any() and lo.any() else (0.0, 1.0)
    summary = {
        "test_patients": int(len(df)), "deaths": int(e.sum()), "censored": int((1 - e).sum()), "c_index": float(c_index),
        "median_survival_time": float(np.median(t)), "median_risk_score": float(np.median(r)),
        "risk_groups": {"low_risk": int(lo.sum()), "high_risk": int(hi.sum())},
        "logrank": {"chi2": chi2, "p_value": p},
        "group_statistics": {},
    }
    os.makedirs(outdir, exist_ok=True)
    km_rows = []
    for name, m in (("Low Risk", lo), ("High Risk", hi)):
        if not m.any():
            continue
        summary["group_statistics"][name] = {
            "patients": int(m.sum()), "deaths": int(e[m].sum()), "censored": int((1 - e[m]).sum()),
            "mean_survival_time": float(t[m].mean()), "median_survival_time": float(np.median(t[m])),
            "km_median_survival_time": SS.median_survival(t[m], e[m]), "mean_risk_score": float(r[m].mean())}
        times, surv, at_risk, deaths = SS.kaplan_meier(t[m], e[m])
        km_rows += [dict(group=name, time=a, survival=b, at_risk=int(c), events=int(d)) for a, b, c, d in zip(times, surv, at_risk, deaths)]
    pd.DataFrame(km_rows).to_csv(os.path.join(outdir, "kaplan_meier_table.csv"), index=False)
    with open(os.path.join(outdir, "evaluation_summary.json"), "w") as f:
        json.dump(summary, f, indent=2)
    if plots:
        _plots(df.assign(risk_group=group), km_rows, summary, outdir)
    return summary


def _plots(df, km_rows, summary, outdir):
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        print("matplotlib not available: figures skipped")
        return
    km = pd.DataFrame(km_rows)
    fig, ax = plt.subplots(figsize=(7, 5))
    for name, colour in (("Low Risk", "tab:blue"), ("High Risk", "tab:red")):
        g = km[km["group"] == name]
        if len(g):
            ax.step(g["time"], g["survival"], where="post", label=name, color=colour)
    ax.set_xlabel("Time (days)"); ax.set_ylabel("Survival Probability"); ax.set_ylim(0, 1.02)
    ax.set_title(f"Kaplan-Meier by risk group (log-rank p = {summary['logrank']['p_value']:.3g})")
    ax.legend(loc="best"); ax.grid(True, alpha=0.3)
    fig.tight_layout(); fig.savefig(os.path.join(outdir, "kaplan_meier_curves.png"), dpi=150); plt.close(fig)
    fig, ax = plt.subplots(figsize=(7, 5))
    for name, colour in (("Low Risk", "blue"), ("High Risk", "red")):
        ax.hist(df[df["risk_group"] == name]["risk_score"], bins=15, alpha=0.6, label=name, color=colour)
    ax.axvline(summary["median_risk_score"], color="black", linestyle="--", label="Median")
    ax.set_xlabel("Risk Score"); ax.set_ylabel("Frequency"); ax.set_title("Risk Score Distribution"); ax.legend(); ax.grid(True, alpha=0.3)
    fig.tight_layout(); fig.savefig(os.path.join(outdir, "risk_score_distribution.png"), dpi=150); plt.close(fig)
    fig, ax = plt.subplots(figsize=(8, 5))
    for val, colour, label in ((0, "blue", "Censored"), (1, "red", "Death")):
        m = df["event"] == val
        ax.scatter(df[m]["risk_score"], df[m]["survival_time"], c=colour, alpha=0.6, s=60, label=label, edgecolors="black", linewidths=0.5)
    ax.set_xlabel("Risk Score"); ax.set_ylabel("Survival Time (days)"); ax.set_title("Survival Time vs Risk Score"); ax.legend(); ax.grid(True, alpha=0.3)
    fig.tight_layout(); fig.savefig(os.path.join(outdir, "survival_vs_risk.png"), dpi=150); plt.close(fig)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--predictions", default="results/test_predictions.csv")
    ap.add_argument("--outdir", default="results")
    ap.add_argument("--predict", metavar="CHECKPOINT", help="state_dict written by a training entry point (models/<name>/fold_k_best.pth)")
    ap.add_argument("--model", choices=sorted(MODELS), default="final")
    ap.add_argument("--fold", type=int, default=1)
    ap.add_argument("--n-folds", type=int, default=0, help="default: the training script's N_FOLDS (MMS_FOLDS)")
    ap.add_argument("--batch-size", type=int, default=0, help="default: the training script's BATCH_SIZE (MMS_BATCH_SIZE)")
    ap.add_argument("--no-plots", action="store_true")
    a = ap.parse_args(argv)
    df = predict(a.predict, a.model, a.fold, a.n_folds, a.batch_size, a.predictions) if a.predict else pd.read_csv(a.predictions)
    s = evaluate(df, a.outdir, plots=not a.no_plots)
    print(f"patients {s['test_patients']} (deaths {s['deaths']}, censored {s['censored']}); C-index {s['c_index']:.4f}; "
          f"median risk {s['median_risk_score']:.4f}; low/high {s['risk_groups']['low_risk']}/{s['risk_groups']['high_risk']}; "
          f"log-rank chi2 {s['logrank']['chi2']:.3f} p {s['logrank']['p_value']:.3g}")
    print(f"saved {os.path.join(a.outdir, 'evaluation_summary.json')}, kaplan_meier_table.csv" + ("" if a.no_plots else ", figures"))
    return s


if __name__ == "__main__":
    main()
