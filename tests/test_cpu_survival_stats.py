"""survival_stats (numpy restatements of lifelines' concordance_index / KaplanMeierFitter / logrank_test used by the
reference's analysis scripts) against textbook values: the Freireich 6-MP leukaemia trial (Kaplan-Meier table and log-rank
chi-square 16.79 as printed in Kleinbaum & Klein, Survival Analysis, ch. 2) and a brute-force pair count."""
import numpy as np
import pytest

from multimodal_survival_prediction_amd import survival_stats as SS

T_6MP = [6, 6, 6, 6, 7, 9, 10, 10, 11, 13, 16, 17, 19, 20, 22, 23, 25, 32, 32, 34, 35]
E_6MP = [1, 1, 1, 0, 1, 0, 1, 0, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0]
T_PLA = [1, 1, 2, 2, 3, 4, 4, 5, 5, 8, 8, 8, 8, 11, 11, 12, 12, 15, 17, 22, 23]
E_PLA = [1] * 21


def test_kaplan_meier_freireich():
    times, surv, at_risk, deaths = SS.kaplan_meier(T_6MP, E_6MP)
    s = dict(zip(times.tolist(), surv.tolist()))
    for t, want in ((6, 0.857), (7, 0.807), (10, 0.753), (13, 0.690), (16, 0.627), (22, 0.538), (23, 0.448)):
        assert s[float(t)] == pytest.approx(want, abs=6e-4)
    assert s[0.0] == 1.0 and s[9.0] == s[7.0]                       # censoring times leave the curve flat
    assert dict(zip(times.tolist(), at_risk.tolist()))[6.0] == 21 and dict(zip(times.tolist(), deaths.tolist()))[6.0] == 3
    assert SS.median_survival(T_6MP, E_6MP) == 23.0
    assert SS.median_survival(T_PLA, E_PLA) == 8.0
    assert SS.median_survival([5, 6, 7], [0, 0, 0]) == float("inf")


def test_logrank_freireich():
    chi2, p = SS.logrank_test(T_6MP, T_PLA, E_6MP, E_PLA)
    assert chi2 == pytest.approx(16.79, abs=0.01)
    assert 1e-5 < p < 1e-4
    chi2_same, p_same = SS.logrank_test(T_PLA, T_PLA, E_PLA, E_PLA)
    assert chi2_same == pytest.approx(0.0, abs=1e-12) and p_same == pytest.approx(1.0)


def test_concordance_index_pairs():
    rng = np.random.default_rng(0)
    n = 60
    t = rng.exponential(100, n).round(0)                             # rounding makes ties in time
    e = rng.random(n) < 0.6
    s = rng.normal(size=n).round(1)                                  # and ties in score
    num = den = 0.0
    for i in range(n):
        for j in range(n):
            if (t[i] < t[j] and e[i]) or (t[i] == t[j] and e[i] and not e[j]):
                den += 1
                num += 1.0 if s[i] < s[j] else (0.5 if s[i] == s[j] else 0.0)
    assert SS.concordance_index(t, s, e) == pytest.approx(num / den, abs=1e-12)
    assert SS.concordance_index([1, 2, 3, 4], [1, 2, 3, 4], [1, 1, 1, 1]) == 1.0
    assert SS.concordance_index([1, 2, 3, 4], [4, 3, 2, 1], [1, 1, 1, 0]) == 0.0
    with pytest.raises(ZeroDivisionError):
        SS.concordance_index([1, 2], [1, 2], [0, 0])


def test_risk_groups_median_split():
    g = SS.risk_groups([0.1, 0.5, 0.3, 0.9, 0.7])
    assert g.tolist() == ["Low Risk", "Low Risk", "Low Risk", "High Risk", "High Risk"]      # median 0.5 itself is low risk


def test_evaluate_model_script_schema(tmp_path):
    """scripts/analysis/evaluate_model.py on a predictions CSV: the reference's evaluation_summary.json keys, the KM table and
    (matplotlib present) the three figures."""
    import importlib.util, json, os
    import pandas as pd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("evaluate_model", os.path.join(root, "scripts", "analysis", "evaluate_model.py"))
    em = importlib.util.module_from_spec(spec); spec.loader.exec_module(em)
    rng = np.random.default_rng(3)
    n = 40
    risk = rng.normal(size=n)
    t = rng.exponential(1000, n) * np.exp(-2.0 * risk) + 1
    e = (rng.random(n) < 0.6).astype(int)
    pd.DataFrame(dict(patient_id=[f"P{i}" for i in range(n)], survival_time=t, event=e, risk_score=risk)).to_csv(tmp_path / "pred.csv", index=False)
    s = em.main(["--predictions", str(tmp_path / "pred.csv"), "--outdir", str(tmp_path / "out")])
    saved = json.load(open(tmp_path / "out" / "evaluation_summary.json"))
    for k in ("test_patients", "deaths", "censored", "c_index", "median_survival_time", "median_risk_score", "risk_groups"):
        assert k in saved                                             # evaluate_model.py:191-203
    assert saved["test_patients"] == n and saved["risk_groups"] == {"low_risk": 20, "high_risk": 20}
    assert saved["c_index"] == pytest.approx(SS.concordance_index(t, -risk, e)) and saved["c_index"] > 0.6   # planted signal
    assert 0 <= saved["logrank"]["p_value"] < 0.05 and s["deaths"] == int(e.sum())
    km = pd.read_csv(tmp_path / "out" / "kaplan_meier_table.csv")
    assert set(km["group"]) == {"Low Risk", "High Risk"} and km["survival"].between(0, 1).all()
    for f in ("kaplan_meier_curves.png", "risk_score_distribution.png", "survival_vs_risk.png"):
        assert os.path.getsize(tmp_path / "out" / f) > 1000
