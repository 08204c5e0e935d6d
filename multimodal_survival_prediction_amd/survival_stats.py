"""Host-side survival statistics used by scripts/analysis/evaluate_model.py -- numpy restatements of the three lifelines
routines the reference's analysis scripts call (R/scripts/analysis/evaluate_model.py:12-13,41-45,80-86;
generate_km_curves.py:14-15): Harrell's concordance index, the Kaplan-Meier product-limit estimator and the two-group
log-rank test.  lifelines is not installed here; the functions are pinned by textbook values in tests/test_cpu_survival_stats.py.
This is downstream analysis on a few hundred numbers, not part of the GPU hot path."""
import numpy as np
from scipy import stats


def concordance_index(event_times, predicted_scores, event_observed):
    """lifelines.utils.concordance_index semantics: the fraction of comparable pairs whose predicted scores are ordered like
    their survival times (higher score = longer survival; the reference passes -risk_score), score ties credited 1/2.
    A pair (i, j) is comparable when the shorter time is an observed event; at equal times only (event, censored) pairs are."""
    t = np.asarray(event_times, dtype=np.float64)
    s = np.asarray(predicted_scores, dtype=np.float64)
    e = np.asarray(event_observed).astype(bool)
    ti, tj = t[:, None], t[None, :]
    comparable = ((ti < tj) & e[:, None]) | ((ti == tj) & e[:, None] & ~e[None, :])
    n = int(comparable.sum())
    if n == 0:
        raise ZeroDivisionError("no comparable pairs")
    si, sj = s[:, None], s[None, :]
    conc = float(((si < sj) & comparable).sum()) + 0.5 * float(((si == sj) & comparable).sum())
    return conc / n


def kaplan_meier(durations, event_observed):
    """-> (times, survival, at_risk, events): the product-limit estimate S(t) at every distinct observed time (S(0) = 1 row first)."""
    t = np.asarray(durations, dtype=np.float64)
    e = np.asarray(event_observed).astype(bool)
    times = np.unique(t)
    at_risk = np.array([(t >= u).sum() for u in times], dtype=np.int64)
    deaths = np.array([((t == u) & e).sum() for u in times], dtype=np.int64)
    surv = np.cumprod(1.0 - deaths / np.maximum(at_risk, 1))
    return (np.concatenate([[0.0], times]), np.concatenate([[1.0], surv]), np.concatenate([[len(t)], at_risk]),
            np.concatenate([[0], deaths]))


def median_survival(durations, event_observed):
    """Smallest time with S(t) <= 0.5 (inf when the curve never gets there), as KaplanMeierFitter.median_survival_time_."""
    times, surv, _, _ = kaplan_meier(durations, event_observed)
    hit = np.nonzero(surv <= 0.5)[0]
    return float(times[hit[0]]) if len(hit) else float("inf")


def logrank_test(durations_a, durations_b, event_observed_a, event_observed_b):
    """Two-group log-rank (Mantel-Cox) test -> (chi2 with 1 dof, p value), hypergeometric variance as in lifelines."""
    ta, tb = np.asarray(durations_a, dtype=np.float64), np.asarray(durations_b, dtype=np.float64)
    ea, eb = np.asarray(event_observed_a).astype(bool), np.asarray(event_observed_b).astype(bool)
    o_minus_e, var = 0.0, 0.0
    for u in np.unique(np.concatenate([ta[ea], tb[eb]])):
        na, nb = float((ta >= u).sum()), float((tb >= u).sum())
        da, db = float(((ta == u) & ea).sum()), float(((tb == u) & eb).sum())
        n, d = na + nb, da + db
        if n < 2:
            continue
        o_minus_e += da - d * na / n
        var += d * (na / n) * (nb / n) * (n - d) / (n - 1)
    if var <= 0:
        return 0.0, 1.0
    chi2 = o_minus_e * o_minus_e / var
    return float(chi2), float(stats.chi2.sf(chi2, 1))


def risk_groups(risk_score):
    """Median split of evaluate_model.py:57-60: 'High Risk' where the score is strictly above the median."""
    r = np.asarray(risk_score, dtype=np.float64)
    return np.where(r > np.median(r), "High Risk", "Low Risk")
