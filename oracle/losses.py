"""CPU restatement of the reference's Cox partial likelihood, C-index and gate-entropy regulariser.

`cox_npll_np` / `cox_npll_grad_np` are the O(B^2) risk-set (Breslow) form in numpy float64 -- the form the
HIP kernel implements.  On batches without tied times it equals the reference's sorted-logcumsumexp forms
(final_multimodal.py:171-186, simple_fusion.py:47-57, train_rnaseq_only.py:40-53); pinned by
tests/golden/g1_cox.npz.

`neg_partial_log_likelihood_efron` / `cox_npll_efron_np` restate torchsurv's tie handling (the final_multimodal.py:158-162 path:
`torchsurv.loss.cox.neg_partial_log_likelihood`, ties_method="efron", reduction="mean"; torchsurv >= 0.1.0 is NOT vendored and NOT
installed -- restated from the published algorithm, PARITY UNPINNED against torchsurv itself).  Pinned by hand-derived values
(tests/golden/efron_hand_derived.md) and by the property Efron == Breslow on distinct times.
"""
import numpy as np
import torch


def cox_npll_np(h, event, time):
    """-(1/n_e) sum_{i: e_i} (h_i - log sum_{j: t_j >= t_i} exp(h_j)); 0 if n<2 or no events
    (final_multimodal.py:173-176)."""
    h = np.asarray(h, np.float64)
    e = np.asarray(event, np.float64)
    t = np.asarray(time, np.float64)
    n = h.shape[0]
    if n < 2 or e.sum() == 0:
        return 0.0
    risk = t[None, :] >= t[:, None]                       # risk[i, j]: j still at risk at t_i
    m = h.max()
    lse = m + np.log((np.exp(h - m)[None, :] * risk).sum(1))
    return float(-((h - lse) * e).sum() / (e.sum() + 1e-8))


def cox_npll_grad_np(h, event, time):
    """dL/dh_k = -(1/n_e) [ e_k - sum_{i: e_i, t_i <= t_k} exp(h_k - lse_i) ]."""
    h = np.asarray(h, np.float64)
    e = np.asarray(event, np.float64)
    t = np.asarray(time, np.float64)
    n = h.shape[0]
    if n < 2 or e.sum() == 0:
        return np.zeros(n)
    risk = t[None, :] >= t[:, None]
    m = h.max()
    lse = m + np.log((np.exp(h - m)[None, :] * risk).sum(1))
    w = np.exp(h[None, :] - lse[:, None]) * risk * e[:, None]      # w[i, k]
    return -(e - w.sum(0)) / (e.sum() + 1e-8)


def cox_npll_efron_np(h, event, time):
    """torchsurv's Efron form in float64 (no autograd).  With the distinct event times tau_1..tau_J, H_j = events at tau_j
    (m_j of them), R_j = {k: t_k >= tau_j}, D_j = sum_{R_j} exp h, T_j = sum_{H_j} exp h:
        pll_j = sum_{H_j} h_i - sum_{l=0}^{m_j-1} log(D_j - (l/m_j) T_j),   loss = -mean_j pll_j   (mean over the J event times).
    0 if n < 2 or no events.  On distinct times m_j = 1 and this is cox_npll_np (up to its 1e-8 in the denominator)."""
    h = np.asarray(h, np.float64)
    e = np.asarray(event) != 0
    t = np.asarray(time, np.float64)
    if h.shape[0] < 2 or e.sum() == 0:
        return 0.0
    pll = []
    for tau in np.unique(t[e]):
        H = e & (t == tau)
        m = int(H.sum())
        D, T = np.exp(h[t >= tau]).sum(), np.exp(h[H]).sum()
        pll.append(h[H].sum() - sum(np.log(D - l / m * T) for l in range(m)))
    return float(-np.mean(pll))


def neg_partial_log_likelihood_efron(log_hz, event, time):
    """torch/autograd restatement of torchsurv.loss.cox.neg_partial_log_likelihood(log_hz, event, time, ties_method="efron",
    reduction="mean") as the reference calls it (final_multimodal.py:158-162, partial_modality_training.py:285-288,
    simple_fusion.py:270): sort by time; all times distinct -> flipped logcumsumexp form, mean over events; otherwise the Efron
    terms per distinct event time (cox_npll_efron_np), mean over those times; no events or a single sample -> 0."""
    event = event.bool()
    if event.sum() == 0 or log_hz.numel() < 2:
        return torch.tensor(0.0, requires_grad=True)
    ts, idx = torch.sort(time)
    hs, es = log_hz.reshape(-1)[idx], event[idx]
    tu = torch.unique(ts)
    if len(tu) == len(ts):
        log_den = torch.logcumsumexp(hs.flip(0), dim=0).flip(0)
        return -(hs - log_den)[es].mean()
    terms = []
    for tau in tu:
        H = (ts == tau) & es
        m = int(H.sum())
        if m == 0:
            continue
        D, T = torch.exp(hs[ts >= tau]).sum(), torch.exp(hs[H]).sum()
        den = sum(torch.log(D - l / m * T) for l in range(m))
        terms.append(hs[H].sum() - den)
    return -torch.stack(terms).mean()


def cox_loss(hazard, event, time):
    """torch/autograd restatement of the custom loss, final_multimodal.py:171-186 (identical text at
    partial_modality_training.py:296-311)."""
    if hazard.shape[0] < 2:
        return torch.tensor(0.0, device=hazard.device, requires_grad=True)
    if event.sum() == 0:
        return torch.tensor(0.0, device=hazard.device, requires_grad=True)
    order = torch.argsort(time, descending=True)
    hazard = hazard[order]
    event = event[order]
    log_cumsum = torch.logcumsumexp(hazard, dim=0)
    return -torch.sum((hazard - log_cumsum) * event) / (event.sum() + 1e-8)


def neg_partial_log_likelihood(log_hazard, event, time):
    """simple_fusion.py:47-57 (log(cumsum(exp)) form, event may be bool)."""
    idx = torch.argsort(time, descending=True)
    log_hazard = log_hazard[idx]
    event = event[idx]
    log_risk = torch.log(torch.cumsum(torch.exp(log_hazard), dim=0))
    return -torch.sum((log_hazard - log_risk) * event) / (torch.sum(event) + 1e-8)


def gate_entropy_loss(gate_weights):
    """partial_modality_training.py:322-331."""
    entropy = -torch.sum(gate_weights * torch.log(gate_weights + 1e-8), dim=1)
    return -entropy.mean()


def concordance_index_np(log_hazard, event, time, tie_credit=0.0):
    """Harrell C as the reference's fallback counts it (simple_fusion.py:59-73, train_rnaseq_only.py:55-70):
    pairs (i event, t_j > t_i); concordant iff h_i > h_j (ties in h count as discordant); 0.5 if no pair.
    tie_credit=0.5: the torchsurv / lifelines rule for tied risk scores (final_multimodal.py:164-169,188-194)."""
    h = np.asarray(log_hazard, np.float64)
    e = np.asarray(event) == 1
    t = np.asarray(time, np.float64)
    perm = (t[None, :] > t[:, None]) & e[:, None]
    conc = perm & (h[:, None] > h[None, :])
    tied = perm & (h[:, None] == h[None, :])
    p = int(perm.sum())
    return (float(conc.sum()) + tie_credit * float(tied.sum())) / p if p > 0 else 0.5
