"""CPU restatement of the reference's Cox partial likelihood, C-index and gate-entropy regulariser.

`cox_npll_np` / `cox_npll_grad_np` are the O(B^2) risk-set (Breslow) form in numpy float64 -- the form the
HIP kernel implements.  On batches without tied times it equals the reference's sorted-logcumsumexp forms
(final_multimodal.py:171-186, simple_fusion.py:47-57, train_rnaseq_only.py:40-53); pinned by
tests/golden/g1_cox.npz.  torchsurv's Efron tie correction (final_multimodal.py:158-162 path) is NOT restated:
parity unpinned, contract = distinct times.
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


def concordance_index_np(log_hazard, event, time):
    """Harrell C as the reference's fallback counts it (simple_fusion.py:59-73, train_rnaseq_only.py:55-70):
    pairs (i event, t_j > t_i); concordant iff h_i > h_j (ties in h count as discordant); 0.5 if no pair."""
    h = np.asarray(log_hazard, np.float64)
    e = np.asarray(event) == 1
    t = np.asarray(time, np.float64)
    perm = (t[None, :] > t[:, None]) & e[:, None]
    conc = perm & (h[:, None] > h[None, :])
    p = int(perm.sum())
    return float(conc.sum()) / p if p > 0 else 0.5
