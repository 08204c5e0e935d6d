"""Efron tie handling of the Cox partial likelihood (torchsurv semantics, R/scripts/training/final_multimodal.py:158-162).
CPU part: the oracle restatements against the hand-derived known answers (tests/golden/efron_hand_derived.md) and the
property Efron == Breslow on distinct times.  GPU part: mms_cox_fwd_bwd with CoxP.tie_mode = 1 through the C ABI."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import losses as OL

G = os.path.join(os.path.dirname(__file__), "golden")
L = math.log


def _closed_forms():
    """The closed forms of tests/golden/efron_hand_derived.md, evaluated with math.log only."""
    return {
        "A": dict(loss=L(6), grad=[-5 / 12, -5 / 12, 5 / 6]),
        "B": dict(loss=(L(10 / 3) + L(19.25)) / 2,
                  grad=[-0.5 * (1 - 1 / 10 - 1 / 7 - 0.5 / 5.5), -0.5 * (1 - 2 / 10 - 2 / 7 - 1 / 5.5), -0.5 * (1 - 3 / 10),
                        -0.5 * (0 - 4 / 10 - 4 / 7 - 4 / 5.5)]),
        "C": dict(loss=L(8), grad=[-0.5, 0.0, 0.5]),
        "D": dict(loss=(L(4) + L(33.75)) / 2,
                  grad=[-0.5 * (1 - 2 / 12 - 2 / 9 - 1 / 7.5), -0.5 * (0 - 1 / 12 - 1 / 9 - 1 / 7.5),
                        -0.5 * (1 - 1 / 12 - 1 / 9 - 0.5 / 7.5), -0.5 * (1 - 3 / 12), -0.5 * (0 - 5 / 12 - 5 / 9 - 5 / 7.5)]),
        "E": dict(loss=(L(7 / 3) + L(5)) / 2, grad=None),
    }


def _cases():
    return json.load(open(os.path.join(G, "efron_cases.json")))


def test_fixture_matches_closed_forms():
    cf = _closed_forms()
    for name, c in _cases().items():
        assert c["loss"] == pytest.approx(cf[name]["loss"], rel=1e-12), name
        if cf[name]["grad"] is not None:
            np.testing.assert_allclose(c["grad"], cf[name]["grad"], rtol=1e-12, err_msg=name)
            assert abs(sum(c["grad"])) < 1e-12          # the partial likelihood is shift invariant


def test_oracle_efron_matches_hand_derived():
    for name, c in _cases().items():
        h, t, e = np.log(np.array(c["exp_h"], np.float64)), np.array(c["time"]), np.array(c["event"])
        v = np.array(c.get("valid", [1] * len(h))) > 0
        assert OL.cox_npll_efron_np(h[v], e[v], t[v]) == pytest.approx(c["loss"], rel=1e-12), name
        ht = torch.tensor(h[v], requires_grad=True)
        loss = OL.neg_partial_log_likelihood_efron(ht, torch.tensor(e[v]), torch.tensor(t[v]))
        assert loss.item() == pytest.approx(c["loss"], rel=1e-12), name
        if c.get("grad") is not None:
            loss.backward()
            np.testing.assert_allclose(ht.grad.numpy(), c["grad"], rtol=1e-10, atol=1e-14, err_msg=name)


@pytest.mark.parametrize("n,seed", [(2, 0), (5, 1), (16, 2), (64, 3), (257, 4)])
def test_oracle_efron_equals_breslow_on_distinct_times(n, seed):
    rng = np.random.default_rng(seed)
    h, t = rng.normal(size=n), rng.exponential(500, n) + 1 + np.arange(n) * 1e-3
    e = (rng.random(n) < 0.6).astype(np.float64)
    e[0] = 1
    assert OL.cox_npll_efron_np(h, e, t) == pytest.approx(OL.cox_npll_np(h, e, t), rel=1e-7)
    a = OL.neg_partial_log_likelihood_efron(torch.tensor(h), torch.tensor(e), torch.tensor(t)).item()
    assert a == pytest.approx(OL.cox_npll_np(h, e, t), rel=1e-7)


def test_oracle_degenerate():
    assert OL.cox_npll_efron_np([0.3], [1], [2.0]) == 0.0
    assert OL.cox_npll_efron_np([0.3, 0.1], [0, 0], [2.0, 2.0]) == 0.0
    assert OL.neg_partial_log_likelihood_efron(torch.tensor([0.3, 0.1]), torch.tensor([0, 0]), torch.tensor([2.0, 2.0])).item() == 0.0


# ---- GPU ------------------------------------------------------------------------------------------------------------
def _dev(a):
    return torch.tensor(np.asarray(a, np.float32)).to("cuda")


@pytest.mark.gpu
def test_hip_efron_matches_hand_derived():
    from multimodal_survival_prediction_amd import ops
    for name, c in _cases().items():
        h = np.log(np.array(c["exp_h"], np.float64))
        valid = _dev(c["valid"]) if "valid" in c else None
        out, dh = ops.cox_fwd_bwd(_dev(h), _dev(c["time"]), _dev(c["event"]), valid=valid, ties="efron")
        torch.cuda.synchronize()
        assert out[0].item() == pytest.approx(c["loss"], rel=1e-5), name
        assert out[1].item() == 1.0
        if c.get("grad") is not None:
            np.testing.assert_allclose(dh.cpu().numpy(), c["grad"], rtol=1e-4, atol=1e-6, err_msg=name)
        elif valid is not None:
            assert float(dh[valid == 0].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_times,seed", [(4, 2, 0), (8, 3, 1), (16, 5, 2), (64, 7, 3), (300, 40, 4), (2048, 97, 5)])
def test_hip_efron_vs_oracle_random_ties(n, n_times, seed):
    """Heavily tied batches (times drawn from n_times values) incl. a label mask: value and gradient vs the fp64 torch oracle."""
    from multimodal_survival_prediction_amd import ops
    rng = np.random.default_rng(seed)
    h = rng.normal(size=n).astype(np.float32)
    t = rng.integers(1, n_times + 1, n).astype(np.float32) * 10
    e = (rng.random(n) < 0.6).astype(np.float32)
    valid = (rng.random(n) < 0.8).astype(np.float32)
    e[0] = 1; valid[0] = 1; valid[1] = 1
    out, dh = ops.cox_fwd_bwd(_dev(h), _dev(t), _dev(e), valid=_dev(valid), scale=1.5, ties="efron")
    torch.cuda.synchronize()
    v = valid > 0
    ht = torch.tensor(h[v].astype(np.float64), requires_grad=True)
    ref = OL.neg_partial_log_likelihood_efron(ht, torch.tensor(e[v]), torch.tensor(t[v].astype(np.float64)))
    ref.backward()
    assert out[0].item() == pytest.approx(ref.item(), rel=1e-4), (n, n_times)
    g = np.zeros(n); g[v] = 1.5 * ht.grad.numpy()
    assert float(np.abs(dh.cpu().numpy() - g).max()) <= 1e-4 * max(float(np.abs(g).max()), 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4, 16, 257, 2048])
def test_hip_efron_equals_breslow_on_distinct_times(n):
    """On distinct times the two tie modes run the same arithmetic: bit-identical loss and gradient."""
    from multimodal_survival_prediction_amd import ops
    rng = np.random.default_rng(n)
    h = rng.normal(size=n).astype(np.float32)
    t = (rng.exponential(500, n) + 1 + np.arange(n) * 1e-2).astype(np.float32)
    assert len(np.unique(t)) == n
    e = (rng.random(n) < 0.6).astype(np.float32); e[0] = 1
    a, da = ops.cox_fwd_bwd(_dev(h), _dev(t), _dev(e), ties="breslow")
    b, db = ops.cox_fwd_bwd(_dev(h), _dev(t), _dev(e), ties="efron")
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(da, db)


@pytest.mark.gpu
def test_hip_python_surface_ties():
    """losses.cox_loss follows the USE_TORCHSURV switch (Efron) and back-propagates; ties="breslow" selects the risk-set form."""
    from multimodal_survival_prediction_amd import losses
    c = _cases()["D"]
    h = torch.tensor(np.log(np.array(c["exp_h"])), dtype=torch.float32, device="cuda", requires_grad=True)
    loss = losses.cox_loss(h, _dev(c["event"]), _dev(c["time"]))
    assert loss.item() == pytest.approx(c["loss"], rel=1e-5)
    loss.backward()
    np.testing.assert_allclose(h.grad.cpu().numpy(), c["grad"], rtol=1e-4, atol=1e-6)
    br = losses.cox_loss(h.detach(), _dev(c["event"]), _dev(c["time"]), ties="breslow").item()
    assert br == pytest.approx(OL.cox_npll_np(np.log(np.array(c["exp_h"])), c["event"], c["time"]), rel=1e-5) and br != pytest.approx(c["loss"], rel=1e-3)
