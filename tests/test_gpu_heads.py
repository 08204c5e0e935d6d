"""GPU parity of the head kernels (linear+BN1d+ReLU+Dropout, gate, Cox, C-index, clip+Adam) through the C ABI."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import DEV, assert_close

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    from multimodal_survival_prediction_amd import ops as o
    return o


def _d(t):
    return t.detach().to(DEV).contiguous()


@pytest.mark.parametrize("M,K,N", [(4, 5005, 512), (4, 96, 512), (8, 1024, 512), (16, 288, 256), (3, 1, 32), (32, 130, 7)])
@pytest.mark.parametrize("train", [True, False])
def test_linear_chain(ops, M, K, N, train):
    """Linear(K,N) -> BN1d(N) -> ReLU -> Dropout(mask) -> Linear(N,64) -> ReLU : both layers fwd + bwd."""
    torch.manual_seed(0)
    l1, bn, l2 = nn.Linear(K, N), nn.BatchNorm1d(N), nn.Linear(N, 64)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2); bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2)
    x = torch.randn(M, K)
    keep = (torch.rand(M, N) > 0.3).float() / 0.7 if train else torch.ones(M, N)
    l1.train(train); bn.train(train); l2.train(train)
    import copy
    bnd = copy.deepcopy(bn).to(DEV)
    y1 = l1(x)
    a = F.relu(bn(y1)) * keep
    y2 = F.relu(l2(a))
    dy2 = torch.randn_like(y2)
    y2.backward(dy2)
    xd, w1, b1, w2, b2 = _d(x), _d(l1.weight), _d(l1.bias), _d(l2.weight), _d(l2.bias)
    y1d, y2d = torch.empty(M, N, device=DEV), torch.empty(M, 64, device=DEV)
    ops.linear_fwd(xd, K, ops.inprolog(train=train), w1, b1, y1d, False)
    keepd = _d(keep)
    pro = ops.inprolog(bnd, train=train, drop_mask=keepd if train else None)
    ops.linear_fwd(y1d, N, pro, w2, b2, y2d, True)
    torch.cuda.synchronize()
    assert_close(y1d, y1, 1e-4, "y1")
    assert_close(y2d, y2, 1e-4, "y2")
    if train:
        assert_close(bnd.running_mean, bn.running_mean, 1e-4, "running_mean")
        assert_close(bnd.running_var, bn.running_var, 1e-4, "running_var")
        assert int(bnd.num_batches_tracked) == int(bn.num_batches_tracked)
    # backward
    dw2, db2 = torch.zeros_like(w2), torch.zeros_like(b2)
    dg, dbt = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    dy1 = torch.empty(M, N, device=DEV)
    rm0 = bnd.running_mean.clone()
    ops.linear_bwd(_d(dy2), y2d, True, y1d, N, pro, w2, dw2, db2, dy1, dg, dbt)
    dw1, db1 = torch.zeros_like(w1), torch.zeros_like(b1)
    ops.linear_bwd(dy1, y1d, False, xd, K, ops.inprolog(train=train), w1, dw1, db1)
    torch.cuda.synchronize()
    assert torch.equal(rm0, bnd.running_mean)      # backward must not touch running stats
    assert_close(dw2, l2.weight.grad, 1e-4, "dW2"); assert_close(db2, l2.bias.grad, 1e-4, "db2")
    assert_close(dg, bn.weight.grad, 1e-4, "dgamma"); assert_close(dbt, bn.bias.grad, 1e-4, "dbeta")
    assert_close(dw1, l1.weight.grad, 1e-4, "dW1")
    if train:   # bias feeding a training-mode BN has exactly zero gradient: both sides are rounding noise
        assert float(db1.abs().max()) <= 1e-4 * float(dy1.abs().max())
    else:
        assert_close(db1, l1.bias.grad, 1e-4, "db1")


def test_dropout_rng_statistics(ops):
    """perf-mode dropout (hash RNG): keep rate ~ 1-p, scale 1/(1-p), same mask in forward and backward."""
    M, K = 8, 4096
    x = torch.ones(M, K, device=DEV)
    w = torch.eye(K, device=DEV)[:64].contiguous()   # y[:, n] = x'[:, n]
    rng = torch.tensor([1234, 7], dtype=torch.int32, device=DEV)
    y = torch.empty(M, 64, device=DEV)
    full = torch.empty(M, K, device=DEV)
    wI = torch.eye(K, device=DEV)
    pro = ops.inprolog(train=True, drop_p=0.3, rng=rng, stream_id=3)
    ops.linear_fwd(x, K, pro, wI, None, full, False)
    torch.cuda.synchronize()
    kept = (full != 0).float().mean().item()
    assert abs(kept - 0.7) < 0.02
    assert_close(full[full != 0], torch.full_like(full[full != 0], 1 / 0.7), 1e-6, "scale")
    dx = torch.empty(M, K, device=DEV)
    dw = torch.zeros_like(wI)
    ops.linear_bwd(torch.ones(M, K, device=DEV), full, False, x, K, pro, wI, dw, None, dx)
    torch.cuda.synchronize()
    assert torch.equal(dx != 0, full != 0)
    rng2 = torch.tensor([1234, 8], dtype=torch.int32, device=DEV)
    full2 = torch.empty(M, K, device=DEV)
    ops.linear_fwd(x, K, ops.inprolog(train=True, drop_p=0.3, rng=rng2, stream_id=3), wI, None, full2, False)
    torch.cuda.synchronize()
    assert not torch.equal(full2 != 0, full != 0)    # next step -> new mask


@pytest.mark.parametrize("M", [1, 4, 8])
def test_gate(ops, M):
    torch.manual_seed(1)
    feats = torch.randn(M, 288).requires_grad_(True)
    mask = torch.tensor([[1, 1, 1], [0, 1, 1], [1, 0, 1], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 1, 1], [1, 0, 0]], dtype=torch.float32)[:M]
    gl1, gl2 = nn.Linear(291, 64), nn.Linear(64, 3)
    segs = [slice(0, 128), slice(128, 256), slice(256, 288)]
    masked = torch.cat([feats[:, s] * mask[:, i:i + 1] for i, s in enumerate(segs)], 1)
    gate = F.softmax(gl2(F.relu(gl1(torch.cat([masked, mask], 1)))), dim=1)
    fused = torch.cat([masked[:, s] * gate[:, i:i + 1] for i, s in enumerate(segs)], 1)
    ent = -(-(gate * torch.log(gate + 1e-8)).sum(1)).mean()
    dfused = torch.randn_like(fused)
    (fused * dfused).sum().backward(retain_graph=True)
    (0.01 * ent).backward()
    fd, md = _d(feats), _d(mask)
    w1, b1, w2, b2 = _d(gl1.weight), _d(gl1.bias), _d(gl2.weight), _d(gl2.bias)
    hidden, gated, fusedd = torch.empty(M, 64, device=DEV), torch.empty(M, 3, device=DEV), torch.empty(M, 288, device=DEV)
    entd = torch.zeros(1, device=DEV)
    dfe = torch.empty(M, 288, device=DEV)
    dw1, db1, dw2, db2 = torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(w2), torch.zeros_like(b2)
    dfd = _d(dfused)
    p = ops.gate_params(fd, md, w1, b1, w2, b2, hidden, gated, fusedd, dfd, 0.01, dfe, dw1, db1, dw2, db2, entd)
    ops.call("mms_gate_fwd", p)
    ops.call("mms_gate_bwd", p)
    torch.cuda.synchronize()
    assert_close(gated, gate, 1e-4, "gate"); assert_close(fusedd, fused, 1e-4, "fused")
    assert_close(entd, ent.reshape(1), 1e-4, "entropy loss")
    assert_close(dfe, feats.grad, 1e-4, "dfeats")
    assert_close(dw1, gl1.weight.grad, 1e-4, "dw1"); assert_close(db1, gl1.bias.grad, 1e-4, "db1")
    assert_close(dw2, gl2.weight.grad, 1e-4, "dw2"); assert_close(db2, gl2.bias.grad, 1e-4, "db2")


def test_cox_golden(ops):
    """Cox NPLL value + gradient against the reference's own cox_loss (tests/golden/g1_cox.npz, 36 cases incl.
    n=1, no events, single event, event at max time, n=2048)."""
    z = np.load(f"{G}/g1_cox.npz")
    cases = sorted({k[:-2] for k in z.files if k.endswith("_h")})
    assert len(cases) == 36
    for c in cases:
        h, e, t = (torch.tensor(z[c + s]).to(DEV) for s in ("_h", "_e", "_t"))
        out, dh = ops.cox_fwd_bwd(h, t, e)
        torch.cuda.synchronize()
        loss = float(z[c + "_loss"])
        assert abs(out[0].item() - loss) <= 1e-4 * max(1.0, abs(loss)), c
        usable = h.shape[0] >= 2 and float(z[c + "_e"].sum()) > 0
        assert out[1].item() == (1.0 if usable else 0.0), c
        ref = torch.tensor(z[c + "_grad"])
        assert float((dh.cpu() - ref).abs().max()) <= 1e-4 * max(float(ref.abs().max()), 1e-3), c


def test_cox_valid_mask_equals_subset(ops):
    """has_survival masking (partial_modality_training.py:401-415) == running on the labelled subset."""
    from oracle.losses import cox_npll_grad_np, cox_npll_np
    rng = np.random.default_rng(3)
    n = 16
    h, t = rng.normal(size=n).astype(np.float32), (rng.exponential(500, n) + 1 + np.arange(n) * 1e-3).astype(np.float32)
    e = (rng.random(n) < 0.6).astype(np.float32)
    valid = (rng.random(n) < 0.6).astype(np.float32)
    out, dh = ops.cox_fwd_bwd(*(torch.tensor(a).to(DEV) for a in (h, t, e)), valid=torch.tensor(valid).to(DEV), scale=2.0)
    torch.cuda.synchronize()
    idx = valid > 0
    assert abs(out[0].item() - cox_npll_np(h[idx], e[idx], t[idx])) < 1e-5
    g = np.zeros(n); g[idx] = 2.0 * cox_npll_grad_np(h[idx], e[idx], t[idx])
    np.testing.assert_allclose(dh.cpu().numpy(), g, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("n", [2048, 8192])
def test_cox_large_risk_set_vs_fp64_oracle(ops, n):
    """BASELINE config 5 (large risk-set stress): O(n^2) kernel pair vs the fp64 numpy oracle, tolerance 1e-4."""
    from oracle.losses import cox_npll_grad_np, cox_npll_np
    rng = np.random.default_rng(n)
    h = rng.normal(size=n).astype(np.float32)
    t = (rng.exponential(1000, n) + 1 + np.arange(n) * 1e-3).astype(np.float32)
    e = (rng.random(n) < 0.57).astype(np.float32)
    out, dh = ops.cox_fwd_bwd(*(torch.tensor(a).to(DEV) for a in (h, t, e)))
    torch.cuda.synchronize()
    ref = cox_npll_np(h, e, t)
    assert abs(out[0].item() - ref) <= 1e-4 * max(1.0, abs(ref)) and out[1].item() == 1.0
    g = cox_npll_grad_np(h, e, t)
    assert float(np.abs(dh.cpu().numpy() - g).max()) <= 1e-4 * float(np.abs(g).max())


def test_cindex_golden(ops):
    z = np.load(f"{G}/g2_cindex.npz")
    for n in ("n4", "n23", "n116", "n1639", "n5none"):
        c = ops.cindex_counts(*(torch.tensor(z[n + s]).to(DEV) for s in ("_h", "_t", "_e"))).cpu()
        got = c[0].item() / c[2].item() if c[2].item() > 0 else 0.5
        assert abs(got - float(z[n + "_cindex"])) < 1e-6, n
        assert c[1].item() == 0     # distinct hazards


@pytest.mark.parametrize("adamw", [False, True])
def test_clip_adam_matches_torch(ops, adamw):
    torch.manual_seed(5)
    n = 100_003
    p0 = torch.randn(n)
    ref = nn.Parameter(p0.clone())
    opt = (torch.optim.AdamW([ref], lr=1e-3, weight_decay=1e-2) if adamw
           else torch.optim.Adam([ref], lr=1e-3, weight_decay=1e-2))
    pd, m, v = _d(p0), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 1e-2, 1.0], device=DEV)
    sumsq, step = torch.zeros(1, dtype=torch.float64, device=DEV), torch.zeros(1, device=DEV)
    skip = torch.ones(1, device=DEV)
    for it in range(4):
        g = torch.randn(n) * (3.0 if it % 2 == 0 else 1e-3)     # clipped / unclipped steps
        ref.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        gd = _d(g)
        sumsq.zero_()
        ap = ops.adam_params(pd, gd, m, v, hyper, sumsq, step, skip, adamw)
        ops.call("mms_grad_sumsq", ap); ops.call("mms_clip_adam", ap)
        torch.cuda.synchronize()
        assert_close(pd, ref.detach(), 2e-6, f"params after step {it}")
    assert step.item() == 4.0
    skip.zero_()      # degenerate batch: nothing may move, step counter included
    before = pd.clone()
    sumsq.zero_()
    ap = ops.adam_params(pd, gd, m, v, hyper, sumsq, step, skip, adamw)
    ops.call("mms_grad_sumsq", ap); ops.call("mms_clip_adam", ap)
    torch.cuda.synchronize()
    assert torch.equal(before, pd) and step.item() == 4.0
