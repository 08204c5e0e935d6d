"""GPU parity of the three networks (DenseNet121 encoder + heads + Cox) against the CPU oracle restatement:
eval hazards, train hazards/gate, loss, gradients (autograd-compatible path) and the fused HIP-graph step."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV, assert_close, rel_err
from test_gpu_densenet import structured_volumes


def _pair(cls, seed, rna_dim):
    from oracle import models as OM
    from multimodal_survival_prediction_amd import models as HM
    torch.manual_seed(seed)
    ref = getattr(OM, cls)(rna_dim=rna_dim, use_monai=True)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, (torch.nn.BatchNorm3d, torch.nn.BatchNorm1d)):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1)
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    net = getattr(HM, cls)(rna_dim=rna_dim)
    net.load_state_dict(ref.state_dict())
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return ref, net.to(DEV)


def _batch(B, dims, rna_dim, seed):
    rng = np.random.default_rng(seed)
    ct = structured_volumes(B, dims, seed)
    rna = torch.tensor(rng.normal(0, 1, (B, rna_dim)).astype(np.float32))
    clin = torch.tensor((np.clip(rng.normal(60, 11, (B, 1)), 30, 90) / 100).astype(np.float32))
    t = torch.tensor((rng.exponential(1000, B) + 1 + np.arange(B) * 1e-3).astype(np.float32))
    e = torch.tensor((rng.random(B) < 0.6).astype(np.float32)); e[0] = 1
    mask = torch.tensor([[1, 1, 1], [0, 1, 1], [1, 0, 1], [1, 1, 0], [0, 1, 0], [1, 1, 1], [0, 0, 1], [1, 0, 0]], dtype=torch.float32)[:B]
    return ct, rna, clin, t, e, mask


def _grad_stats(ref, net):
    """-> (p10, max, global relative L2, head-only max) of per-tensor errors.  Parameters whose gradient is exactly
    zero in exact arithmetic (a bias feeding a training-mode BatchNorm; cox_head.bias under the shift-invariant Cox
    loss) hold only rounding noise on both sides and are skipped."""
    errs, heads, num, den = [], [], 0.0, 0.0
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters())
    for (k, p), (k2, q) in zip(ref.named_parameters(), net.named_parameters()):
        assert k == k2
        a, b = p.grad.double(), q.grad.double().cpu()
        num += float(((a - b) ** 2).sum()); den += float((a ** 2).sum())
        if float(a.abs().max()) < 1e-5 * gmax:
            assert float(b.abs().max()) < 1e-4 * gmax, k
            continue
        errs.append(rel_err(b, a))
        if "encoder.features" not in k and "encoder.class_layers" not in k:
            heads.append(errs[-1])
    return float(np.percentile(errs, 10)), max(errs), (num / den) ** 0.5, max(heads)


@pytest.mark.parametrize("cls", ["MultiModalSurvivalNet", "PartialModalityNet", "SimpleFusionModel"])
def test_model_parity_autograd_path(cls):
    from oracle import losses as OL
    from multimodal_survival_prediction_amd import losses as HL
    B, dims, rna_dim = 4, (64, 64, 32), 5005
    ref, net = _pair(cls, 3, rna_dim)
    ct, rna, clin, t, e, mask = _batch(B, dims, rna_dim, 9)
    d = lambda x: x.to(DEV)
    args_ref = {"MultiModalSurvivalNet": (ct, rna, clin), "PartialModalityNet": (ct, rna, clin, mask), "SimpleFusionModel": (ct, rna)}[cls]
    args_net = tuple(d(a) for a in args_ref)
    # eval
    ref.eval(); net.eval()
    with torch.no_grad():
        w, g = ref(*args_ref), net(*args_net)
    if cls == "PartialModalityNet":
        assert_close(g[0], w[0], 1e-4, "eval hazard"); assert_close(g[1], w[1], 1e-4, "eval gate")
    else:
        assert_close(g, w, 1e-4, "eval hazard")
    # train: forward + loss + backward
    ref.train(); net.train()
    w, g = ref(*args_ref), net(*args_net)
    if cls == "PartialModalityNet":
        lw = OL.cox_loss(w[0], e, t) + 0.01 * OL.gate_entropy_loss(w[1])
        lg = HL.cox_loss(g[0], d(e), d(t)) + 0.01 * HL.gate_entropy_loss(g[1])
        assert_close(g[0], w[0], 1e-4, "train hazard"); assert_close(g[1], w[1], 1e-4, "train gate")
    else:
        lw, lg = OL.cox_loss(w, e, t), HL.cox_loss(g, d(e), d(t))
        assert_close(g, w, 1e-4, "train hazard")
    assert abs(lg.item() - lw.item()) <= 1e-4 * max(1.0, abs(lw.item()))
    lw.backward(); lg.backward()
    torch.cuda.synchronize()
    p10, mx, l2, hmax = _grad_stats(ref, net)
    print(f"{cls}: grad parity p10 {p10:.2e} max {mx:.2e} global-L2 {l2:.2e} heads-max {hmax:.2e}")
    # Head gradients do not pass through the encoder's ReLUs: strict 1e-4.  Encoder gradients: flip-aware statistical
    # criteria (test_gpu_densenet.py); one flip in a small-M block moves everything upstream of it by
    # ~1/sqrt(#elements of that block) ~ 3e-3, so the bounds are p10 <= 5e-5, global L2 <= 1e-2, worst tensor <= 0.15.
    assert hmax <= 1e-4, hmax
    assert p10 <= 5e-5 and mx <= 0.15 and l2 <= 1e-2


@pytest.mark.parametrize("cls", ["MultiModalSurvivalNet", "PartialModalityNet", "SimpleFusionModel"])
def test_fused_graph_step_matches_reference_loop_body(cls):
    """Two optimisation steps of the fused HIP-graph step == two iterations of the reference loop body
    (zero_grad, backward, clip_grad_norm_(1.0), Adam/AdamW step) run with torch on the CPU oracle."""
    from oracle import losses as OL
    from multimodal_survival_prediction_amd.training import FusedOptimizer
    B, dims, rna_dim = 4, (64, 64, 32), 5005
    ref, net = _pair(cls, 4, rna_dim)
    adamw = cls == "SimpleFusionModel"
    opt_ref = (torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=1e-3) if adamw
               else torch.optim.Adam(ref.parameters(), lr=1e-4, weight_decay=1e-4))
    fo = FusedOptimizer(net, lr=1e-4, weight_decay=1e-3 if adamw else 1e-4, adamw=adamw)
    eng = fo.engine
    ref.train(); net.train()
    losses_ref = []
    for it in range(2):
        ct, rna, clin, t, e, mask = _batch(B, dims, rna_dim, 20 + it)
        valid = torch.tensor([1, 1, 0, 1], dtype=torch.float32) if cls != "MultiModalSurvivalNet" else None
        if cls == "MultiModalSurvivalNet":
            hz = ref(ct, rna, clin); loss = OL.cox_loss(hz, e, t)
            eng.train_step(ct, rna, clin, time=t, event=e, skip_if_unusable=True)
        elif cls == "PartialModalityNet":
            hz, gw = ref(ct, rna, clin, mask)
            sm = valid.bool()
            loss = OL.cox_loss(hz[sm], e[sm], t[sm]) + 0.01 * OL.gate_entropy_loss(gw)
            eng.train_step(ct, rna, clin, mask=mask, time=t, event=e, valid=valid, skip_if_unusable=False)
        else:
            hz = ref(ct, rna)
            sm = valid.bool()
            loss = OL.neg_partial_log_likelihood(hz[sm], e[sm].bool(), t[sm])
            eng.train_step(ct, rna, time=t, event=e, valid=valid, skip_if_unusable=True)
        opt_ref.zero_grad(); loss.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt_ref.step()
        losses_ref.append(loss.item())
    torch.cuda.synchronize()
    st = eng.epoch_stats()
    assert st["n_batches"] == 2 and st["n_usable"] == 2
    # Adam's first steps move every weight by ~lr regardless of gradient scale: compare the UPDATE, not the weight
    ref0, _ = _pair(cls, 4, rna_dim)
    worst = 0.0
    for (k, p), (_, q), (_, p0) in zip(ref.named_parameters(), net.named_parameters(), ref0.named_parameters()):
        du_ref, du_net = (p.detach() - p0.detach()).double(), (q.detach().cpu() - p0.detach()).double()
        worst = max(worst, float((du_ref - du_net).abs().max()))
    # Adam moves a weight by <= ~lr per step whatever the gradient scale, so parameters whose exact gradient is zero
    # (rounding noise on both sides) can end up 2 steps x 2 lr apart; everything else must agree closely.
    assert worst <= 4.2e-4, worst
    tot = sum(p.numel() for p in ref.parameters())
    close = sum(float(((p.detach() - q.detach().cpu()).abs() <= 2e-5).double().sum())
                for (k, p), (_, q) in zip(ref.named_parameters(), net.named_parameters()))
    print(f"{cls}: fused step: worst update diff {worst:.2e}, {close / tot:.4f} of all weights within 2e-5 (lr = 1e-4)")
    assert close / tot >= 0.93, close / tot     # flip lottery: one early-block flip moves ~3 % of the weights by > 2e-5
    # BN running statistics after two training forwards
    for (k, b), (_, c) in zip(ref.named_buffers(), net.named_buffers()):
        if "num_batches" in k:
            assert int(b) == int(c), k
        else:   # second forward ran on weights that already differ by the (noise-gradient) Adam updates above
            assert_close(c, b, 2e-3, k)


def test_config4_volume_shape_parity():
    """BASELINE config 4's volume shape: CT 128x128x64 -- 8x the voxels of the headline shape: block-1 grid 32x32x16 (W = 16, the
    widest window the multi-tap forward stages), more statistic replicas per level.  Eval and training-mode hazards, loss and
    gradients against the CPU oracle.  Batch 4 rather than the config's 2 per rank: with two rows every training-mode
    BatchNorm1d output is exactly +-1 and its backward is a difference of nearly equal numbers (measured at B = 2: hazards and
    loss within 1e-4, head gradients 5e-4 -- conditioning of the reference's own arithmetic, not a kernel property)."""
    from oracle import losses as OL
    from multimodal_survival_prediction_amd import losses as HL
    B, dims, rna_dim = 4, (128, 128, 64), 5005
    ref, net = _pair("MultiModalSurvivalNet", 5, rna_dim)
    ct, rna, clin, t, e, _ = _batch(B, dims, rna_dim, 31)
    e[:] = 1
    d = lambda x: x.to(DEV)
    ref.eval(); net.eval()
    with torch.no_grad():
        assert_close(net(d(ct), d(rna), d(clin)), ref(ct, rna, clin), 1e-4, "eval hazard")
    ref.train(); net.train()
    w, g = ref(ct, rna, clin), net(d(ct), d(rna), d(clin))
    assert_close(g, w, 1e-4, "train hazard")
    lw, lg = OL.cox_loss(w, e, t), HL.cox_loss(g, d(e), d(t))
    assert abs(lg.item() - lw.item()) <= 1e-4 * max(1.0, abs(lw.item()))
    lw.backward(); lg.backward()
    torch.cuda.synchronize()
    p10, mx, l2, hmax = _grad_stats(ref, net)
    print(f"config-4 shape: grad parity p10 {p10:.2e} max {mx:.2e} global-L2 {l2:.2e} heads-max {hmax:.2e}")
    # heads: strict.  Encoder: the flip-aware criteria of test_model_parity_autograd_path; eight times as many ReLU inputs mean
    # proportionally more sign flips between two fp32 summation orders, so the 10th-percentile bound is 5e-4 here (measured
    # 1.6e-4; global L2 1.1e-3, worst tensor 3.3e-2, heads 1.2e-5)
    assert hmax <= 1e-4, hmax
    assert p10 <= 5e-4 and mx <= 0.15 and l2 <= 1e-2


def _per_tensor_err(ref_named, got_named):
    """{name: max-abs error relative to the reference tensor's max} over tensors with a non-negligible exact gradient."""
    gmax = max(float(g.abs().max()) for g in ref_named.values())
    out = {}
    for k, a in ref_named.items():
        if float(a.abs().max()) < 1e-5 * gmax:
            continue
        out[k] = float((got_named[k].double() - a.double()).abs().max() / a.double().abs().max())
    return out


def test_gradient_error_vs_fp64_is_fp32_rounding():
    """Evidence for the loose network-level gradient bounds above: the SAME batch through (a) the oracle in
    fp64, (b) the oracle in fp32, (c) the HIP path, for two model classes.  Taking fp64 as exact, the HIP path's per-tensor gradient
    errors must lie inside the envelope of the fp32 CPU oracle's OWN errors: both are fp32 implementations that differ from exact
    arithmetic by rounding and by the handful of ReLU inputs that sit within rounding of zero.  Which implementation draws an
    early-block flip on a given batch is a lottery (measured: MultiModalSurvivalNet -- oracle median 3.2e-3 / HIP 1.0e-3;
    PartialModalityNet -- oracle 1.2e-5 / HIP 3.9e-3), so the envelope is taken over the cases, with a factor 2."""
    from oracle import losses as OL
    from multimodal_survival_prediction_amd import losses as HL
    B, dims, rna_dim = 4, (64, 64, 32), 5005
    rows = []
    for cls in ("MultiModalSurvivalNet", "PartialModalityNet"):
        ref, net = _pair(cls, 3, rna_dim)
        ct, rna, clin, t, e, mask = _batch(B, dims, rna_dim, 9)
        ref64 = copy.deepcopy(ref).double()
        d = lambda x: x.to(DEV)

        def run(model, cast):
            model.train()
            args = (cast(ct), cast(rna), cast(clin)) + ((cast(mask),) if cls == "PartialModalityNet" else ())
            out = model(*args)
            if cls == "PartialModalityNet":
                loss = OL.cox_loss(out[0], cast(e), cast(t)) + 0.01 * OL.gate_entropy_loss(out[1])
            else:
                loss = OL.cox_loss(out, cast(e), cast(t))
            loss.backward()
            return {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        g64 = run(ref64, lambda x: x.double())
        g32 = run(ref, lambda x: x)
        net.train()
        args = (d(ct), d(rna), d(clin)) + ((d(mask),) if cls == "PartialModalityNet" else ())
        out = net(*args)
        lg = (HL.cox_loss(out[0], d(e), d(t)) + 0.01 * HL.gate_entropy_loss(out[1])) if cls == "PartialModalityNet" else HL.cox_loss(out, d(e), d(t))
        lg.backward()
        torch.cuda.synchronize()
        ghip = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
        e32, ehip = _per_tensor_err(g64, g32), _per_tensor_err(g64, ghip)
        keys = sorted(e32)
        a32, ah = np.array([e32[k] for k in keys]), np.array([ehip[k] for k in keys])
        l2 = lambda g: (sum(float(((g[k].double() - g64[k]) ** 2).sum()) for k in g64) / sum(float((g64[k] ** 2).sum()) for k in g64)) ** 0.5
        heads = [k for k in keys if "encoder.features" not in k and "encoder.class_layers" not in k]
        rows.append(dict(cls=cls, med32=np.median(a32), max32=a32.max(), l232=l2(g32), medh=np.median(ah), maxh=ah.max(), l2h=l2(ghip),
                         headh=max(ehip[k] for k in heads), head32=max(e32[k] for k in heads)))
        print(f"{cls}: vs fp64 -- fp32 oracle: median {rows[-1]['med32']:.2e} max {rows[-1]['max32']:.2e} L2 {rows[-1]['l232']:.2e} heads {rows[-1]['head32']:.2e}"
              f" | HIP: median {rows[-1]['medh']:.2e} max {rows[-1]['maxh']:.2e} L2 {rows[-1]['l2h']:.2e} heads {rows[-1]['headh']:.2e}")
    env = {k: max(r[k] for r in rows) for k in ("med32", "max32", "l232")}
    for r in rows:
        assert r["medh"] <= 2 * env["med32"] + 1e-5, r
        assert r["maxh"] <= 2 * env["max32"] + 1e-5, r
        assert r["l2h"] <= 2 * env["l232"] + 1e-5, r
        assert r["headh"] <= 1e-4, r          # no encoder ReLU on the heads' gradient path: strict


def test_run_twice_spread():
    """Determinism (SURVEY.md section 5 "run twice, compare"): the weight gradients are accumulated with fp32 atomics and the
    BatchNorm statistics with fp64 atomics, so the summation order is not fixed and two runs are NOT bit-identical; this test
    quantifies the spread: same weights, same batch, two fresh engines -> hazards within 1e-6, every gradient tensor within 1e-4
    of its maximum (measured: worst tensor 1.1e-6, 338 of 364+ tensors bit-identical)."""
    from multimodal_survival_prediction_amd import losses as HL
    B, dims, rna_dim = 4, (64, 64, 32), 5005
    ct, rna, clin, t, e, mask = _batch(B, dims, rna_dim, 9)
    d = lambda x: x.to(DEV)
    runs = []
    for _ in range(2):
        _, net = _pair("MultiModalSurvivalNet", 3, rna_dim)
        net.train()
        hz = net(d(ct), d(rna), d(clin))
        HL.cox_loss(hz, d(e), d(t)).backward()
        torch.cuda.synchronize()
        runs.append((hz.detach().cpu(), [p.grad.detach().cpu().clone() for p in net.parameters()]))
    assert_close(runs[1][0], runs[0][0], 1e-6, "hazards of two runs")
    gmax = max(float(g.abs().max()) for g in runs[0][1])
    worst, same = 0.0, 0
    for a, b in zip(*[r[1] for r in runs]):
        if float(a.abs().max()) < 1e-5 * gmax:
            continue
        worst = max(worst, rel_err(b, a))
        same += int(torch.equal(a, b))
    print(f"run-twice spread: worst per-tensor gradient difference {worst:.2e}; {same} tensors bit-identical")
    assert worst <= 1e-4
