"""HIP path vs golden vectors produced by the REFERENCE's own class definitions (tests/golden/g3_models.npz,
generate_golden.py): the three networks with the reference's fallback CT encoder (the branch it takes without MONAI),
eval hazards, train hazards / gate weights, losses, gradients -- no oracle in the loop."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV, assert_close

G = os.path.join(os.path.dirname(__file__), "golden")


def _inputs(seed, B, rna_dim, vol):
    rng = np.random.default_rng(100 + seed)
    ct = rng.random((B, 1) + vol, dtype=np.float32)
    rna = rng.normal(0, 1, (B, rna_dim)).astype(np.float32)
    clin = (np.clip(rng.normal(60, 11, (B, 1)), 30, 90) / 100.0).astype(np.float32)
    return torch.tensor(ct).to(DEV), torch.tensor(rna).to(DEV), torch.tensor(clin).to(DEV)


def _zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0


@pytest.fixture()
def fallback_models():
    from multimodal_survival_prediction_amd import models
    old = models.USE_MONAI
    models.USE_MONAI = False
    yield models
    models.USE_MONAI = old


@pytest.mark.parametrize("tag,rna_dim,vol,seed", [("small", 96, (16, 16, 8), 7), ("full", 5005, (64, 64, 32), 11)])
def test_hip_models_match_reference_goldens(fallback_models, tag, rna_dim, vol, seed):
    from multimodal_survival_prediction_amd import losses as HL
    M = fallback_models
    z = np.load(f"{G}/g3_models.npz")
    ct, rna, clin = _inputs(seed, 4, rna_dim, vol)
    e, t, mask = (torch.tensor(z[f"{tag}_{k}"]).to(DEV) for k in ("e", "t", "mask"))

    def check_gnorm(m, pre, tol=2e-4):
        refs = {k: float(z[f"{tag}_{pre}_gnorm/{k}"]) for k, _ in m.named_parameters()}
        gmax = max(refs.values())
        for k, p in m.named_parameters():
            ref = refs[k]
            got = float(np.linalg.norm(p.grad.detach().cpu().numpy().astype(np.float64)))
            if ref < 1e-5 * gmax:     # exactly-zero gradients (conv/linear bias feeding a training-mode BN, cox bias): noise
                assert got < 1e-4 * gmax, (pre, k, got, ref)
            else:
                assert abs(got - ref) <= tol * ref, (pre, k, got, ref)

    # ---- MultiModalSurvivalNet: weights from seed construction (same creation order as the reference class)
    torch.manual_seed(seed)
    m = M.MultiModalSurvivalNet(rna_dim=rna_dim)
    assert m.use_monai is False and list(m.ct_encoder.state_dict())[0] == "0.weight"
    _zero_dropout(m)
    if tag == "small":
        for k, v in m.state_dict().items():
            np.testing.assert_array_equal(v.numpy(), z[f"small_mm_sd/{k}"], err_msg=k)
    m.to(DEV).eval()
    with torch.no_grad():
        assert_close(m(ct, rna, clin), torch.tensor(z[f"{tag}_mm_eval_hazard"]), 1e-4, "mm eval hazard")
    m.train()
    hz = m(ct, rna, clin)
    loss = HL.cox_loss(hz, e, t)
    loss.backward()
    torch.cuda.synchronize()
    assert_close(hz, torch.tensor(z[f"{tag}_mm_train_hazard"]), 1e-4, "mm train hazard")
    assert abs(loss.item() - float(z[f"{tag}_mm_train_loss"])) <= 1e-4 * max(1.0, abs(loss.item()))
    check_gnorm(m, "mm")
    if tag == "small":
        gmax = max(float(np.abs(z[f"small_mm_grad/{k}"]).max()) for k, _ in m.named_parameters())
        for k, p in m.named_parameters():
            ref = torch.tensor(z[f"small_mm_grad/{k}"])
            if float(ref.abs().max()) < 1e-5 * gmax:        # exactly-zero gradients (bias before BN, cox_head.bias): noise
                assert float(p.grad.abs().max()) < 1e-4 * gmax, k
            else:
                assert_close(p.grad, ref, 1e-4, k)
        for k, v in m.state_dict().items():
            if "running" in k:
                assert_close(v, torch.tensor(z[f"small_mm_sd_after/{k}"]), 1e-4, k)

    # ---- PartialModalityNet
    torch.manual_seed(seed)
    m = M.PartialModalityNet(rna_dim=rna_dim)
    _zero_dropout(m)
    m.to(DEV).eval()
    with torch.no_grad():
        hz, gw = m(ct, rna, clin, mask)
    assert_close(hz, torch.tensor(z[f"{tag}_pm_eval_hazard"]), 1e-4, "pm eval hazard")
    assert_close(gw, torch.tensor(z[f"{tag}_pm_eval_gate"]), 1e-4, "pm eval gate")
    m.train()
    hz, gw = m(ct, rna, clin, mask)
    c_loss, e_loss = HL.cox_loss(hz, e, t), HL.gate_entropy_loss(gw)
    (c_loss + 0.01 * e_loss).backward()
    torch.cuda.synchronize()
    assert_close(hz, torch.tensor(z[f"{tag}_pm_train_hazard"]), 1e-4, "pm train hazard")
    assert_close(gw, torch.tensor(z[f"{tag}_pm_train_gate"]), 1e-4, "pm train gate")
    assert abs(c_loss.item() - float(z[f"{tag}_pm_cox"])) <= 1e-4 and abs(e_loss.item() - float(z[f"{tag}_pm_entropy"])) <= 1e-4
    check_gnorm(m, "pm")

    # ---- SimpleFusionModel
    torch.manual_seed(seed)
    m = M.SimpleFusionModel(rna_dim=rna_dim)
    _zero_dropout(m)
    m.to(DEV).eval()
    with torch.no_grad():
        assert_close(m(ct, rna), torch.tensor(z[f"{tag}_sf_eval_hazard"]), 1e-4, "sf eval hazard")
    m.train()
    hz = m(ct, rna)
    loss = HL.neg_partial_log_likelihood(hz, e, t)
    loss.backward()
    torch.cuda.synchronize()
    assert_close(hz, torch.tensor(z[f"{tag}_sf_train_hazard"]), 1e-4, "sf train hazard")
    assert abs(loss.item() - float(z[f"{tag}_sf_train_loss"])) <= 1e-4 * max(1.0, abs(loss.item()))
    check_gnorm(m, "sf")


def test_fused_step_runs_with_fallback_encoder(fallback_models):
    """odd volume sizes are fine for the fallback encoder (ceil-halving), and the fused HIP-graph step trains it."""
    from multimodal_survival_prediction_amd import data
    from multimodal_survival_prediction_amd.training import FusedOptimizer
    torch.manual_seed(0)
    m = fallback_models.MultiModalSurvivalNet(rna_dim=64).to(DEV)
    opt = FusedOptimizer(m, lr=1e-3, weight_decay=1e-4)
    c = data.make_cohort(n=8, dims=(20, 12, 10), rna_dim=64, seed=3)
    m.train()
    losses = []
    for it in range(6):
        opt.engine.reset_epoch_stats()
        opt.engine.train_step(c["image"], c["rnaseq"], c["clinical"], time=c["label"][:, 0], event=c["label"][:, 1])
        losses.append(opt.engine.epoch_stats()["sum_loss"])
    assert losses[-1] < losses[0], losses
