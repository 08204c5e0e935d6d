"""Pin the CPU oracle against golden vectors produced by the reference's own definitions
(tests/golden/generate_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import losses as OL
from oracle import loops as OLP
from oracle import models as OM

G = os.path.join(os.path.dirname(__file__), "golden")


def _cases(npz):
    return sorted({k[:-2] for k in npz.files if k.endswith("_h")})


def test_cox_numpy_riskset_matches_reference():
    z = np.load(f"{G}/g1_cox.npz")
    cases = _cases(z)
    assert len(cases) == 36
    for c in cases:
        h, e, t = z[c + "_h"], z[c + "_e"], z[c + "_t"]
        loss = OL.cox_npll_np(h, e, t)
        grad = OL.cox_npll_grad_np(h, e, t)
        # reference computes in fp32; tolerance 1e-4 relative (north_star) -- observed ~1e-6
        assert abs(loss - z[c + "_loss"]) <= 1e-4 * max(1.0, abs(loss)), c
        np.testing.assert_allclose(grad, z[c + "_grad"], rtol=1e-4, atol=2e-6, err_msg=c)


def test_cox_torch_restatements_match_reference():
    z = np.load(f"{G}/g1_cox.npz")
    for c in _cases(z):
        h = torch.tensor(z[c + "_h"], requires_grad=True)
        e, t = torch.tensor(z[c + "_e"]), torch.tensor(z[c + "_t"])
        loss = OL.cox_loss(h, e, t)
        assert loss.item() == pytest.approx(float(z[c + "_loss"]), rel=1e-6, abs=1e-7), c
        if loss.grad_fn is not None:
            loss.backward()
            np.testing.assert_allclose(h.grad.numpy(), z[c + "_grad"], rtol=1e-5, atol=1e-7)
            l2 = OL.neg_partial_log_likelihood(h.detach(), e, t).item()
            assert l2 == pytest.approx(loss.item(), rel=2e-5, abs=1e-6)


def test_cindex_matches_reference():
    z = np.load(f"{G}/g2_cindex.npz")
    for n in ("n4", "n23", "n116", "n1639", "n5none"):
        c = OL.concordance_index_np(z[n + "_h"], z[n + "_e"], z[n + "_t"])
        assert c == pytest.approx(float(z[n + "_cindex"]), abs=6e-8), n  # reference returns an fp32 tensor


def _zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0


def _inputs(seed, B, rna_dim, vol):
    rng = np.random.default_rng(100 + seed)
    ct = rng.random((B, 1) + vol, dtype=np.float32)
    rna = rng.normal(0, 1, (B, rna_dim)).astype(np.float32)
    clin = (np.clip(rng.normal(60, 11, (B, 1)), 30, 90) / 100.0).astype(np.float32)
    return torch.tensor(ct), torch.tensor(rna), torch.tensor(clin)


@pytest.mark.parametrize("tag,rna_dim,vol,seed", [("small", 96, (16, 16, 8), 7), ("full", 5005, (64, 64, 32), 11)])
def test_models_fallback_encoder_match_reference(tag, rna_dim, vol, seed):
    z = np.load(f"{G}/g3_models.npz")
    ct, rna, clin = _inputs(seed, 4, rna_dim, vol)
    if tag == "small":
        np.testing.assert_array_equal(ct.numpy(), z["small_ct"])
    e, t, mask = torch.tensor(z[f"{tag}_e"]), torch.tensor(z[f"{tag}_t"]), torch.tensor(z[f"{tag}_mask"])

    def check_gnorm(m, pre):
        refs = {k: float(z[f"{tag}_{pre}_gnorm/{k}"]) for k, _ in m.named_parameters()}
        gmax = max(refs.values())
        for k, p in m.named_parameters():
            ref = refs[k]
            got = float(np.linalg.norm(p.grad.numpy().astype(np.float64)))
            if ref < 1e-5 * gmax:   # exactly-zero gradients (bias feeding a training-mode BN, cox bias): rounding noise that
                assert got < 1e-4 * gmax, (pre, k)          # depends on the host's thread count / summation order
            else:
                assert got == pytest.approx(ref, rel=1e-4), (pre, k)

    # MultiModalSurvivalNet
    torch.manual_seed(seed)
    m = OM.MultiModalSurvivalNet(rna_dim=rna_dim, use_monai=False)
    _zero_dropout(m)
    if tag == "small":  # seed construction reproduces the reference's weights exactly
        for k, v in m.state_dict().items():
            np.testing.assert_array_equal(v.numpy(), z[f"small_mm_sd/{k}"], err_msg=k)
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m(ct, rna, clin).numpy(), z[f"{tag}_mm_eval_hazard"], rtol=1e-5, atol=1e-6)
    m.train()
    hz = m(ct, rna, clin)
    loss = OL.cox_loss(hz, e, t)
    loss.backward()
    np.testing.assert_allclose(hz.detach().numpy(), z[f"{tag}_mm_train_hazard"], rtol=1e-5, atol=1e-6)
    assert loss.item() == pytest.approx(float(z[f"{tag}_mm_train_loss"]), rel=1e-5)
    check_gnorm(m, "mm")
    if tag == "small":
        gmax = max(float(np.abs(z[f"small_mm_grad/{k}"]).max()) for k, _ in m.named_parameters())
        for k, p in m.named_parameters():
            ref = z[f"small_mm_grad/{k}"]
            if float(np.abs(ref).max()) < 1e-5 * gmax:
                assert float(p.grad.abs().max()) < 1e-4 * gmax, k
            else:
                np.testing.assert_allclose(p.grad.numpy(), ref, rtol=1e-3, atol=1e-5 * float(np.abs(ref).max()), err_msg=k)
        for k, v in m.state_dict().items():
            if "running" in k:
                np.testing.assert_allclose(v.numpy(), z[f"small_mm_sd_after/{k}"], rtol=1e-6, atol=1e-7)

    # PartialModalityNet
    torch.manual_seed(seed)
    m = OM.PartialModalityNet(rna_dim=rna_dim, use_monai=False)
    _zero_dropout(m)
    m.eval()
    with torch.no_grad():
        hz, gw = m(ct, rna, clin, mask)
    np.testing.assert_allclose(hz.numpy(), z[f"{tag}_pm_eval_hazard"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gw.numpy(), z[f"{tag}_pm_eval_gate"], rtol=1e-5, atol=1e-6)
    m.train()
    hz, gw = m(ct, rna, clin, mask)
    c_loss, e_loss = OL.cox_loss(hz, e, t), OL.gate_entropy_loss(gw)
    (c_loss + 0.01 * e_loss).backward()
    np.testing.assert_allclose(hz.detach().numpy(), z[f"{tag}_pm_train_hazard"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gw.detach().numpy(), z[f"{tag}_pm_train_gate"], rtol=1e-5, atol=1e-6)
    assert c_loss.item() == pytest.approx(float(z[f"{tag}_pm_cox"]), rel=1e-5)
    assert e_loss.item() == pytest.approx(float(z[f"{tag}_pm_entropy"]), rel=1e-5)
    check_gnorm(m, "pm")

    # SimpleFusionModel
    torch.manual_seed(seed)
    m = OM.SimpleFusionModel(rna_dim=rna_dim, use_monai=False)
    _zero_dropout(m)
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m(ct, rna).numpy(), z[f"{tag}_sf_eval_hazard"], rtol=1e-5, atol=1e-6)
    m.train()
    hz = m(ct, rna)
    loss = OL.neg_partial_log_likelihood(hz, e, t)
    loss.backward()
    np.testing.assert_allclose(hz.detach().numpy(), z[f"{tag}_sf_train_hazard"], rtol=1e-5, atol=1e-6)
    assert loss.item() == pytest.approx(float(z[f"{tag}_sf_train_loss"]), rel=1e-5)
    check_gnorm(m, "sf")


def test_train_epoch_trajectory_matches_reference():
    z = np.load(f"{G}/g5_epoch.npz")
    rng = np.random.default_rng(88)
    N, rna_dim, vol, B = 88, 96, (16, 16, 8), 4
    ct = rng.random((N, 1) + vol, dtype=np.float32)
    rna = rng.normal(0, 1, (N, rna_dim)).astype(np.float32)
    clin = (np.clip(rng.normal(60, 11, (N, 1)), 30, 90) / 100.0).astype(np.float32)
    # same draw order as generate_golden.surv_batch(rng, N, "mixed")
    t = (rng.exponential(1000.0, size=N) + np.arange(N) * 1e-3 + 1.0).astype(np.float32)
    e = (rng.random(N) < 0.57).astype(np.float32)
    rng.normal(0, 1.0, size=N)
    label = np.stack([t, e], 1).astype(np.float32)
    batches = [dict(image=torch.tensor(ct[i:i + B]), rnaseq=torch.tensor(rna[i:i + B]),
                    clinical=torch.tensor(clin[i:i + B]), label=torch.tensor(label[i:i + B]))
               for i in range(0, N, B)]
    torch.manual_seed(5)
    m = OM.MultiModalSurvivalNet(rna_dim=rna_dim, use_monai=False)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-4)
    losses = [OLP.train_epoch_final(m, batches, opt, torch.device("cpu")) for _ in range(2)]
    np.testing.assert_allclose(losses, z["train_losses"], rtol=1e-4)
    val_loss, _ = OLP.validate_final(m, batches[:6], torch.device("cpu"))
    assert val_loss == pytest.approx(float(z["val_loss"]), rel=1e-4)
    m.eval()
    with torch.no_grad():
        hz = m(batches[0]["image"], batches[0]["rnaseq"], batches[0]["clinical"]).numpy()
    np.testing.assert_allclose(hz, z["final_hazard_b0"], rtol=1e-3, atol=1e-5)


def test_densenet121_3d_structure():
    """MONAI DenseNet121-3D restatement: parity unpinned; sanity pins only."""
    from oracle.densenet3d import DenseNet121
    m = DenseNet121(spatial_dims=3, in_channels=1, out_channels=128)
    assert sum(p.numel() for p in m.parameters()) == 11_373_824
    sd = m.state_dict()
    for k in ("features.conv0.weight", "features.norm0.running_mean",
              "features.denseblock1.denselayer1.layers.conv1.weight",
              "features.denseblock4.denselayer16.layers.conv2.weight",
              "features.transition3.conv.weight", "features.norm5.weight", "class_layers.out.bias"):
        assert k in sd
    assert sd["features.denseblock3.denselayer24.layers.norm1.weight"].shape == (992,)
    assert sd["features.transition1.conv.weight"].shape == (128, 256, 1, 1, 1)
    assert tuple(m(torch.rand(2, 1, 32, 32, 32)).shape) == (2, 128)


@pytest.mark.parametrize("tag,B,rna_dim,vol,seed", [("small", 6, 40, (16, 16, 8), 21), ("full", 16, 5005, (32, 32, 16), 22)])
def test_extra_models_match_reference(tag, B, rna_dim, vol, seed):
    """RNASeqSurvivalModel (train_rnaseq_only.py) and FlexibleMultimodalModel (flexible_multimodal.py, fallback encoder)
    against outputs of the reference's own classes (g6_extra_models.npz)."""
    z = np.load(f"{G}/g6_extra_models.npz")
    rng = np.random.default_rng(300 + seed)
    ct = torch.tensor(rng.random((B, 1) + vol, dtype=np.float32))
    rna = torch.tensor(rng.normal(0, 1, (B, rna_dim)).astype(np.float32))
    if tag == "small":
        np.testing.assert_array_equal(ct.numpy(), z["small_ct"]); np.testing.assert_array_equal(rna.numpy(), z["small_rna"])
    e, t, mask = torch.tensor(z[f"{tag}_e"]), torch.tensor(z[f"{tag}_t"]), torch.tensor(z[f"{tag}_mask2"])

    def check(m, pre, fwd):
        refs = {k: float(z[f"{tag}_{pre}_gnorm/{k}"]) for k, _ in m.named_parameters()}
        gmax = max(refs.values())
        m.eval()
        with torch.no_grad():
            np.testing.assert_allclose(fwd(m).numpy(), z[f"{tag}_{pre}_eval"], rtol=1e-5, atol=1e-6)
        m.train()
        hz = fwd(m)
        loss = OL.neg_partial_log_likelihood(hz, e.bool(), t)
        loss.backward()
        np.testing.assert_allclose(hz.detach().numpy(), z[f"{tag}_{pre}_train"], rtol=1e-5, atol=1e-6)
        assert loss.item() == pytest.approx(float(z[f"{tag}_{pre}_loss"]), rel=1e-5)
        for k, p in m.named_parameters():
            got = float(np.linalg.norm(p.grad.numpy().astype(np.float64)))
            if refs[k] < 1e-5 * gmax:
                assert got < 1e-4 * gmax, (pre, k)
            else:
                assert got == pytest.approx(refs[k], rel=1e-4), (pre, k)
        if tag == "small":
            for key in [k for k in z.files if k.startswith(f"small_{pre}_grad/")]:
                k = key.split("/", 1)[1]
                ref = z[key]
                if float(np.abs(ref).max()) >= 1e-5 * gmax:
                    np.testing.assert_allclose(dict(m.named_parameters())[k].grad.numpy(), ref, rtol=2e-4, atol=1e-6 * gmax, err_msg=k)

    torch.manual_seed(seed)
    m = OM.RNASeqSurvivalModel(input_dim=rna_dim) if tag == "full" else OM.RNASeqSurvivalModel(input_dim=rna_dim, hidden_dims=[48, 32, 16])
    _zero_dropout(m)
    if tag == "small":
        for k, v in m.state_dict().items():
            np.testing.assert_array_equal(v.numpy(), z[f"small_rs_sd/{k}"], err_msg=k)
    check(m, "rs", lambda mm: mm(rna).squeeze())

    torch.manual_seed(seed)
    m = OM.FlexibleMultimodalModel(rna_dim=rna_dim, use_monai=False)
    _zero_dropout(m)
    if tag == "small":      # creation order (encoder, rna_encoder, randn biases, fusion) reproduces the reference's draw
        sd = m.state_dict()
        for key in [k for k in z.files if k.startswith("small_fx_sd/")]:
            np.testing.assert_array_equal(sd[key.split("/", 1)[1]].numpy(), z[key], err_msg=key)
    check(m, "fx", lambda mm: mm(ct, rna, mask))


def test_rnaseq_epoch_matches_reference():
    """train_epoch / validate of train_rnaseq_only.py (no gradient clipping, loss / len(loader)): 3 epochs of the oracle loop."""
    from oracle import loops as LO
    z = np.load(f"{G}/g6_extra_models.npz")
    rna, t, e = z["ep_rna"], z["ep_t"], z["ep_e"]
    B = 8
    torch.manual_seed(5)
    m = OM.RNASeqSurvivalModel(input_dim=rna.shape[1], hidden_dims=[32, 16])
    _zero_dropout(m)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-3)
    loader = [dict(rnaseq=torch.tensor(rna[i:i + B]), time=torch.tensor(t[i:i + B]).view(-1, 1),
                   event=torch.tensor(e[i:i + B]).long().view(-1, 1)) for i in range(0, len(t), B)]
    losses = [LO.train_epoch_rnaseq(m, loader, opt) for _ in range(3)]
    np.testing.assert_allclose(losses, z["ep_losses"], rtol=2e-5)
    vl, vc = LO.validate_rnaseq(m, loader)
    np.testing.assert_allclose([vl, vc], z["ep_val"], rtol=2e-5)
