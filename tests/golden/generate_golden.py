#!/usr/bin/env python3
"""Generate golden input/output vectors by executing the REFERENCE's own definitions.

Runs only in the build container (needs /root/reference); the fixtures it writes
(`tests/golden/*.npz`) are data only -- seeds, inputs, expected outputs -- and are what
travels to the GPU box.  No reference source is copied into the repo: the reference
scripts are read as text from /root/reference at generation time and executed in a
scratch namespace / scratch cwd.

How each reference file is harvested:
  * scripts/training/final_multimodal.py   exec'd whole; stops with FileNotFoundError at
    its first CSV read (:205) AFTER MultiModalSurvivalNet (:59), custom cox_loss (:171)
    and calculate_cindex (:188) exist; train_epoch/validate (:238-305) are then exec'd by line range
    into that namespace.  MONAI/torchsurv are absent here, so the fallback
    3-conv encoder and the custom Cox loss are what is defined.
  * scripts/training/train_rnaseq_only.py  exec'd whole; stops at :217.  Gives fallback
    neg_partial_log_likelihood (:40) and the O(n^2) ConcordanceIndex (:55).
  * scripts/training/partial_modality_training.py / simple_fusion.py import SimpleITK
    before any class is defined, so the class blocks are exec'd by line range
    (partial :165-331 -> PartialModalityNet, cox_loss, gate_entropy_loss;
     simple  :46-73,160-236 -> fallback NPLL, ConcordanceIndex, SimpleFusionModel)
    with USE_MONAI = USE_TORCHSURV = False, exactly the branch an install without
    MONAI/torchsurv would take.

Usage:  python tests/golden/generate_golden.py   (from the repo root)
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference/scripts/training"
OUT = os.path.dirname(os.path.abspath(__file__))


def _exec_until_data(path):
    """exec a reference script in a scratch cwd; return its namespace at the first data access."""
    src = open(path, encoding="utf-8").read()
    ns = {"__name__": "__ref__", "__file__": path}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                exec(compile(src, path, "exec"), ns)
        except FileNotFoundError:
            pass
        finally:
            os.chdir(cwd)
    return ns


def _exec_lines(path, ranges, ns):
    lines = open(path, encoding="utf-8").read().split("\n")
    for lo, hi in ranges:
        # keep original line numbers for tracebacks
        src = "\n" * (lo - 1) + "\n".join(lines[lo - 1:hi])
        with contextlib.redirect_stdout(io.StringIO()):
            exec(compile(src, path, "exec"), ns)
    return ns


def harvest():
    fm = _exec_until_data(f"{REF}/final_multimodal.py")
    # train_epoch/validate sit after the CSV read (:238-305): exec'd by line range into the same namespace
    _exec_lines(f"{REF}/final_multimodal.py", [(238, 305)], fm)
    rn = _exec_until_data(f"{REF}/train_rnaseq_only.py")
    # train_rnaseq_only.py defines its model and train_epoch/validate before the data access: rn holds RNASeqSurvivalModel too
    base = {"torch": torch, "nn": nn, "np": np, "USE_MONAI": False, "USE_TORCHSURV": False}
    pm = _exec_lines(f"{REF}/partial_modality_training.py", [(165, 331)], dict(base))
    sf = _exec_lines(f"{REF}/simple_fusion.py", [(46, 73), (160, 236)], dict(base))
    return fm, rn, pm, sf


# ----------------------------------------------------------------------------------------------
# synthetic inputs (all from numpy default_rng so the tests can regenerate them without torch RNG)
# ----------------------------------------------------------------------------------------------

def surv_batch(rng, n, mode):
    """distinct survival times (no ties) + event pattern `mode`."""
    t = rng.exponential(1000.0, size=n) + np.arange(n) * 1e-3 + 1.0
    t = t.astype(np.float32)
    assert len(np.unique(t)) == n
    if mode == "all":
        e = np.ones(n, np.float32)
    elif mode == "none":
        e = np.zeros(n, np.float32)
    elif mode == "one":
        e = np.zeros(n, np.float32)
        e[rng.integers(n)] = 1
    elif mode == "maxtime":  # the single event sits at the largest time
        e = np.zeros(n, np.float32)
        e[np.argmax(t)] = 1
    else:
        e = (rng.random(n) < 0.57).astype(np.float32)
        if e.sum() == 0:
            e[0] = 1
    h = rng.normal(0, 1.0, size=n).astype(np.float32)
    return h, e, t


def gen_cox(fm, rn, sf):
    out = {}
    rng = np.random.default_rng(1234)
    cases = []
    for n in (1, 2, 3, 4, 8, 16, 64, 2048):
        for mode in ("all", "mixed", "one", "none", "maxtime"):
            if n == 1 and mode != "all":
                continue
            cases.append((n, mode))
    for ci, (n, mode) in enumerate(cases):
        h, e, t = surv_batch(rng, n, mode)
        key = f"c{ci:02d}_n{n}_{mode}"
        ht = torch.tensor(h, requires_grad=True)
        loss = fm["cox_loss"](ht, torch.tensor(e), torch.tensor(t))
        if loss.grad_fn is not None:
            loss.backward()
            g = ht.grad.numpy()
        else:  # degenerate batch: fresh leaf, no graph (final_multimodal.py:173-176)
            g = np.zeros_like(h)
        out[key + "_h"], out[key + "_e"], out[key + "_t"] = h, e, t
        out[key + "_loss"] = np.float32(loss.item())
        out[key + "_grad"] = g.astype(np.float32)
        # the two other in-file formulations must agree on these no-tie batches
        if e.sum() > 0 and n >= 2:
            l2 = rn["neg_partial_log_likelihood"](torch.tensor(h), torch.tensor(e), torch.tensor(t)).item()
            l3 = sf["neg_partial_log_likelihood"](torch.tensor(h), torch.tensor(e), torch.tensor(t)).item()
            assert abs(l2 - loss.item()) <= 2e-5 * max(1, abs(l2)), (key, l2, loss.item())
            assert abs(l3 - loss.item()) <= 2e-5 * max(1, abs(l3)), (key, l3, loss.item())
    np.savez_compressed(f"{OUT}/g1_cox.npz", **out)
    print("g1_cox:", len(cases), "cases")


def gen_cindex(rn, sf):
    out = {}
    rng = np.random.default_rng(4321)
    for n in (4, 23, 116, 1639):
        h, e, t = surv_batch(rng, n, "mixed")
        c = rn["ConcordanceIndex"]()(torch.tensor(h), torch.tensor(e), torch.tensor(t)).item()
        if n <= 116:
            c2 = sf["ConcordanceIndex"]()(torch.tensor(h), torch.tensor(e), torch.tensor(t)).item()
            assert c == c2
        out[f"n{n}_h"], out[f"n{n}_e"], out[f"n{n}_t"] = h, e, t
        out[f"n{n}_cindex"] = np.float64(c)
    # no permissible pair -> 0.5 (train_rnaseq_only.py:70)
    h, e, t = surv_batch(rng, 5, "none")
    out["n5none_h"], out["n5none_e"], out["n5none_t"] = h, e, t
    out["n5none_cindex"] = np.float64(rn["ConcordanceIndex"]()(torch.tensor(h), torch.tensor(e), torch.tensor(t)).item())
    np.savez_compressed(f"{OUT}/g2_cindex.npz", **out)
    print("g2_cindex done")


def _zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0


def model_inputs(rng, B, rna_dim, vol):
    ct = rng.random((B, 1) + vol, dtype=np.float32)
    rna = rng.normal(0, 1, (B, rna_dim)).astype(np.float32)
    clin = (np.clip(rng.normal(60, 11, (B, 1)), 30, 90) / 100.0).astype(np.float32)
    return ct, rna, clin


def _grads(model):
    return {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()}


def gen_models(fm, pm, sf):
    """G3: the three classes with the reference's fallback CT encoder.

    small variant (rna_dim=96, volume 16x16x8): inputs + outputs + per-parameter grad norms for all three;
    full state_dict + all grads for MultiModalSurvivalNet, head grads for the other two (weights of those
    come from torch.manual_seed(seed) construction, as in the full variant).
    full variant (rna_dim=5005, volume 64x64x32): weights come from torch.manual_seed(seed) construction,
    only inputs' seed, outputs and per-parameter grad norms are stored.
    """
    out = {}
    for tag, rna_dim, vol, B, seed in (("small", 96, (16, 16, 8), 4, 7), ("full", 5005, (64, 64, 32), 4, 11)):
        rng = np.random.default_rng(100 + seed)
        ct, rna, clin = model_inputs(rng, B, rna_dim, vol)
        h0, e, t = surv_batch(rng, B, "mixed")
        mask = np.array([[1, 1, 1], [0, 1, 1], [1, 0, 1], [0, 1, 0]], np.float32)[:B]
        out[f"{tag}_e"], out[f"{tag}_t"], out[f"{tag}_mask"] = e, t, mask
        if tag == "small":
            out["small_ct"], out["small_rna"], out["small_clin"] = ct, rna, clin
        tct, trna, tclin = torch.tensor(ct), torch.tensor(rna), torch.tensor(clin)
        te, tt, tmask = torch.tensor(e), torch.tensor(t), torch.tensor(mask)

        # ---- MultiModalSurvivalNet (final_multimodal.py:59) ----
        torch.manual_seed(seed)
        m = fm["MultiModalSurvivalNet"](rna_dim=rna_dim)
        _zero_dropout(m)
        sd0 = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
        m.eval()
        with torch.no_grad():
            out[f"{tag}_mm_eval_hazard"] = m(tct, trna, tclin).numpy()
        m.train()
        hz = m(tct, trna, tclin)
        loss = fm["cox_loss"](hz, te, tt)
        loss.backward()
        out[f"{tag}_mm_train_hazard"] = hz.detach().numpy()
        out[f"{tag}_mm_train_loss"] = np.float32(loss.item())
        g = _grads(m)
        sd1 = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
        for k in g:
            out[f"{tag}_mm_gnorm/{k}"] = np.float64(np.linalg.norm(g[k].astype(np.float64)))
        if tag == "small":
            for k in sd0:
                out[f"small_mm_sd/{k}"] = sd0[k]
            for k in g:
                out[f"small_mm_grad/{k}"] = g[k]
            for k in sd1:  # BN running stats after the one train-mode forward
                if "running" in k:
                    out[f"small_mm_sd_after/{k}"] = sd1[k]

        # ---- PartialModalityNet (partial_modality_training.py:165) ----
        torch.manual_seed(seed)
        m = pm["PartialModalityNet"](rna_dim=rna_dim)
        _zero_dropout(m)
        sd0 = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
        m.eval()
        with torch.no_grad():
            hz, gw = m(tct, trna, tclin, tmask)
            out[f"{tag}_pm_eval_hazard"], out[f"{tag}_pm_eval_gate"] = hz.numpy(), gw.numpy()
        m.train()
        hz, gw = m(tct, trna, tclin, tmask)
        c_loss = pm["cox_loss"](hz, te, tt)
        e_loss = pm["gate_entropy_loss"](gw)
        loss = c_loss + 0.01 * e_loss  # partial_modality_training.py:422
        loss.backward()
        out[f"{tag}_pm_train_hazard"], out[f"{tag}_pm_train_gate"] = hz.detach().numpy(), gw.detach().numpy()
        out[f"{tag}_pm_cox"], out[f"{tag}_pm_entropy"] = np.float32(c_loss.item()), np.float32(e_loss.item())
        g = _grads(m)
        for k in g:
            out[f"{tag}_pm_gnorm/{k}"] = np.float64(np.linalg.norm(g[k].astype(np.float64)))
        if tag == "small":  # heads only (encoder weights are pinned by the mm variant + seed construction)
            for k in g:
                if not k.startswith("ct_encoder") and not k.startswith("rna_encoder.0"):
                    out[f"small_pm_grad/{k}"] = g[k]

        # ---- SimpleFusionModel (simple_fusion.py:160) ----
        torch.manual_seed(seed)
        m = sf["SimpleFusionModel"](rna_dim=rna_dim)
        _zero_dropout(m)
        sd0 = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
        m.eval()
        with torch.no_grad():
            out[f"{tag}_sf_eval_hazard"] = m(tct, trna).numpy()
        m.train()
        hz = m(tct, trna)
        loss = sf["neg_partial_log_likelihood"](hz, te, tt)
        loss.backward()
        out[f"{tag}_sf_train_hazard"] = hz.detach().numpy()
        out[f"{tag}_sf_train_loss"] = np.float32(loss.item())
        g = _grads(m)
        for k in g:
            out[f"{tag}_sf_gnorm/{k}"] = np.float64(np.linalg.norm(g[k].astype(np.float64)))
        if tag == "small":
            for k in g:
                if k.startswith("fusion"):
                    out[f"small_sf_grad/{k}"] = g[k]
    np.savez_compressed(f"{OUT}/g3_models.npz", **out)
    print("g3_models done:", len(out), "arrays")


def gen_extra_models(rn):
    """g6: RNASeqSurvivalModel (train_rnaseq_only.py:126-151, harvested whole) and FlexibleMultimodalModel
    (flexible_multimodal.py:157-256, class block exec'd by line range with USE_MONAI = False: SimpleITK is imported at :38,
    before any class exists).  Dropout forced to 0; weights from torch.manual_seed construction; small variants store the
    full state_dict and gradients, default-width variants only outputs and gradient norms."""
    out = {}
    fx = _exec_lines(f"{REF}/flexible_multimodal.py", [(157, 256)], {"torch": torch, "nn": nn, "np": np, "USE_MONAI": False})
    for tag, seed, B, rna_dim, vol in (("small", 21, 6, 40, (16, 16, 8)), ("full", 22, 16, 5005, (32, 32, 16))):
        rng = np.random.default_rng(300 + seed)
        ct = rng.random((B, 1) + vol, dtype=np.float32)
        rna = rng.normal(0, 1, (B, rna_dim)).astype(np.float32)
        h0, e, t = surv_batch(rng, B, "mixed")
        mask = (rng.random((B, 2)) < 0.6).astype(np.float32)
        mask[0] = (1, 1); mask[1] = (0, 1); mask[2] = (1, 0); mask[3] = (0, 0)
        out[f"{tag}_e"], out[f"{tag}_t"], out[f"{tag}_mask2"] = e, t, mask
        if tag == "small":
            out["small_ct"], out["small_rna"] = ct, rna
        tct, trna, te, tt, tmask = torch.tensor(ct), torch.tensor(rna), torch.tensor(e), torch.tensor(t), torch.tensor(mask)
        # ---- RNASeqSurvivalModel ----
        torch.manual_seed(seed)
        m = rn["RNASeqSurvivalModel"](input_dim=rna_dim) if tag == "full" else rn["RNASeqSurvivalModel"](input_dim=rna_dim, hidden_dims=[48, 32, 16])
        _zero_dropout(m)
        sd0 = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
        m.eval()
        with torch.no_grad():
            out[f"{tag}_rs_eval"] = m(trna).squeeze().numpy()
        m.train()
        hz = m(trna).squeeze()
        loss = rn["neg_partial_log_likelihood"](hz, te.long(), tt)
        loss.backward()
        out[f"{tag}_rs_train"], out[f"{tag}_rs_loss"] = hz.detach().numpy(), np.float32(loss.item())
        g = _grads(m)
        for k in g:
            out[f"{tag}_rs_gnorm/{k}"] = np.float64(np.linalg.norm(g[k].astype(np.float64)))
        if tag == "small":
            for k in sd0:
                out[f"small_rs_sd/{k}"] = sd0[k]
            for k in g:
                out[f"small_rs_grad/{k}"] = g[k]
        # ---- FlexibleMultimodalModel (fallback encoder branch) ----
        torch.manual_seed(seed)
        m = fx["FlexibleMultimodalModel"](rna_dim=rna_dim)
        _zero_dropout(m)
        sd0 = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
        m.eval()
        with torch.no_grad():
            out[f"{tag}_fx_eval"] = m(tct, trna, tmask).numpy()
        m.train()
        hz = m(tct, trna, tmask)
        loss = rn["neg_partial_log_likelihood"](hz, te.long(), tt)
        loss.backward()
        out[f"{tag}_fx_train"], out[f"{tag}_fx_loss"] = hz.detach().numpy(), np.float32(loss.item())
        g = _grads(m)
        for k in g:
            out[f"{tag}_fx_gnorm/{k}"] = np.float64(np.linalg.norm(g[k].astype(np.float64)))
        if tag == "small":     # a few whole tensors (creation-order / RNG pin + element-wise gradients); the rest as norms above
            keep = [k for k in sd0 if k.startswith("missing_") or k.startswith("fusion.7") or k.startswith("rna_encoder.8.bias")
                    or k.startswith("image_encoder.0.")]
            for k in keep:
                out[f"small_fx_sd/{k}"] = sd0[k]
                if k in g:
                    out[f"small_fx_grad/{k}"] = g[k]
    # one epoch of the reference's own train_epoch (train_rnaseq_only.py:157-176: no clipping, loss / len(loader))
    rng = np.random.default_rng(77)
    n, B, rna_dim = 40, 8, 24
    rna = rng.normal(0, 1, (n, rna_dim)).astype(np.float32)
    t = (rng.exponential(1000.0, n) + 1 + np.arange(n) * 1e-3).astype(np.float32)
    e = (rng.random(n) < 0.6).astype(np.int64)
    torch.manual_seed(5)
    m = rn["RNASeqSurvivalModel"](input_dim=rna_dim, hidden_dims=[32, 16])
    _zero_dropout(m)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-3)
    loader = [dict(rnaseq=torch.tensor(rna[i:i + B]), time=torch.tensor(t[i:i + B]).view(-1, 1),
                   event=torch.tensor(e[i:i + B]).view(-1, 1)) for i in range(0, n, B)]
    losses = [rn["train_epoch"](m, loader, opt, torch.device("cpu")) for _ in range(3)]
    vl, vc = rn["validate"](m, loader, torch.device("cpu"))
    out["ep_rna"], out["ep_t"], out["ep_e"] = rna, t, e.astype(np.float32)
    out["ep_losses"], out["ep_val"] = np.array(losses, np.float64), np.array([vl, vc], np.float64)
    np.savez_compressed(f"{OUT}/g6_extra_models.npz", **out)
    print("g6_extra_models done:", len(out), "arrays")


def gen_epoch(fm):
    """G5: one epoch of the reference's own train_epoch/validate (final_multimodal.py:238-305) on a seeded
    synthetic 88-patient cohort, config-1 style (small volume so it runs in seconds; dropout ACTIVE, so the
    trajectory also pins RNG consumption order of the restated model)."""
    rng = np.random.default_rng(88)
    N, rna_dim, vol, B = 88, 96, (16, 16, 8), 4
    ct, rna, clin = model_inputs(rng, N, rna_dim, vol)
    _, e, t = surv_batch(rng, N, "mixed")
    label = np.stack([t, e], 1).astype(np.float32)
    batches = [dict(image=torch.tensor(ct[i:i + B]), rnaseq=torch.tensor(rna[i:i + B]),
                    clinical=torch.tensor(clin[i:i + B]), label=torch.tensor(label[i:i + B]))
               for i in range(0, N, B)]
    torch.manual_seed(5)
    m = fm["MultiModalSurvivalNet"](rna_dim=rna_dim)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-4)  # final_multimodal.py:350
    losses = []
    for ep in range(2):
        losses.append(fm["train_epoch"](m, batches, opt, torch.device("cpu")))
    val_loss, _ = fm["validate"](m, batches[:6], torch.device("cpu"))
    m.eval()
    with torch.no_grad():
        hz = m(batches[0]["image"], batches[0]["rnaseq"], batches[0]["clinical"]).numpy()
    np.savez_compressed(f"{OUT}/g5_epoch.npz", train_losses=np.array(losses, np.float64),
                        val_loss=np.float64(val_loss), final_hazard_b0=hz,
                        cohort_seed=np.int64(88), model_seed=np.int64(5))
    print("g5_epoch:", losses, val_loss)


if __name__ == "__main__":
    torch.set_num_threads(8)
    fm, rn, pm, sf = harvest()
    for name, ns, keys in (("final_multimodal", fm, ["MultiModalSurvivalNet", "cox_loss", "calculate_cindex", "train_epoch", "validate"]),
                           ("train_rnaseq_only", rn, ["neg_partial_log_likelihood", "ConcordanceIndex", "RNASeqSurvivalModel", "train_epoch", "validate"]),
                           ("partial_modality_training", pm, ["PartialModalityNet", "cox_loss", "gate_entropy_loss"]),
                           ("simple_fusion", sf, ["SimpleFusionModel", "neg_partial_log_likelihood", "ConcordanceIndex"])):
        missing = [k for k in keys if k not in ns]
        assert not missing, (name, missing)
    gen_cox(fm, rn, sf)
    gen_cindex(rn, sf)
    gen_models(fm, pm, sf)
    gen_extra_models(rn)
    gen_epoch(fm)
