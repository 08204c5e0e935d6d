"""On-disk cohort contract of the reference + a GPU-side loader.

The reference's datasets read, relative to the cwd (partial_modality_training.py:60-158, simple_fusion.py:98-154,
flexible_multimodal.py:96-151, create_full_matching_table.py:124-134):
    data/processed/full_matching_table.csv    patient_id, nifti_path, has_imaging, has_rnaseq, has_clinical, age,
                                              survival_time, survival_status, has_survival
    data/processed/rnaseq_normalized_mapped.csv    patient-indexed, one column per gene (5005)
and, per item and on the CPU, look the patient's row up in a DataFrame, read the NIfTI volume, min-max normalise it and
scipy.zoom it (order 1) to 64x64x32 -- O(cohort) pandas work plus a resample per sample per epoch, with num_workers = 0.

`write_cohort` lays a synthetic cohort down in that contract; `load_cohort` reads the contract ONCE into an HBM-resident
tensor store (the dict `data.make_cohort` produces), doing the volume preprocessing on the GPU (`mms_ct_preprocess`).
Volumes are stored as .npy at `nifti_path` (NIfTI needs SimpleITK/nibabel, absent in this image); everything else is the
reference's layout, so its own CSV tooling reads these files.
"""
import os

import numpy as np
import torch

from . import _lib, ops

TABLE = os.path.join("data", "processed", "full_matching_table.csv")
RNASEQ = os.path.join("data", "processed", "rnaseq_normalized_mapped.csv")
COLUMNS = ["patient_id", "nifti_path", "has_imaging", "has_rnaseq", "has_clinical", "age", "survival_time", "survival_status",
           "has_survival"]


def write_cohort(root, cohort, native_dims=None, seed=0):
    """cohort: CPU dict from data.make_cohort.  native_dims: store each volume resampled to this (D, H, W) -- i.e. at a
    'scanner' resolution different from the network grid -- so that loading exercises the resample; None = network grid.
    Raw intensities get a per-patient offset/scale (the loader's min-max must undo it)."""
    import pandas as pd
    import torch.nn.functional as F
    n = cohort["n"]
    os.makedirs(os.path.join(root, "data", "processed"), exist_ok=True)
    os.makedirs(os.path.join(root, "data", "volumes"), exist_ok=True)
    rng = np.random.default_rng(seed)
    mask = cohort["mask"].numpy().astype(bool)
    has_surv = cohort["has_survival"].numpy().astype(bool)
    rows, rna_rows, rna_ids = [], [], []
    for i in range(n):
        pid = "TCGA-SYN-%04d" % i
        path = None
        if mask[i, 0]:
            v = cohort["image"][i:i + 1]
            if native_dims is not None:
                v = F.interpolate(v, size=tuple(native_dims), mode="trilinear", align_corners=True)
            v = v[0, 0].numpy() * float(rng.uniform(500, 3000)) + float(rng.uniform(-1000, 0))      # HU-like range
            path = os.path.join(root, "data", "volumes", pid + ".npy")
            np.save(path, v.astype(np.float32))
        if mask[i, 1]:
            rna_ids.append(pid); rna_rows.append(cohort["rnaseq"][i].numpy())
        lab = cohort["label"][i].numpy()
        rows.append([pid, path, bool(mask[i, 0]), bool(mask[i, 1]), bool(mask[i, 2]),
                     float(cohort["clinical"][i, 0]) * 100.0 if mask[i, 2] else np.nan,
                     float(lab[0]) if has_surv[i] else np.nan, int(lab[1]) if has_surv[i] else np.nan, bool(has_surv[i])])
    pd.DataFrame(rows, columns=COLUMNS).to_csv(os.path.join(root, TABLE), index=False)
    g = cohort["rnaseq"].shape[1]
    pd.DataFrame(np.stack(rna_rows) if rna_rows else np.zeros((0, g), np.float32), index=rna_ids,
                 columns=["ENSG%011d" % j for j in range(g)]).to_csv(os.path.join(root, RNASEQ), float_format="%.9g")
    return os.path.join(root, TABLE)


def ct_preprocess(vol_dev, target_size, out=None, scratch=None):
    """vol_dev: (D, H, W) fp32 device tensor at its native resolution -> (tD, tH, tW): min-max normalise + order-1 resample."""
    if not vol_dev.is_cuda:
        raise RuntimeError("ct_preprocess runs on the MI355X (no CPU fallback)")
    v = vol_dev.contiguous().float()
    if out is None:
        out = torch.empty(tuple(target_size), device=v.device)
    if scratch is None:
        scratch = torch.empty(512, device=v.device)
    _lib.check(_lib.load_library().mms_ct_preprocess(v.data_ptr(), v.shape[0], v.shape[1], v.shape[2], out.data_ptr(),
                                                    target_size[0], target_size[1], target_size[2], scratch.data_ptr(), ops.stream()),
               "mms_ct_preprocess")
    return out


def rna_log_zscore(counts_dev):
    """counts_dev: (n, genes) fp32 device tensor of raw counts -> z-scored log2(count + 1) (preprocess_genomic.py:108-117)."""
    if not counts_dev.is_cuda:
        raise RuntimeError("rna_log_zscore runs on the MI355X (no CPU fallback)")
    c = counts_dev.contiguous().float()
    out = torch.empty_like(c)
    _lib.check(_lib.load_library().mms_rna_log_zscore(c.data_ptr(), out.data_ptr(), c.shape[0], c.shape[1], ops.stream()),
               "mms_rna_log_zscore")
    return out


def read_tables(root):
    """-> (matching table DataFrame, rnaseq DataFrame or None): the host side of the contract (CPU only)."""
    import pandas as pd
    mt = pd.read_csv(os.path.join(root, TABLE))
    missing = [c for c in COLUMNS if c not in mt.columns]
    if missing:
        raise ValueError("full_matching_table.csv lacks columns %s" % missing)
    rp = os.path.join(root, RNASEQ)
    rn = pd.read_csv(rp, index_col=0) if os.path.exists(rp) else None
    return mt, rn


def load_cohort(root, device, target_size=(64, 64, 32), rna_dim=None):
    """The reference datasets' __getitem__ for every patient, once: -> the HBM-resident cohort dict of data.make_cohort
    (image zeros / rnaseq zeros / clinical 0 where the modality is missing, mask, label, has_survival, patient_id)."""
    import pandas as pd
    mt, rn = read_tables(root)
    n = len(mt)
    g = rn.shape[1] if rn is not None else (rna_dim or 5005)
    dev = torch.device(device)
    image = torch.zeros(n, 1, *target_size, device=dev)
    rna = torch.zeros(n, g)
    clin = torch.zeros(n, 1)
    label = torch.zeros(n, 2)
    mask = torch.zeros(n, 3)
    has_surv = torch.zeros(n, dtype=torch.bool)
    scratch = torch.empty(512, device=dev)
    rn_pos = {pid: i for i, pid in enumerate(rn.index)} if rn is not None else {}
    rn_vals = torch.from_numpy(rn.values.astype(np.float32)) if rn is not None else None
    for i, row in enumerate(mt.itertuples(index=False)):
        p = row.nifti_path
        if isinstance(p, str) and os.path.exists(p):            # pd.notna(nifti_path) and os.path.exists (:92)
            vol = torch.from_numpy(np.load(p).astype(np.float32)).to(dev)
            ct_preprocess(vol, target_size, out=image[i, 0], scratch=scratch)
            mask[i, 0] = 1
        if row.patient_id in rn_pos:
            rna[i] = rn_vals[rn_pos[row.patient_id]]
            mask[i, 1] = 1
        if pd.notna(row.age):
            clin[i, 0] = row.age / 100.0
            mask[i, 2] = 1
        if pd.notna(row.survival_time):
            label[i, 0], label[i, 1] = float(row.survival_time), int(row.survival_status)
            has_surv[i] = True
    # real survival times repeat (days); the fused step's Cox loss then follows losses.USE_TORCHSURV (Efron, torchsurv's rule) -- with
    # the switch off it is the Breslow risk-set form, whereas the reference's in-file fallback is order-dependent under ties
    t_lab = label[has_surv, 0]
    n_dup = int(t_lab.numel() - torch.unique(t_lab).numel())
    if n_dup:
        import warnings
        from . import losses
        warnings.warn("cohort has %d repeated survival times among %d labelled patients: tie handling = %s (losses.USE_TORCHSURV = %s)"
                      % (n_dup, int(t_lab.numel()), losses.default_ties(), losses.USE_TORCHSURV))
    return dict(image=image, rnaseq=rna.to(dev), clinical=clin.to(dev), label=label.to(dev), mask=mask.to(dev),
                has_survival=has_surv.to(dev), n=n, dims=tuple(target_size), patient_id=list(mt["patient_id"]), tied_times=n_dup)
