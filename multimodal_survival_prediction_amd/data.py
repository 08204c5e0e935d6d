"""Synthetic cohorts shaped like the reference's TCGA-OV data and in-memory batch loaders.

The reference reads per-patient rows from cwd-relative CSVs and NIfTI files (partial_modality_training.py:60-158,
simple_fusion.py:98-154); real data is not redistributable and there is no network, so the entry points and
bench.py build a seeded synthetic cohort with the same tensors, shapes, dtypes and missing-modality conventions:
  image    (1, D, H, W) fp32 in [0,1] (min-max normalised, partial...:96); zeros when imaging is missing
  rnaseq   (rna_dim,)   fp32 z-scored; zeros when missing
  clinical (1,) = age/100; 0 when missing                       (partial...:128)
  label    (2,) = [survival_time, survival_status]; mask (3,) = [has_image, has_rnaseq, has_clinical]
Survival times are distinct (no ties), so every Cox formulation of the reference coincides.
"""
import numpy as np
import torch
import torch.nn.functional as F

COHORT_608 = dict(n=608, imaging=142, rnaseq=427, clinical=587, survival=348, complete=109)   # R/README.md:21-25


def ct_like_volumes(n, dims, rng):
    """Smooth multi-scale fields + noise, min-max normalised per volume to [0,1] like the reference's loader."""
    g = torch.Generator().manual_seed(int(rng.integers(1 << 31)))
    lo = torch.rand(n, 1, 4, 4, 2, generator=g)
    mid = torch.rand(n, 1, 16, 16, 8, generator=g)
    x = (F.interpolate(lo, size=dims, mode="trilinear", align_corners=False) * 0.6
         + F.interpolate(mid, size=dims, mode="trilinear", align_corners=False) * 0.3
         + torch.rand(n, 1, *dims, generator=g) * 0.1)
    x = x * torch.linspace(0.4, 1.0, n).view(n, 1, 1, 1, 1)
    mn = x.amin(dim=(1, 2, 3, 4), keepdim=True)
    mx = x.amax(dim=(1, 2, 3, 4), keepdim=True)
    return ((x - mn) / (mx - mn + 1e-8)).contiguous()


def make_cohort(n=109, dims=(64, 64, 32), rna_dim=5005, seed=608, complete=True, counts=None, signal=True):
    """-> dict of CPU tensors for n patients.  complete=True: every modality + label present (configs 1/2);
    otherwise modality/label availability follows `counts` (config 3, COHORT_608 proportions)."""
    rng = np.random.default_rng(seed)
    rna = rng.normal(0, 1, (n, rna_dim)).astype(np.float32)
    age = np.clip(rng.normal(60, 11, n), 30, 90).astype(np.float32)
    ns = min(16, rna_dim)
    w = rng.normal(0, 0.5, ns).astype(np.float32)
    risk = rna[:, :ns] @ w if signal else np.zeros(n, np.float32)
    time = (rng.exponential(1000.0, n) * np.exp(-risk) + 1.0 + np.arange(n) * 1e-3).astype(np.float32)
    assert len(np.unique(time)) == n
    event = (rng.random(n) < 0.57).astype(np.float32)
    has = np.ones((n, 4), bool)   # image, rnaseq, clinical, survival
    if not complete:
        c = counts or COHORT_608
        for j, key in enumerate(("imaging", "rnaseq", "clinical", "survival")):
            has[:, j] = False
            has[rng.permutation(n)[:int(round(c[key] * n / c["n"]))], j] = True
    img = ct_like_volumes(n, dims, rng)
    img[torch.tensor(~has[:, 0])] = 0.0
    rna[~has[:, 1]] = 0.0
    clin = (age / 100.0) * has[:, 2]
    time = np.where(has[:, 3], time, 0.0).astype(np.float32)
    event = np.where(has[:, 3], event, 0.0).astype(np.float32)
    return dict(image=img, rnaseq=torch.tensor(rna), clinical=torch.tensor(clin.astype(np.float32)).view(n, 1),
                label=torch.tensor(np.stack([time, event], 1)), mask=torch.tensor(has[:, :3].astype(np.float32)),
                has_survival=torch.tensor(has[:, 3]), n=n, dims=tuple(dims))


def cohort_to(cohort, device):
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in cohort.items()}


def cohort_pin(cohort):
    """The cohort in PINNED host memory (the reference's situation: data on the host, `.to(device)` per batch,
    final_multimodal.py:228-247).  A lazy BatchLoader over it names its batches by index exactly as over a device-resident cohort;
    the consumer's gather launch then reads the patients' rows over PCIe -- the batch's host-to-device copy is that one launch."""
    return {k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in cohort.items()}


def gather_view(cohort, with_valid):
    """The dict the fused batch gather (FoldGroupEngine.train_step_indexed) reads: the cohort itself plus, when the style masks
    unlabeled patients out of the loss, a float `valid` column = has_survival.  One object per (cohort, with_valid), so that all
    folds' loaders share it and a group's batches are assembled by ONE gather launch."""
    key = "_gather_view_%d" % int(with_valid)
    if key not in cohort:
        v = {k: t for k, t in cohort.items() if not k.startswith("_gather_view_") and k != "valid"}
        if with_valid:
            v["valid"] = cohort["has_survival"].to(torch.float32)
            if cohort["label"].is_pinned():
                v["valid"] = v["valid"].pin_memory()
        cohort[key] = v
    return cohort[key]


class BatchLoader:
    """Minimal DataLoader stand-in over a tensor-resident cohort: yields the reference's batch dicts.
    shuffle uses its own seeded generator; drop_last=False like the reference's DataLoader calls.
    lazy=True (device-resident cohorts, lock-step training): a batch is only NAMED -- dict(index=[B] patient indices,
    has_survival=[...], gather=<gather_view>) -- and assembled on the GPU by the consumer's single gather launch instead of
    ~8 indexing kernels + copies per fold and batch here."""

    def __init__(self, cohort, indices, batch_size, shuffle=False, seed=0, style="final", lazy=False, with_valid=True):
        self.c, self.idx, self.bs, self.shuffle, self.style = cohort, torch.as_tensor(indices), batch_size, shuffle, style
        self.gen = torch.Generator().manual_seed(seed)
        self.lazy = lazy
        if lazy:
            self.view = gather_view(cohort, with_valid)
            self.hs_cpu = cohort["has_survival"].cpu().tolist()

    def __len__(self):
        return (len(self.idx) + self.bs - 1) // self.bs

    def __iter__(self):
        idx = self.idx[torch.randperm(len(self.idx), generator=self.gen)] if self.shuffle else self.idx
        dev = self.c["image"].device
        for i in range(0, len(idx), self.bs):
            if self.lazy:
                jj = idx[i:i + self.bs]
                yield dict(index=jj, has_survival=[self.hs_cpu[int(k)] for k in jj], gather=self.view)
                continue
            j = idx[i:i + self.bs].to(dev)
            b = dict(image=self.c["image"][j], rnaseq=self.c["rnaseq"][j], clinical=self.c["clinical"][j],
                     label=self.c["label"][j], mask=self.c["mask"][j], has_survival=self.c["has_survival"][j].tolist())
            if self.style == "simple":   # simple_fusion.py:152-153
                b["time"] = self.c["label"][j, 0:1]
                b["event"] = self.c["label"][j, 1:2].long()
            yield b


def kfold_indices(n, n_splits, seed=42):
    """sklearn KFold(n_splits, shuffle=True, random_state=seed).split(range(n)) (final_multimodal.py:316)."""
    from sklearn.model_selection import KFold
    return list(KFold(n_splits=n_splits, shuffle=True, random_state=seed).split(np.arange(n)))
