"""GPU preprocessing and the on-disk cohort contract against the reference's own library
calls (scipy.ndimage.zoom order 1, sklearn StandardScaler)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV, assert_close


@pytest.mark.parametrize("native,target", [((40, 50, 23), (64, 64, 32)), ((128, 96, 70), (64, 64, 32)), ((64, 64, 32), (64, 64, 32)),
                                            ((5, 7, 1), (8, 8, 4))])
def test_ct_preprocess_matches_scipy_zoom(native, target):
    from oracle import preproc as OP
    from multimodal_survival_prediction_amd import cohort_io
    rng = np.random.default_rng(sum(native))
    vol = (rng.random(native, dtype=np.float32) * 2500 - 900).astype(np.float32)
    want = OP.ct_preprocess(vol.astype(np.float64), target)
    assert want.shape == target
    got = cohort_io.ct_preprocess(torch.tensor(vol, device=DEV), target)
    torch.cuda.synchronize()
    assert_close(got, torch.tensor(want), 1e-5, "ct preprocess")


def test_rna_log_zscore_matches_sklearn():
    from oracle import preproc as OP
    from multimodal_survival_prediction_amd import cohort_io
    rng = np.random.default_rng(1)
    counts = rng.poisson(rng.gamma(2.0, 50.0, size=(1, 300)), size=(97, 300)).astype(np.float32)
    counts[:, 5] = 7.0                                   # zero-variance gene: StandardScaler leaves scale 1 -> all zeros
    want = OP.rna_log_zscore(counts)
    got = cohort_io.rna_log_zscore(torch.tensor(counts, device=DEV))
    torch.cuda.synchronize()
    assert_close(got, torch.tensor(want), 1e-5, "rna log2 + zscore")
    assert float(got[:, 5].abs().max()) == 0.0


def test_cohort_roundtrip_through_disk(tmp_path):
    """write_cohort -> load_cohort reproduces the in-memory cohort: identity when stored at the network grid, scipy-zoom of
    the stored volume otherwise; missing modalities stay zeros with mask 0."""
    from oracle import preproc as OP
    from multimodal_survival_prediction_amd import cohort_io, data
    dims = (32, 32, 16)
    c = data.make_cohort(n=14, dims=dims, rna_dim=40, seed=9, complete=False, counts=dict(n=14, imaging=9, rnaseq=10, clinical=12, survival=8))
    cohort_io.write_cohort(str(tmp_path / "a"), c)
    got = cohort_io.load_cohort(str(tmp_path / "a"), DEV, target_size=dims)
    torch.cuda.synchronize()
    assert got["n"] == 14 and got["rnaseq"].shape == (14, 40)
    assert torch.equal(got["mask"].cpu(), c["mask"]) and torch.equal(got["has_survival"].cpu(), c["has_survival"])
    assert_close(got["rnaseq"], c["rnaseq"], 1e-6, "rnaseq"); assert_close(got["clinical"], c["clinical"], 1e-6, "clinical")
    assert_close(got["label"], c["label"], 1e-6, "label")
    assert_close(got["image"], c["image"], 2e-5, "image (stored at the network grid: min-max undoes the HU-like scaling)")
    # stored at another resolution: the loader must equal scipy.zoom of what is on disk
    cohort_io.write_cohort(str(tmp_path / "b"), c, native_dims=(20, 45, 24), seed=3)
    got = cohort_io.load_cohort(str(tmp_path / "b"), DEV, target_size=dims)
    mt, _ = cohort_io.read_tables(str(tmp_path / "b"))
    for i, p in enumerate(mt["nifti_path"]):
        if isinstance(p, str):
            want = OP.ct_preprocess(np.load(p).astype(np.float64), dims)
            assert_close(got["image"][i, 0], torch.tensor(want), 1e-5, "image %d" % i)
        else:
            assert float(got["image"][i].abs().max()) == 0.0
