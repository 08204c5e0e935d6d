"""Helpers shared by the GPU parity tests."""
import numpy as np
import torch

DEV = "cuda:0"


def rel_err(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def assert_close(got, ref, tol=1e-4, what=""):
    """max-abs error relative to the reference's max magnitude (fp32 tolerance from BASELINE.json north_star: 1e-4)."""
    e = rel_err(got, ref)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol}"


def cl(x):
    """NCDHW -> channels-last matrix [B*D*H*W, C]."""
    B, C = x.shape[:2]
    return x.permute(0, 2, 3, 4, 1).reshape(-1, C).contiguous()


def uncl(m, B, dims):
    D, H, W = dims
    return m.reshape(B, D, H, W, -1).permute(0, 4, 1, 2, 3).contiguous()


def stats(device, C):
    return torch.zeros(C, dtype=torch.float64, device=device), torch.zeros(C, dtype=torch.float64, device=device)


# Launch-shape options (MmsDnOpts, include/mmsurv.h) under which a model's arithmetic does not depend on how many models share its
# launches: no group-size dependent tap split / tile shapes / kernel forms / row chunks.  Group-vs-single comparisons that must be
# tight pass these to BOTH sides; the default options are compared in tests/test_gpu_fold_group.py::test_group_default_options.
GROUP_INDEPENDENT_OPTS = dict(split_wgs=1000000, conv1_ksplit=-1, conv1_small=-1, conv3_mt=-1, big_ng=-1, ms3_rows=512, conv3w_mt=-1,
                              ms1_div=1)
