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
