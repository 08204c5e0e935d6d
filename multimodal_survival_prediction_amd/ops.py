"""Thin per-op wrappers over the C ABI taking torch tensors (used by tests and by the autograd-compatible
host path).  Pointers are taken with .data_ptr(); all launches go to torch's current stream."""
import ctypes

import torch

from . import _lib


def _S():
    return _lib.structs()


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else t.data_ptr()


def dims3(d):
    return _S()["Dims3"](int(d[0]), int(d[1]), int(d[2]))


def bnsrc(gamma, beta, count, train, sum=None, sumsq=None, rmean=None, rvar=None, eps=1e-5):
    return _S()["BnSrc"](ptr(sum), ptr(sumsq), ptr(rmean), ptr(rvar), ptr(gamma), ptr(beta),
                         1.0 / float(count), float(eps), 1 if train else 0)


def call(name, p):
    lib = _lib.load_library()
    _lib.check(getattr(lib, name)(ctypes.byref(p), stream()), name)


def init_coords(B, dims, device):
    D, H, W = dims
    out = torch.empty(B * D * H * W, dtype=torch.int32, device=device)
    _lib.check(_lib.load_library().mms_init_coords(out.data_ptr(), B, D, H, W, stream()), "mms_init_coords")
    return out


def conv1_fwd(x, K, w, y, bn, M, osum=None, osumsq=None, pool=False, in_dims=(0, 0, 0)):
    """x: [Min, ldx] slab (first K columns); w: [N, K]; y: [M, ldy] view (column offset applied by slicing)."""
    p = _S()["Conv1FwdP"](ptr(x), x.stride(0), M, K, ptr(w), w.shape[0], ptr(y), y.stride(0), bn,
                          ptr(osum), ptr(osumsq), 1 if pool else 0, dims3(in_dims))
    call("mms_conv1_fwd", p)


def conv3_fwd(y1, coords, dims, wp, out, bn, osum=None, osumsq=None):
    M = y1.shape[0]
    p = _S()["Conv3FwdP"](ptr(y1), ptr(coords), dims3(dims), M, ptr(wp), ptr(out), out.stride(0), bn,
                          ptr(osum), ptr(osumsq))
    call("mms_conv3_fwd", p)


def conv0_fwd(x, in_dims, out_dims, coords, w, y, osum=None, osumsq=None):
    p = _S()["Conv0FwdP"](ptr(x), dims3(in_dims), dims3(out_dims), ptr(coords), y.shape[0], ptr(w), ptr(y),
                          ptr(osum), ptr(osumsq))
    call("mms_conv0_fwd", p)


def pool_fwd(y0, in_dims, out_dims, B, slab, argmax, bn, osum=None, osumsq=None):
    p = _S()["PoolFwdP"](ptr(y0), dims3(in_dims), dims3(out_dims), B, ptr(slab), slab.stride(0), ptr(argmax), bn,
                         ptr(osum), ptr(osumsq))
    call("mms_pool_fwd", p)


def head_fwd(slab, C, B, V, bn, w, bias, pooled, out):
    p = _S()["HeadFwdP"](ptr(slab), slab.stride(0), C, B, V, bn, ptr(w), ptr(bias), w.shape[0], ptr(pooled), ptr(out))
    call("mms_head_fwd", p)


def pack_conv3(w):
    wpf = torch.empty(32 * 27 * 128, dtype=torch.float32, device=w.device)
    wpb = torch.empty(128 * 27 * 32, dtype=torch.float32, device=w.device)
    _lib.check(_lib.load_library().mms_pack_conv3(w.data_ptr(), wpf.data_ptr(), wpb.data_ptr(), stream()), "mms_pack_conv3")
    return wpf, wpb
