"""GPU parity of the DenseNet forward ops against plain torch fp32 CPU ops (through the C ABI)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import DEV, assert_close, cl, stats, uncl


@pytest.fixture(scope="module")
def ops():
    from multimodal_survival_prediction_amd import ops as o
    return o


def _bn_ref(x, g, b, train, rm=None, rv=None):
    return F.batch_norm(x, rm, rv, g, b, training=train, momentum=0.0, eps=1e-5)


@pytest.mark.parametrize("B,dims,K,N,ld", [(4, (16, 16, 8), 64, 128, 256), (4, (16, 16, 8), 224, 128, 256),
                                           (4, (8, 8, 4), 480, 128, 512), (4, (2, 2, 1), 992, 128, 1024),
                                           (1, (4, 4, 2), 256, 128, 1024)])
@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("small", [-1, 1])      # MmsDnOpts.conv1_small: tile-GEMM forms / the 16 x 16 whole-K kernel of dn_c1s.hip (round 3)
def test_conv1_fwd(ops, B, dims, K, N, ld, train, small):
    o = ops.dn_opts(conv1_small=small)
    torch.manual_seed(0)
    M = B * dims[0] * dims[1] * dims[2]
    x = torch.randn(B, K, *dims) * 1.5 + 0.3
    g, b = torch.rand(K) + 0.5, torch.randn(K) * 0.1
    rm, rv = torch.randn(K) * 0.2, torch.rand(K) + 0.5
    w = torch.randn(N, K) / K ** 0.5
    ref = F.conv3d(F.relu(_bn_ref(x, g, b, train, rm, rv)), w.view(N, K, 1, 1, 1))
    slab = torch.zeros(M, ld, device=DEV)
    slab[:, :K] = cl(x).to(DEV)
    s, q = stats(DEV, K)
    s += slab[:, :K].double().sum(0); q += (slab[:, :K].double() ** 2).sum(0)
    bn = ops.bnsrc(g.to(DEV), b.to(DEV), M, train, s, q, rm.to(DEV), rv.to(DEV))
    y = torch.zeros(M, N, device=DEV)
    os_, oq = stats(DEV, N)
    keep = (g.to(DEV), b.to(DEV), rm.to(DEV), rv.to(DEV))
    bn = ops.bnsrc(keep[0], keep[1], M, train, s, q, keep[2], keep[3])
    ops.conv1_fwd(slab, K, w.to(DEV), y, bn, M, os_ if train else None, oq if train else None, opts=o)
    torch.cuda.synchronize()
    assert_close(y, cl(ref), 1e-4, "conv1 y")
    if train:
        assert_close(os_, cl(ref).double().sum(0), 1e-4, "conv1 sum")
        assert_close(oq, (cl(ref).double() ** 2).sum(0), 1e-4, "conv1 sumsq")


@pytest.mark.parametrize("M,K,ksplit", [(128, 640, 5), (16, 992, 8), (100, 288, 3), (128, 256, 2)])
def test_conv1_fwd_ksplit(ops, M, K, ksplit):
    """K loop split over workgroups + last-arriver fixup (no second launch): same y and statistics as the unsplit kernel;
    repeated launches reuse the self-re-arming ticket counters.  (MmsDnOpts.conv1_small = -1: the tile-GEMM forms under test; the same
    shapes -- ragged rows, K not a multiple of 64 -- go through the small-launch kernel at the end.)"""
    gemm, small = ops.dn_opts(conv1_small=-1), ops.dn_opts(conv1_small=1)
    torch.manual_seed(3)
    N, ld = 128, 1024
    slab = torch.randn(M, ld, device=DEV) * 1.5 + 0.3
    g, b = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    s, q = slab[:, :K].double().sum(0).contiguous(), (slab[:, :K].double() ** 2).sum(0).contiguous()
    bn = ops.bnsrc(g, b, M, True, s, q)
    y0 = torch.zeros(M, N, device=DEV); s0, q0 = stats(DEV, N)
    ops.conv1_fwd(slab, K, w, y0, bn, M, s0, q0, opts=gemm)
    partial = torch.full((ksplit * M * N,), float("nan"), device=DEV)
    counters = torch.zeros(64, dtype=torch.int32, device=DEV)
    for rep in range(3):
        y1 = torch.zeros(M, N, device=DEV); s1, q1 = stats(DEV, N)
        ops.conv1_fwd(slab, K, w, y1, bn, M, s1, q1, partial=partial, ksplit=ksplit, counters=counters, opts=gemm)
        torch.cuda.synchronize()
        assert int(counters.abs().sum()) == 0                    # re-armed
        assert_close(y1, y0, 2e-6, "ksplit y")
        assert_close(s1, s0, 1e-6, "ksplit sum"); assert_close(q1, q0, 1e-6, "ksplit sumsq")
    a = torch.relu((slab[:, :K].double() - (s / M)) / torch.sqrt(q / M - (s / M) ** 2 + 1e-5) * g.double() + b.double())
    assert_close(y1, a @ w.double().t(), 1e-4, "ksplit y vs fp64 reference")
    y2 = torch.zeros(M, N, device=DEV); s2, q2 = stats(DEV, N)
    ops.conv1_fwd(slab, K, w, y2, bn, M, s2, q2, opts=small)
    torch.cuda.synchronize()
    assert_close(y2, a @ w.double().t(), 1e-4, "small-launch kernel y vs fp64 reference")
    assert_close(s2, s0, 1e-5, "small-launch sum"); assert_close(q2, q0, 1e-5, "small-launch sumsq")


@pytest.mark.parametrize("B,dims,K", [(4, (16, 16, 8), 256), (2, (8, 8, 4), 512), (4, (4, 4, 2), 1024)])
def test_transition_fwd(ops, B, dims, K):
    torch.manual_seed(1)
    N = K // 2
    M = B * dims[0] * dims[1] * dims[2]
    x = torch.randn(B, K, *dims) + 0.2
    g, b = torch.rand(K) + 0.5, torch.randn(K) * 0.1
    w = torch.randn(N, K) / K ** 0.5
    ref = F.avg_pool3d(F.conv3d(F.relu(_bn_ref(x, g, b, True)), w.view(N, K, 1, 1, 1)), 2, 2)
    slab = cl(x).to(DEV)
    s, q = slab.double().sum(0), (slab.double() ** 2).sum(0)
    gd, bd = g.to(DEV), b.to(DEV)
    bn = ops.bnsrc(gd, bd, M, True, s, q)
    ldn = 2 * K
    nxt = torch.zeros(M // 8, ldn, device=DEV)
    os_, oq = stats(DEV, N)
    ops.conv1_fwd(slab, K, w.to(DEV), nxt, bn, M // 8, os_, oq, pool=True, in_dims=dims)
    torch.cuda.synchronize()
    assert_close(nxt[:, :N], cl(ref), 1e-4, "transition y")
    assert float(nxt[:, N:].abs().max()) == 0.0
    assert_close(os_, cl(ref).double().sum(0), 1e-4, "transition sum")
    # the driver's default form: norm / relu / pool by mms_pool_act into a scratch, then the plain 1x1x1 convolution over the pooled rows
    # with the IDENTITY BatchNorm block (BnSrc.gamma == NULL), tile-GEMM and small-launch kernels
    pooled = torch.full((M // 8, K + 4), float("nan"), device=DEV)
    ops.pool_act(slab, K, bn, dims, pooled)
    torch.cuda.synchronize()
    assert_close(pooled[:, :K], cl(F.avg_pool3d(F.relu(_bn_ref(x, g, b, True)), 2, 2)), 1e-5, "pooled operand")
    assert bool(torch.isnan(pooled[:, K:]).all())
    ident = ops._S()["BnSrc"]()
    for small in (-1, 1):
        nx2 = torch.zeros(M // 8, ldn, device=DEV)
        o2, q2 = stats(DEV, N)
        ops.conv1_fwd(pooled, K, w.to(DEV), nx2, ident, M // 8, o2, q2, opts=ops.dn_opts(conv1_small=small))
        torch.cuda.synchronize()
        assert_close(nx2[:, :N], cl(ref), 1e-4, "transition y (pre-pass, conv1_small=%d)" % small)
        assert_close(o2, os_, 1e-5, "transition sum (pre-pass)")


# (no tap split + W <= 16: MmsDnOpts.conv3_mt = 2 takes the 64-row multi-tap kernel for M >= 1024 rows (cases 1 and 6-8; 7 = ragged last tile),
# conv3_mt = 3 its 32-row form for every case with M >= 64 (ragged tiles, 2x2x1 and 5x3x2 grids included))
_C3_GRIDS = [(4, (16, 16, 8)), (2, (8, 8, 4)), (4, (4, 4, 2)), (4, (2, 2, 1)), (3, (5, 3, 2)), (4, (8, 8, 4)), (3, (7, 7, 8)), (2, (8, 16, 16))]
# small: MmsDnOpts.conv3_small -- None = default (the all-tap 16-row kernels of dn_c3s.hip wherever the rows' neighbourhood window fits: every
# grid here except 16x16x8, 7x7x8, 8x16x16), "0" = off (tile-GEMM form), "1" / "2" = force one / two 16-column output tiles per wave
# "f1" / "f2": the same with the weights in MFMA-fragment order (Conv3FwdP.wfrag -- what the network driver feeds those kernels; small grids only)
_C3_FORMS = [(0, 2, None), (0, 2, "0"), (0, 2, "1"), (0, 2, "2"), (0, 2, "f1"), (0, 2, "f2"), (0, 3, None), (27, 2, None), (3, 2, None), (5, 2, None)]
_small_grid = lambda dims: 16 + 2 * (dims[1] * dims[2] + dims[2] + 1) <= 120


def _c3_opts(ops, small, **kw):
    if small is not None:
        kw["conv3_small"] = {"0": -1, "1": 1, "2": 2}[small[-1]]
    return ops.dn_opts(**kw)


@pytest.mark.parametrize("B,dims,split,mt,small", [g + f for g in _C3_GRIDS for f in _C3_FORMS if not (f[2] and f[2][0] == "f" and not _small_grid(g[1]))])
@pytest.mark.parametrize("train", [True, False])
def test_conv3_fwd(ops, B, dims, train, split, mt, small):
    o = _c3_opts(ops, small, conv3_mt=mt)      # take a multi-tap kernel whenever the shape allows it (default: by tile count)
    frag = small is not None and small[0] == "f"
    torch.manual_seed(2)
    M = B * dims[0] * dims[1] * dims[2]
    y1 = torch.randn(B, 128, *dims) + 0.1
    g, b = torch.rand(128) + 0.5, torch.randn(128) * 0.1
    rm, rv = torch.randn(128) * 0.2, torch.rand(128) + 0.5
    w = torch.randn(32, 128, 3, 3, 3) / (128 * 27) ** 0.5
    ref = F.conv3d(F.relu(_bn_ref(y1, g, b, train, rm, rv)), w, padding=1)
    y1d = cl(y1).to(DEV)
    s, q = y1d.double().sum(0), (y1d.double() ** 2).sum(0)
    keep = (g.to(DEV), b.to(DEV), rm.to(DEV), rv.to(DEV))
    bn = ops.bnsrc(keep[0], keep[1], M, train, s, q, keep[2], keep[3])
    coords = ops.init_coords(B, dims, DEV)
    wd = w.to(DEV)
    wpf, wpb = ops.pack_conv3(wd)
    slab = torch.zeros(M, 256, device=DEV)
    os_, oq = stats(DEV, 32)
    part = torch.empty(27 * M * 32, device=DEV) if split else None      # tap-split path: partial tiles + reduce kernel
    ops.conv3_fwd(y1d, coords, dims, ops.pack_conv3_frag(wd)[0] if frag else wpf, slab[:, 64:96], bn, os_ if train else None, oq if train else None, part,
                  split or 27, wfrag=frag, opts=o)
    torch.cuda.synchronize()
    assert_close(slab[:, 64:96], cl(ref), 1e-4, "conv3 out")
    assert float(slab[:, :64].abs().max()) == 0.0 and float(slab[:, 96:].abs().max()) == 0.0
    if train:
        assert_close(os_, cl(ref).double().sum(0), 1e-4, "conv3 sum")
        assert_close(oq, (cl(ref).double() ** 2).sum(0), 1e-4, "conv3 sumsq")
    # packed layouts
    assert torch.equal(wpf.view(32, 27, 128), wd.view(32, 128, 27).permute(0, 2, 1))
    assert torch.equal(wpb.view(128, 27, 32), wd.view(32, 128, 27).permute(1, 2, 0))


@pytest.mark.parametrize("B,dims", [(4, (64, 64, 32)), (2, (32, 64, 64)), (1, (16, 16, 8)), (2, (10, 6, 8))])
def test_conv0_pool_fwd(ops, B, dims):
    torch.manual_seed(3)
    x = torch.rand(B, 1, *dims)
    w = torch.randn(64, 1, 7, 7, 7) / 343 ** 0.5
    g, b = torch.rand(64) + 0.5, torch.randn(64) * 0.1
    y0 = F.conv3d(x, w, stride=2, padding=3)
    od = tuple(y0.shape[2:])
    a0 = F.relu(_bn_ref(y0, g, b, True))
    p0, idx = F.max_pool3d(a0, 3, 2, 1, return_indices=True)
    pd = tuple(p0.shape[2:])
    M0, M1 = B * od[0] * od[1] * od[2], B * pd[0] * pd[1] * pd[2]
    coords = ops.init_coords(B, od, DEV)
    y0d = torch.empty(M0, 64, device=DEV)
    s, q = stats(DEV, 64)
    ops.conv0_fwd(x.to(DEV).contiguous(), dims, od, coords, w.view(64, 343).to(DEV).contiguous(), y0d, s, q)
    torch.cuda.synchronize()
    assert_close(y0d, cl(y0), 1e-4, "conv0")
    assert_close(s, cl(y0).double().sum(0), 1e-4, "conv0 sum")
    gd, bd = g.to(DEV), b.to(DEV)
    bn = ops.bnsrc(gd, bd, M0, True, s, q)
    slab = torch.zeros(M1, 256, device=DEV)
    am = torch.zeros(M1, 64, dtype=torch.uint8, device=DEV)
    s1, q1 = stats(DEV, 64)
    ops.pool_fwd(y0d, od, pd, B, slab, am, bn, s1, q1)
    torch.cuda.synchronize()
    assert_close(slab[:, :64], cl(p0), 1e-4, "pool")
    assert_close(s1, cl(p0).double().sum(0), 1e-4, "pool sum")
    assert_close(q1, (cl(p0).double() ** 2).sum(0), 1e-4, "pool sumsq")
    # argmax taps decode to torch's flat indices (first max in scan order)
    amc = am.cpu().long().view(B, *pd, 64).permute(0, 4, 1, 2, 3)
    odd, ohh, oww = torch.meshgrid(torch.arange(pd[0]), torch.arange(pd[1]), torch.arange(pd[2]), indexing="ij")
    idd = 2 * odd - 1 + amc // 9
    ihh = 2 * ohh - 1 + (amc // 3) % 3
    iww = 2 * oww - 1 + amc % 3
    flat = (idd * od[1] + ihh) * od[2] + iww
    # ties (e.g. all-zero windows) must resolve to the same element torch picks
    assert float((flat == idx).double().mean()) > 0.9999


@pytest.mark.parametrize("B,V,C", [(4, 4, 1024), (2, 8, 1024), (16, 4, 1024)])
def test_head_fwd(ops, B, V, C):
    torch.manual_seed(4)
    x = torch.randn(B, C, V, 1, 1) + 0.2
    g, b = torch.rand(C) + 0.5, torch.randn(C) * 0.1
    w, bias = torch.randn(128, C) / C ** 0.5, torch.randn(128) * 0.1
    a = F.relu(_bn_ref(x, g, b, True))
    pooled = a.mean(dim=(2, 3, 4))
    ref = pooled @ w.t() + bias
    slab = cl(x).to(DEV)
    s, q = slab.double().sum(0), (slab.double() ** 2).sum(0)
    gd, bd = g.to(DEV), b.to(DEV)
    bn = ops.bnsrc(gd, bd, B * V, True, s, q)   # s, q, gd, bd stay referenced: bnsrc stores raw pointers
    pd_, out = torch.empty(B, C, device=DEV), torch.empty(B, 128, device=DEV)
    ops.head_fwd(slab, C, B, V, bn, w.to(DEV), bias.to(DEV), pd_, out)
    torch.cuda.synchronize()
    assert_close(pd_, pooled, 1e-4, "pooled")
    assert_close(out, ref, 1e-4, "head out")
