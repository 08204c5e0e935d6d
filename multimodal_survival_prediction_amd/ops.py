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


def dn_opts(base=None, **kw):
    """MmsDnOpts (include/mmsurv.h): launch-shape options of the DenseNet121 drivers and convolution entry points.  All-zero = the
    defaults; the library reads no environment variable, so this block is the only way to select a kernel form (A/B measurements, the
    parity tests that must reach one form).  base: another block or a dict to start from."""
    S = _S()["MmsDnOpts"]
    names = {f[0] for f in S._fields_}
    o = S()
    if isinstance(base, dict):
        kw = dict(base, **kw)
    elif base is not None:
        ctypes.memmove(ctypes.byref(o), ctypes.byref(base), ctypes.sizeof(S))
    for k, v in kw.items():
        if k not in names:
            raise AttributeError("MmsDnOpts has no field %r (include/mmsurv.h)" % k)
        setattr(o, k, int(v))
    return o


def opts_ref(o):
    """ctypes argument for a `const MmsDnOpts*` parameter: NULL (defaults) for None."""
    return None if o is None else ctypes.byref(o)


def call(name, p, opts=None):
    """Single-model entry point name(p, stream); with opts: its fold-group form name_group(p, 1, opts, stream) (the entry points that
    have several kernel forms take the launch-shape options there)."""
    lib = _lib.load_library()
    if opts is None:
        _lib.check(getattr(lib, name)(ctypes.byref(p), stream()), name)
    else:
        _lib.check(getattr(lib, name + "_group")(ctypes.byref(p), 1, ctypes.byref(opts), stream()), name + "_group")


def init_coords(B, dims, device):
    D, H, W = dims
    out = torch.empty(B * D * H * W, dtype=torch.int32, device=device)
    _lib.check(_lib.load_library().mms_init_coords(out.data_ptr(), B, D, H, W, stream()), "mms_init_coords")
    return out


def conv1_fwd(x, K, w, y, bn, M, osum=None, osumsq=None, pool=False, in_dims=(0, 0, 0), partial=None, ksplit=0,
              counters=None, opts=None):
    """x: [Min, ldx] slab (first K columns); w: [N, K]; y: [M, ldy] view (column offset applied by slicing).
    partial/ksplit/counters: split the K loop over workgroups with a last-arriver fixup (Conv1FwdP in mmsurv.h)."""
    p = _S()["Conv1FwdP"](ptr(x), x.stride(0), M, K, ptr(w), w.shape[0], ptr(y), y.stride(0), bn,
                          ptr(osum), ptr(osumsq), 1 if pool else 0, dims3(in_dims), 0, 0, ptr(partial), ksplit, ptr(counters))
    call("mms_conv1_fwd", p, opts)


def conv3_fwd(y1, coords, dims, wp, out, bn, osum=None, osumsq=None, partial=None, nsplit=27, wfrag=False, opts=None):
    """wfrag: wp is the fragment-ordered pack (pack_conv3_frag) -- small grids only (Conv3FwdP.wfrag in mmsurv.h)."""
    M = y1.shape[0]
    p = _S()["Conv3FwdP"](ptr(y1), ptr(coords), dims3(dims), M, ptr(wp), ptr(out), out.stride(0), bn,
                          ptr(osum), ptr(osumsq), ptr(partial), nsplit)
    p.wfrag = 1 if wfrag else 0
    call("mms_conv3_fwd", p, opts)


def conv0_fwd(x, in_dims, out_dims, coords, w, y, osum=None, osumsq=None):
    p = _S()["Conv0FwdP"](ptr(x), dims3(in_dims), dims3(out_dims), ptr(coords), y.shape[0], ptr(w), ptr(y),
                          ptr(osum), ptr(osumsq))
    call("mms_conv0_fwd", p)


def pool_fwd(y0, in_dims, out_dims, B, slab, argmax, bn, osum=None, osumsq=None):
    p = _S()["PoolFwdP"](ptr(y0), dims3(in_dims), dims3(out_dims), B, ptr(slab), slab.stride(0), ptr(argmax), bn,
                         ptr(osum), ptr(osumsq))
    call("mms_pool_fwd", p)


def head_fwd(slab, C, B, V, bn, w, bias, pooled, out):
    p = _S()["HeadFwdP"](ptr(slab), slab.stride(0), C, B, V, bn, ptr(w), ptr(bias), w.shape[0], ptr(pooled), ptr(out), out.stride(0))
    call("mms_head_fwd", p)


def pack_conv3(w):
    wpf = torch.empty(32 * 27 * 128, dtype=torch.float32, device=w.device)
    wpb = torch.empty(128 * 27 * 32, dtype=torch.float32, device=w.device)
    _lib.check(_lib.load_library().mms_pack_conv3(w.data_ptr(), wpf.data_ptr(), wpb.data_ptr(), stream()), "mms_pack_conv3")
    return wpf, wpb


def pack_conv3_frag(w):
    """canonical conv2 weight -> the two MFMA-fragment-ordered packs the small-grid kernels read with contiguous 1-KB loads."""
    wff = torch.empty(32 * 27 * 128, dtype=torch.float32, device=w.device)
    wfb = torch.empty(128 * 27 * 32, dtype=torch.float32, device=w.device)
    _lib.check(_lib.load_library().mms_pack_conv3_frag(w.data_ptr(), wff.data_ptr(), wfb.data_ptr(), stream()), "mms_pack_conv3_frag")
    return wff, wfb


# ---- backward ops ---------------------------------------------------------------------------------------
def bnbwd(s1, s2):
    return _S()["BnBwd"](ptr(s1), ptr(s2))


def conv3_bwd_data(dz, coords, dims, wpb, y1, bn, dbn, s1, s2, partial=None, nsplit=27, wfrag=False, opts=None):
    p = _S()["Conv3BwdDataP"](ptr(dz), dz.stride(0), ptr(coords), dims3(dims), y1.shape[0], ptr(wpb), ptr(y1), bn,
                              ptr(dbn), ptr(s1), ptr(s2), ptr(partial), nsplit)
    p.wfrag = 1 if wfrag else 0
    call("mms_conv3_bwd_data", p, opts)


def conv3_bwd_weight(y1, coords, dims, bn, dz, dw, msplit=1, tapmajor=False, opts=None, layout=None):
    p = _S()["Conv3BwdWP"](ptr(y1), ptr(coords), dims3(dims), y1.shape[0], bn, ptr(dz), dz.stride(0), ptr(dw), msplit,
                           layout if layout is not None else (1 if tapmajor else 0))          # Conv3BwdWP.dw_layout
    call("mms_conv3_bwd_weight", p, opts)


def conv1_bwd(which, dyraw, M, N, x, K, bn_in, w, dw, dbn, s1, s2, y=None, bn_out=None, bb_out=None, pool=False,
              in_dims=(0, 0, 0), msplit=1, dgamma_out=None, dbeta_out=None, fuse_dx=None, fuse_accumulate=True, fuse_dgamma=None,
              fuse_dbeta=None, opts=None):
    S = _S()
    p = S["Conv1BwdP"]()
    p.dyraw, p.lddy = ptr(dyraw), dyraw.stride(0)
    p.y, p.ldy = (ptr(y), y.stride(0)) if y is not None else (None, 0)
    p.has_bn_out = 1 if bn_out is not None else 0
    p.bn_out = bn_out if bn_out is not None else bn_in
    p.bb_out = bb_out if bb_out is not None else S["BnBwd"](None, None)
    p.M, p.N = M, N
    p.x, p.ldx, p.K, p.bn_in = ptr(x), x.stride(0), K, bn_in
    setattr(p, "in", dims3(in_dims))
    p.w = ptr(w)
    p.pool = 1 if pool else 0
    p.dw = ptr(dw)
    p.dbn, p.lddbn = ptr(dbn), dbn.stride(0)
    p.s1, p.s2 = ptr(s1), ptr(s2)
    p.msplit = msplit
    p.dgamma_out, p.dbeta_out = ptr(dgamma_out), ptr(dbeta_out)
    if fuse_dx is not None:          # norm1 backward in the data kernel's epilogue (Conv1BwdP.fuse_dx, M <= 128)
        p.fuse_dx, p.fuse_lddx, p.fuse_accumulate = ptr(fuse_dx), fuse_dx.stride(0), 1 if fuse_accumulate else 0
        p.fuse_dgamma, p.fuse_dbeta = ptr(fuse_dgamma), ptr(fuse_dbeta)
    if which == "weight":
        call("mms_conv1_bwd_weight", p)
    else:
        call("mms_conv1_bwd_data", p, opts)


def bn_bwd_apply(dbn, x, dx, M, C, bn, bb, accumulate, dgamma, dbeta):
    p = _S()["BnBwdApplyP"](ptr(dbn), dbn.stride(0), ptr(x), x.stride(0), ptr(dx), dx.stride(0), M, C, bn, bb,
                            1 if accumulate else 0, ptr(dgamma), ptr(dbeta))
    call("mms_bn_bwd_apply", p)


def head_bwd(dout, pooled, slab, C, B, V, bn, w, dw, dbias, dgamma, dbeta, dslab):
    p = _S()["HeadBwdP"](ptr(dout), dout.stride(0), ptr(pooled), ptr(slab), slab.stride(0), C, B, V, bn, ptr(w), w.shape[0],
                         ptr(dw), ptr(dbias), ptr(dgamma), ptr(dbeta), ptr(dslab), dslab.stride(0))
    call("mms_head_bwd", p)


def pool_bwd(dslab, argmax, out_dims, in_dims, B, y0, bn, dbn, s1, s2, coords=None):
    p = _S()["PoolBwdP"](ptr(dslab), dslab.stride(0), ptr(argmax), dims3(out_dims), dims3(in_dims), B, ptr(y0), bn,
                         ptr(dbn), ptr(s1), ptr(s2), ptr(coords))
    call("mms_pool_bwd", p)


def pool_act(x, K, bn, in_dims, y):
    """Transition pre-pass: y[m'][:K] = AvgPool3d(2, 2)(relu(bn(x)))[m'] for channels-last x [B*in][ldx] (mms_pool_act)."""
    p = _S()["PoolActP"](ptr(x), x.stride(0), K, bn, dims3(in_dims), y.shape[0], ptr(y), y.stride(0))
    _lib.check(_lib.load_library().mms_pool_act_group(ctypes.byref(p), 1, stream()), "mms_pool_act_group")


def conv0_bwd_weight(dbn, y0, bn, bb, x, in_dims, out_dims, coords, dw, msplit, dgamma, dbeta, dw_rep=None):
    """dw_rep: optional zeroed [nrep <= 8][64 * 343] replica scratch (Conv0BwdWP.dw_rep: spreads the gradient atomics, a second launch adds
    the replicas into dw and leaves them zeroed)"""
    p = _S()["Conv0BwdWP"](ptr(dbn), ptr(y0), bn, bb, ptr(x), dims3(in_dims), dims3(out_dims), ptr(coords),
                           y0.shape[0], ptr(dw), msplit, ptr(dgamma), ptr(dbeta), ptr(dw_rep), 0 if dw_rep is None else dw_rep.shape[0])
    call("mms_conv0_bwd_weight", p)


# ---- heads ------------------------------------------------------------------------------------------------
def inprolog(bn=None, train=False, drop_p=0.0, drop_mask=None, rng=None, stream_id=0):
    """bn: None or a torch.nn.BatchNorm1d-like holder with weight/bias/running_mean/running_var/num_batches_tracked."""
    S = _S()["InProlog"]
    if bn is None:
        return S(0, None, None, None, None, None, 1e-5, 0.1, 1 if train else 0, float(drop_p), ptr(drop_mask), ptr(rng), stream_id)
    return S(1, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked),
             float(bn.eps), float(bn.momentum), 1 if train else 0, float(drop_p), ptr(drop_mask), ptr(rng), stream_id)


def linear_fwd(x, K, pro, w, bias, y, out_relu):
    p = _S()["LinearFwdP"](ptr(x), x.stride(0), x.shape[0], K, pro, ptr(w), ptr(bias), w.shape[0], ptr(y), y.stride(0),
                           1 if out_relu else 0)
    call("mms_linear_fwd", p)


def linear_bwd(dy, y, out_relu, x, K, pro, w, dw, dbias, dx=None, dgamma=None, dbeta=None):
    p = _S()["LinearBwdP"](ptr(dy), dy.stride(0), ptr(y), y.stride(0), 1 if out_relu else 0, ptr(x), x.stride(0),
                           x.shape[0], K, pro, ptr(w), w.shape[0], ptr(dw), ptr(dbias), ptr(dx),
                           dx.stride(0) if dx is not None else 0, ptr(dgamma), ptr(dbeta))
    call("mms_linear_bwd", p)


def gate_params(feats, mask, w1, b1, w2, b2, hidden, gate, fused, dfused=None, ent_weight=0.0, dfeats=None,
                dw1=None, db1=None, dw2=None, db2=None, entropy=None, dgate_ext=None):
    return _S()["GateP"](ptr(feats), ptr(mask), feats.shape[0], ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(hidden), ptr(gate),
                         ptr(fused), ptr(dfused), float(ent_weight), ptr(dgate_ext), ptr(dfeats), ptr(dw1), ptr(db1),
                         ptr(dw2), ptr(db2), ptr(entropy))


TIE_MODES = {"breslow": 0, "efron": 1}


def cox_fwd_bwd(h, time, event, valid=None, scale=1.0, want_grad=True, ties="breslow"):
    """h: [n] or [n,1] fp32 on device -> (out[2] = {loss, usable}, dh or None).  ties: CoxP.tie_mode (include/mmsurv.h)."""
    n = h.shape[0]
    lse = torch.empty(n, device=h.device)
    out = torch.empty(2, device=h.device)
    dh = torch.empty(n, device=h.device) if want_grad else None
    frac = torch.empty(n, device=h.device) if TIE_MODES[ties] == 1 else None
    p = _S()["CoxP"](ptr(h), h.stride(0), ptr(time), ptr(event), ptr(valid), n, float(scale), ptr(lse), ptr(dh), 1, ptr(out),
                     TIE_MODES[ties], ptr(frac))
    call("mms_cox_fwd_bwd", p)
    return out, dh


def cindex_counts(h, time, event):
    counts = torch.zeros(3, dtype=torch.int64, device=h.device)
    p = _S()["CindexP"](ptr(h), ptr(time), ptr(event), h.shape[0], ptr(counts))
    call("mms_cindex_counts", p)
    return counts


def adam_params(p_, g, m, v, hyper, sumsq, step, skip_flag=None, adamw=False, acc=None, cox_out=None, entropy=None,
                rng=None, w2=None):
    """w2: None or dict(off=int64[n] device tensor, pack_b=ptr table (int64[n] device tensor of addresses), pack_f=the same,
    fragmask=int): the conv2 tensors kept in packed primary storage inside the flat buffers (AdamP.w2_*, include/mmsurv.h)."""
    a = _S()["AdamP"](ptr(p_), ptr(g), ptr(m), ptr(v), p_.numel(), ptr(hyper), ptr(sumsq), ptr(step), ptr(skip_flag),
                      1 if adamw else 0, ptr(acc), ptr(cox_out), ptr(entropy), ptr(rng))
    if w2 is not None:
        a.w2_off, a.w2_pack_b, a.w2_pack_f = ptr(w2["off"]), ptr(w2["pack_b"]), ptr(w2["pack_f"])
        a.n_w2, a.w2_fragmask = int(w2["off"].numel()), int(w2["fragmask"])
    return a


_WORKER_STREAMS = {}


def worker_streams(device, n):
    """The process-wide HIP streams that concurrent fold (sub-)groups step on: created once, in a fixed order, and shared by every
    consumer (training epoch, validation pass, the benchmark's legs).  HIP multiplexes streams onto a few hardware queues in the order
    they were first created/used, so side streams created later in a process (after graph-capture warm-up streams, per-engine streams ...)
    can end up sharing a queue: measured on the 2 x 10-model leg of bench.py -- 3550 vs 2740 patients/s depending only on how many
    streams the legs before it had created.  One early, fixed set keeps the mapping the same for the whole run."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    lst = _WORKER_STREAMS.setdefault(key, [])
    if n > len(lst):
        import os, warnings
        cap = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)        # ROCm's cap on hardware queues per device; the null stream takes one
        if n + 1 > cap:
            warnings.warn(f"{n} worker streams + the null stream on GPU_MAX_HW_QUEUES={cap} hardware queues: two streams will share a queue and "
                          f"serialise (K = 5 epoch, four sub-groups: 1879 patients/s against 2696 with GPU_MAX_HW_QUEUES=5; "
                          f"profiles/r03_cu_partition_experiments.txt).  Set GPU_MAX_HW_QUEUES >= {n + 1} before the process starts, "
                          f"or use <= {cap - 1} streams.", RuntimeWarning, stacklevel=2)
    while len(lst) < n:
        lst.append(torch.cuda.Stream(device=device))
    return lst[:n]


def cluster_workgroups(B, dims):
    """Workgroups per model of the persistent per-block ("cluster") launches of dense blocks 3 and 4 (csrc/dn_cl.hip; the rule of
    csrc/dn_net.hip make_plan): clusters of 8 workgroups, each owning whole samples with <= 16 or <= 32 rows; 0 = the block has too many
    voxels per sample (or too many clusters) and runs per layer."""
    out = []
    for shift in (4, 5):
        vox = max(1, (dims[0] >> shift) * (dims[1] >> shift) * (dims[2] >> shift))
        rt = 1 if vox <= 16 else (2 if vox <= 32 else 0)
        if not rt:
            out.append(0)
            continue
        spc = min(B, (16 * rt) // vox)
        ncl = -(-B // spc)
        out.append(8 * ncl if ncl <= 8 else 0)
    return tuple(out)


def persistent_opts(base, device, ng, B, dims):
    """MmsDnOpts for a driver call of ng lock-step models: `base` with persist_b3 / persist_b4 switched to the per-layer path (-1) where
    the persistent launches of all worker streams could not be co-resident -- their workgroups hand data to each other inside the
    launch, one such launch per worker stream may be in flight, and the chip has a fixed number of CUs (the kernels' bounded sweeps +
    time-out word stay as the backstop).  The rule shared by SurvivalEngine, FoldGroupEngine and bench.py; decided at launch (=
    graph-capture) time and passed to the drivers as an argument."""
    if dims is None or len(dims) != 3:
        return base
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    w3, w4 = cluster_workgroups(B, dims)
    kw = {}
    if base.persist_b3 > 0 and w3 * ng * max_worker_streams() > cus:
        kw["persist_b3"] = -1
    if base.persist_b4 >= 0 and w4 * ng * max_worker_streams() > cus:
        kw["persist_b4"] = -1
    return dn_opts(base, **kw) if kw else base


def persistent_b4_fits(device, ng, B=4, dims=(64, 64, 32)):
    """Whether dense block 4 runs as one persistent launch per pass for ng lock-step models per worker stream (bench.py's mirror of the drivers)."""
    return persistent_opts(dn_opts(), device, ng, B, dims).persist_b4 >= 0


def max_worker_streams():
    """How many worker streams this process has created (over all devices: an upper bound on concurrently stepping sub-groups)."""
    return max([len(v) for v in _WORKER_STREAMS.values()] or [1])
