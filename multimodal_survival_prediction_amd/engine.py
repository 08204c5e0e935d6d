"""Host-side execution engine for the three survival networks.

One `SurvivalEngine` per model instance.  It
  * re-points every parameter (and creates every gradient) as a view into ONE flat fp32 buffer, so the
    gradient zeroing, the global-norm reduction and the Adam update are single launches over 14-17 M floats;
  * keeps, per batch size, a *plan*: static input buffers, the DenseNet workspace, the head activation buffers
    and the prebuilt C-ABI parameter blocks (ctypes structs holding raw device pointers) for every launch;
  * runs a plan either eagerly (autograd-compatible path used by the nn.Module's forward/backward) or as a
    captured HIP graph of the WHOLE training step (zero-grad -> forward -> Cox -> backward -> clip -> Adam),
    which is what `training.train_epoch_*` and bench.py replay once per batch.

PyTorch supplies device memory, streams and graph capture only; every numeric op is a kernel behind
include/mmsurv.h.  Nothing here falls back to torch math.
"""
import ctypes
import os

import torch
import torch.nn as nn

from . import _lib, ops

_S = _lib.structs



def check_train_batch(B, bn_world=1):
    """A training-mode forward on ONE patient: torch's BatchNorm1d (every head of the reference's models has one) raises
    `ValueError: Expected more than 1 value per channel when training` -- e.g. a DataLoader tail of 1 in the reference's loops.
    Same error here, instead of the kernels' bare MMS_ERR_ARG."""
    if B * max(int(bn_world), 1) <= 1:
        raise ValueError("Expected more than 1 value per channel when training: a batch of 1 patient cannot be normalised by the "
                         "heads' BatchNorm1d layers (torch raises the same in the reference's train_epoch; choose a batch size / "
                         "fold split without a tail of 1)")

class _Lin:
    """One nn.Linear application with the neighbouring BN1d/ReLU/Dropout folded in (see InProlog in mmsurv.h)."""

    def __init__(self, lin, src, dst, out_relu, pro_bn=None, pro_drop=None, need_dx=True):
        self.lin, self.src, self.dst, self.out_relu = lin, src, dst, out_relu
        self.pro_bn, self.pro_drop, self.need_dx = pro_bn, pro_drop, need_dx


def head_program(model):
    """-> dict(kind, width, ct_cols, lins=[...], gate=(l1, l2) | None, bufs={name: width}, encoder).
    Buffer/column wiring restates the reference forward()s: final_multimodal.py:122-150,
    partial_modality_training.py:234-277, simple_fusion.py:217-236."""
    kind = type(model).__name__
    if kind in ("MultiModalSurvivalNet", "PartialModalityNet"):
        r, c, f = model.rna_encoder, model.clinical_encoder, model.fusion
        rna_dim = r[0].in_features
        lins = [
            _Lin(r[0], ("rna", 0), ("r1", 0), False, need_dx=False),
            _Lin(r[4], ("r1", 0), ("feats", 128), True, pro_bn=r[1], pro_drop=r[3]),
            _Lin(c[0], ("clin", 0), ("feats", 256), True, need_dx=False),
            _Lin(f[0], ("fused" if kind == "PartialModalityNet" else "feats", 0), ("f1", 0), False),
            _Lin(f[4], ("f1", 0), ("f2", 0), True, pro_bn=f[1], pro_drop=f[3]),
            _Lin(model.cox_head, ("f2", 0), ("hz", 0), False),
        ]
        bufs = dict(rna=rna_dim, clin=c[0].in_features, r1=512, feats=288, f1=256, f2=128, hz=1)
        gate = None
        if kind == "PartialModalityNet":
            bufs["fused"] = 288
            gate = (model.gate[0], model.gate[2])
        return dict(kind=kind, width=288, ct_cols=0, lins=lins, gate=gate, bufs=bufs, encoder=model.ct_encoder,
                    n_pre=3)
    if kind == "SimpleFusionModel":
        r, f = model.rna_encoder, model.fusion
        lins = [
            _Lin(r[0], ("rna", 0), ("r1", 0), False, need_dx=False),
            _Lin(r[4], ("r1", 0), ("r2", 0), False, pro_bn=r[1], pro_drop=r[3]),
            _Lin(r[8], ("r2", 0), ("feats", 0), True, pro_bn=r[5], pro_drop=r[7]),
            _Lin(f[0], ("feats", 0), ("f1", 0), False),
            _Lin(f[4], ("f1", 0), ("f2", 0), True, pro_bn=f[1], pro_drop=f[3]),
            _Lin(f[7], ("f2", 0), ("hz", 0), False, pro_drop=f[6]),
        ]
        rd = r[8].out_features                      # rna_feature_dim (simple_fusion.py:163); the encoder's img_feature_dim columns follow
        idim = f[0].in_features - rd
        bufs = dict(rna=r[0].in_features, r1=1024, r2=512, feats=rd + idim, f1=256, f2=128, hz=1)
        return dict(kind=kind, width=rd + idim, ct_cols=rd, lins=lins, gate=None, bufs=bufs, encoder=model.image_encoder,
                    n_pre=3, enc_width=idim)
    if kind == "FlexibleMultimodalModel":      # flexible_multimodal.py:157-256: simple-fusion heads, [image | rna] order, missing bias
        r, f = model.rna_encoder, model.fusion
        lins = [
            _Lin(r[0], ("rna", 0), ("r1", 0), False, need_dx=False),
            _Lin(r[4], ("r1", 0), ("r2", 0), False, pro_bn=r[1], pro_drop=r[3]),
            _Lin(r[8], ("r2", 0), ("feats", model.missing_image_bias.numel()), True, pro_bn=r[5], pro_drop=r[7]),
            _Lin(f[0], ("feats", 0), ("f1", 0), False),
            _Lin(f[4], ("f1", 0), ("f2", 0), True, pro_bn=f[1], pro_drop=f[3]),
            _Lin(f[7], ("f2", 0), ("hz", 0), False, pro_drop=f[6]),
        ]
        rd, idim = r[8].out_features, model.missing_image_bias.numel()
        bufs = dict(rna=r[0].in_features, r1=1024, r2=512, feats=idim + rd, f1=256, f2=128, hz=1)
        return dict(kind=kind, width=idim + rd, ct_cols=0, lins=lins, gate=None, bufs=bufs, encoder=model.image_encoder, n_pre=3,
                    mix=dict(biases=[model.missing_image_bias, model.missing_rna_bias], segs=[(0, idim), (idim, rd)]), enc_width=idim)
    if kind == "RNASeqSurvivalModel":          # train_rnaseq_only.py:126-151: [Linear, BN1d, ReLU, Dropout] x n + Linear(., 1)
        mods = list(model.mlp)
        lin_idx = [i for i, m in enumerate(mods) if isinstance(m, nn.Linear)]
        lins, bufs = [], dict(rna=mods[0].in_features)
        for n, i in enumerate(lin_idx):
            src = ("rna", 0) if n == 0 else ("h%d" % n, 0)
            last = n == len(lin_idx) - 1
            dst = ("hz", 0) if last else ("h%d" % (n + 1), 0)
            if not last:
                bufs["h%d" % (n + 1)] = mods[i].out_features
            pro_bn = mods[lin_idx[n - 1] + 1] if n > 0 else None
            pro_drop = mods[lin_idx[n - 1] + 3] if n > 0 else None
            lins.append(_Lin(mods[i], src, dst, False, pro_bn=pro_bn, pro_drop=pro_drop, need_dx=n > 0))
        bufs["hz"] = 1
        return dict(kind=kind, width=0, ct_cols=0, lins=lins, gate=None, bufs=bufs, encoder=None, n_pre=0)
    raise TypeError("unsupported model %s" % kind)


class _Plan:
    """Everything that depends on the batch size B (and the volume dims)."""
    pass


class SurvivalEngine:
    def __init__(self, model, adamw=None, lr=1e-4, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0,
                 gate_entropy_weight=0.01, cox_ties=None, dn_opts=None, _slots=None):
        """dn_opts: launch-shape options of the encoder drivers (dict of MmsDnOpts fields, include/mmsurv.h; None = defaults).
        _slots (used by FoldGroupEngine): dict(gflat=[n] fp32, sumsq=[1] fp64, entropy=[1] fp32) views of group-wide
        buffers, so the per-step zeroing of a whole fold group is three memsets."""
        self.lib = _lib.load_library()
        self.model = model
        self.prog = head_program(model)
        # MmsDnOpts of every encoder driver call of this engine (the width of class_layers.out rides in it)
        from .densenet import DenseNet121
        self.packed = isinstance(self.prog["encoder"], DenseNet121)      # conv2 weights in packed primary storage (MmsDnOpts.w2_packed)
        self.dn_opts = ops.dn_opts(dict(dn_opts or {}), out_features=self.prog.get("enc_width", 128), w2_packed=1 if self.packed else 0)
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise RuntimeError("SurvivalEngine: move the model to the GPU first (model.to('cuda')); no CPU fallback")
        self.device = p0.device
        self.params = list(model.parameters())
        self._slots = _slots or {}
        self._flatten()
        n = self.flat.numel()
        self.m = torch.zeros(n, device=self.device)
        self.v = torch.zeros(n, device=self.device)
        self.adamw = bool(adamw) if adamw is not None else self.prog["kind"] in ("SimpleFusionModel", "FlexibleMultimodalModel",
                                                                                 "RNASeqSurvivalModel")
        self.hyper = torch.tensor([lr, betas[0], betas[1], eps, weight_decay, max_norm], device=self.device)
        self.sumsq = self._slots["sumsq"] if "sumsq" in self._slots else torch.zeros(1, dtype=torch.float64, device=self.device)
        self.entropy = self._slots["entropy"] if "entropy" in self._slots else torch.zeros(1, device=self.device)
        self.step_count = torch.zeros(1, device=self.device)
        self.rng = torch.tensor([0x5EED, 0], dtype=torch.int32, device=self.device)
        self.ent_weight = gate_entropy_weight
        from . import losses
        self.tie_mode = ops.TIE_MODES[cox_ties or losses.default_ties()]    # CoxP.tie_mode of the fused step's loss
        self.side_stream = torch.cuda.Stream(device=self.device)
        self.ev_fork, self.ev_join = torch.cuda.Event(), torch.cuda.Event()
        self.ev_fork.record(); self.ev_join.record()          # materialise the handles
        self.plans = {}
        self.dropout_masks = {}      # parity mode: {lin index: [B, K] multiplicative mask}
        # device-side epoch accumulators: [sum loss*usable, n usable, sum entropy, n batches]
        self.acc = torch.zeros(4, device=self.device)
        self.acc_eval = torch.zeros(4, device=self.device)      # validation: [sum of batch losses, usable batches, -, batches]
        model._mms_engine = self

    # ---- parameters ------------------------------------------------------------------------------
    def _flatten(self):
        n = sum(p.numel() for p in self.params)
        pad = (-n) % 4
        self.flat = torch.zeros(n + pad, device=self.device)
        self.gflat = self._slots["gflat"] if "gflat" in self._slots else torch.zeros(n + pad, device=self.device)
        assert self.gflat.numel() == n + pad and self.gflat.is_contiguous()
        o = 0
        self.gviews = []
        # DenseNet121's 58 conv2 weights (_DenseLayer.layers.conv2, [32, 128, 3, 3, 3]) are stored [cout][tap][cin] -- the layout the
        # forward kernels read and the weight-gradient kernels flush -- as strided VIEWS with the torch shape: state_dict(),
        # load_state_dict(), .grad and torch optimisers see ordinary tensors (MmsDnOpts.w2_packed; csrc/heads.hip w2_adam_pack_kernel)
        w2ids = set()
        if self.packed:
            w2ids = {id(p) for k, p in self.prog["encoder"].named_parameters() if k.endswith("layers.conv2.weight")}
            assert len(w2ids) == 58
        self.w2_offsets = []

        def view(buf, o, p):
            if id(p) in w2ids:
                return buf.as_strided((32, 128, 3, 3, 3), (27 * 128, 1, 9 * 128, 3 * 128, 128), buf.storage_offset() + o)      # (the offset is absolute in the storage)
            return buf[o:o + p.numel()].view_as(p)
        with torch.no_grad():
            for p in self.params:
                if id(p) in w2ids:
                    assert tuple(p.shape) == (32, 128, 3, 3, 3) and o % 4 == 0
                    self.w2_offsets.append(o)
                v = view(self.flat, o, p)
                v.copy_(p.data)
                p.data = v
                self.gviews.append(view(self.gflat, o, p))
                o += p.numel()
        self._key = (self.params[0].data_ptr(), self.params[-1].data_ptr())
        self.w2 = None
        if self.w2_offsets:
            # the derived packs (backward-data pack; forward MFMA-fragment pack of the small-grid layers) and the device tables of AdamP.w2_*
            self.w2_packs = torch.zeros(len(self.w2_offsets), 2, 32 * 27 * 128, device=self.device)
            base, step = self.w2_packs.data_ptr(), 32 * 27 * 128 * 4
            self.w2 = dict(off=torch.tensor(self.w2_offsets, dtype=torch.int64, device=self.device),
                           pack_b=torch.tensor([base + 2 * i * step for i in range(len(self.w2_offsets))], dtype=torch.int64, device=self.device),
                           pack_f=torch.tensor([base + (2 * i + 1) * step for i in range(len(self.w2_offsets))], dtype=torch.int64, device=self.device),
                           fragmask=None)
            self._packs_version = None

    def _check_params(self):
        if (self.params[0].data_ptr(), self.params[-1].data_ptr()) != self._key:
            raise RuntimeError("model parameters were re-allocated (e.g. .to()/.load on another device) after the "
                               "engine was built; create a new SurvivalEngine")

    def sync_packs(self, force=False):
        """Packed primary conv2 storage: the derived packs are written by the fused optimiser step (mms_clip_adam); after ANY other
        change of the weights -- load_state_dict, a torch optimiser, the roll-back around graph capture -- they are rebuilt here
        (mms_w2_pack).  Detected through the version counter torch keeps for the flat buffer (its views share it); cheap to call."""
        if self.w2 is None or self.w2["fragmask"] is None:
            return
        if force or self._packs_version != self.flat._version:
            _lib.check(self.lib.mms_w2_pack(ctypes.byref(self._w2_adam), ops.stream()), "mms_w2_pack")
            self._packs_version = self.flat._version

    def set_lr(self, lr):
        self.hyper[0] = lr

    def get_lr(self):
        return float(self.hyper[0])

    def attach_grads(self):
        """Expose the flat gradient buffer as .grad views (what clip_grad_norm_/torch.optim would read)."""
        for p, g in zip(self.params, self.gviews):
            p.grad = g

    # ---- plans -----------------------------------------------------------------------------------
    def plan(self, B, dims, heads_only=False, bn_world=1):
        """heads_only: no encoder workspace (the replicated global-batch heads of the SyncBN data-parallel step: the encoder's
        128 feature columns are filled by an all-gather); bn_world: ranks the encoder's BatchNorm statistics span."""
        dims = tuple(dims) if dims is not None else ()
        key = (B,) + dims + (("heads",) if heads_only else ()) + ((("bnw", bn_world),) if bn_world > 1 else ())
        if key in self.plans:
            return self.plans[key]
        self._check_params()
        P = _Plan()
        P.B, P.dims = B, dims
        dev = self.device
        prog = self.prog
        P.has_enc = prog["encoder"] is not None and not heads_only
        P.bn_world = bn_world
        D, H, W = dims if P.has_enc else (1, 1, 1)
        P.ct = torch.zeros(B, 1, D, H, W, device=dev)
        P.big = B > 32                 # rows beyond the one-lane-per-column head kernels: MFMA GEMM path (mms_linear_big_*)
        if P.big and (P.has_enc or prog["gate"] is not None or prog.get("mix") is not None):
            raise RuntimeError("batches of more than 32 rows are supported for the encoder-less RNASeqSurvivalModel only "
                               "(the reference trains the imaging models at batch 4-16)")
        # row pitch padded to a multiple of 4 floats so that the 5005-wide RNA rows are 16-B aligned (float4 operand loads)
        P.buf = {k: torch.zeros(B, (w + 3) & ~3, device=dev)[:, :w] if P.big and w > 1 else torch.zeros(B, w, device=dev)
                 for k, w in prog["bufs"].items()}
        P.dbuf = {k: torch.zeros(B, w, device=dev) for k, w in prog["bufs"].items() if k not in ("rna", "clin")}
        P.mask = torch.ones(B, 3, device=dev)
        P.time = torch.zeros(B, device=dev)
        P.event = torch.zeros(B, device=dev)
        P.valid = torch.ones(B, device=dev)
        P.cox_out = torch.zeros(2, device=dev)
        P.lse = torch.zeros(B, device=dev)
        P.tie_frac = torch.zeros(B, device=dev)
        # encoder: DenseNet121-3D (MONAI topology) or the reference's 3-conv fallback
        enc = prog["encoder"]
        gmap = {id(p): g for p, g in zip(self.params, self.gviews)}
        P.fallback = isinstance(enc, nn.Sequential)
        if P.has_enc:
            eparams = list(enc.parameters())
            ebufs = list(enc.buffers())
            npar, nbuf = (12, 9) if P.fallback else (364, 363)
            assert len(eparams) == npar and len(ebufs) == nbuf
            if self.w2 is not None:
                mask = ctypes.c_uint64(0)
                _lib.check(self.lib.mms_dn121_w2_fragmask(B, D, H, W, ctypes.byref(self.dn_opts), ctypes.byref(mask)), "mms_dn121_w2_fragmask")
                if self.w2["fragmask"] is None:
                    self.w2["fragmask"] = mask.value
                    self._w2_adam = ops.adam_params(self.flat, self.gflat, self.m, self.v, self.hyper, self.sumsq, self.step_count, w2=self.w2)
                elif self.w2["fragmask"] != mask.value:
                    raise RuntimeError("this engine's conv2 packs were laid out for another volume size (fragment-order layers differ); "
                                       "use one volume size per model")
            nbytes = ctypes.c_size_t(0)
            wsfn = self.lib.mms_fb_workspace_bytes if P.fallback else self.lib.mms_dn121_workspace_bytes
            _lib.check(wsfn(B, D, H, W, ctypes.byref(nbytes)), "workspace_bytes")
            P.ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
            ptrs = [p.data_ptr() for p in eparams]
            if self.w2 is not None:       # packed primary conv2 storage: + 116 pointers (layer l: backward-data pack, forward fragment pack)
                base, step = self.w2_packs.data_ptr(), 32 * 27 * 128 * 4
                ptrs += [base + j * step for j in range(2 * len(self.w2_offsets))]
            P.ptab = (ctypes.c_void_p * len(ptrs))(*ptrs)
            P.btab = (ctypes.c_void_p * nbuf)(*[b.data_ptr() for b in ebufs])
            P.gtab = (ctypes.c_void_p * npar)(*[gmap[id(p)].data_ptr() for p in eparams])
            if P.fallback:
                if bn_world > 1:
                    raise RuntimeError("SyncBN drives the DenseNet121-3D encoder only")
                _lib.check(self.lib.mms_fb_init(P.ws.data_ptr(), B, D, H, W, P.btab, ops.stream()), "mms_fb_init")
            else:
                _lib.check(self.lib.mms_dn121_init_sync(P.ws.data_ptr(), B, D, H, W, P.ptab, P.btab, bn_world, ctypes.byref(self.dn_opts),
                                                        ops.stream()), "mms_dn121_init_sync")
        # head launches (train / eval variants)
        P.lin_fwd = {True: [], False: []}
        P.lin_bwd = []
        if P.big:
            self._plan_big(P, gmap)
        for i, L in enumerate(() if P.big else prog["lins"]):
            xs, xo = L.src
            ys, yo = L.dst
            x = P.buf[xs][:, xo:]
            y = P.buf[ys][:, yo:]
            K, N = L.lin.in_features, L.lin.out_features
            for train in (True, False):
                pro = self._prolog(L, i, train)
                P.lin_fwd[train].append(_S()["LinearFwdP"](x.data_ptr(), x.stride(0), B, K, pro, L.lin.weight.data_ptr(),
                                                            L.lin.bias.data_ptr(), N, y.data_ptr(), y.stride(0),
                                                            1 if L.out_relu else 0))
            dy = P.dbuf[ys][:, yo:]
            dx = P.dbuf[xs][:, xo:] if L.need_dx else None
            pro = self._prolog(L, i, True)
            P.lin_bwd.append(_S()["LinearBwdP"](
                dy.data_ptr(), dy.stride(0), y.data_ptr(), y.stride(0), 1 if L.out_relu else 0,
                x.data_ptr(), x.stride(0), B, K, pro, L.lin.weight.data_ptr(), N,
                gmap[id(L.lin.weight)].data_ptr(), gmap[id(L.lin.bias)].data_ptr(),
                dx.data_ptr() if dx is not None else None, dx.stride(0) if dx is not None else 0,
                gmap[id(L.pro_bn.weight)].data_ptr() if L.pro_bn is not None else None,
                gmap[id(L.pro_bn.bias)].data_ptr() if L.pro_bn is not None else None))
        P.gate = None
        if prog["gate"] is not None:
            g1, g2 = prog["gate"]
            P.hidden = torch.zeros(B, 64, device=dev)
            P.gatew = torch.zeros(B, 3, device=dev)
            P.gate = ops.gate_params(P.buf["feats"], P.mask, g1.weight, g1.bias, g2.weight, g2.bias, P.hidden, P.gatew,
                                     P.buf["fused"], P.dbuf["fused"], self.ent_weight, P.dbuf["feats"],
                                     gmap[id(g1.weight)], gmap[id(g1.bias)], gmap[id(g2.weight)], gmap[id(g2.bias)],
                                     self.entropy)
        P.mix = None
        if prog.get("mix") is not None:            # learnable missing-modality bias between the encoders and the fusion MLP
            mx = prog["mix"]
            P.mask2 = torch.ones(B, len(mx["segs"]), device=dev)
            M = _S()["MixP"]()
            fe, dfe = P.buf["feats"], P.dbuf["feats"]
            M.feats, M.ld, M.M = fe.data_ptr(), fe.stride(0), B
            M.mask, M.ldm, M.nseg = P.mask2.data_ptr(), P.mask2.stride(0), len(mx["segs"])
            M.dfeats, M.ldd = dfe.data_ptr(), dfe.stride(0)
            for i, ((b0, w), bias) in enumerate(zip(mx["segs"], mx["biases"])):
                M.seg_begin[i], M.seg_width[i] = b0, w
                M.bias[i], M.dbias[i] = bias.data_ptr(), gmap[id(bias)].data_ptr()
            P.mix = M
        hz = P.buf["hz"]
        P.cox = _S()["CoxP"](hz.data_ptr(), 1, P.time.data_ptr(), P.event.data_ptr(), P.valid.data_ptr(), B, 1.0,
                             P.lse.data_ptr(), P.dbuf["hz"].data_ptr(), 1, P.cox_out.data_ptr(), self.tie_mode, P.tie_frac.data_ptr())
        # validation: the batch's Cox value only (no gradient), (loss | usable) kept per batch for validate_*'s bookkeeping
        P.cox_eval_out = torch.zeros(2, device=self.device)
        P.cox_eval = _S()["CoxP"](hz.data_ptr(), 1, P.time.data_ptr(), P.event.data_ptr(), P.valid.data_ptr(), B, 1.0,
                                  P.lse.data_ptr(), None, 1, P.cox_eval_out.data_ptr(), self.tie_mode, P.tie_frac.data_ptr())
        book = dict(acc=self.acc, cox_out=P.cox_out, entropy=self.entropy, rng=self.rng)   # per-step bookkeeping, in-kernel
        w2 = self.w2 if (self.w2 is not None and self.w2["fragmask"] is not None) else None
        if self.w2 is not None and w2 is None:        # (an encoder-less plan of an imaging model, e.g. the SyncBN heads plan, made first)
            raise RuntimeError("create the engine's first plan with the encoder (packed conv2 storage needs the volume size)")
        P.adam = ops.adam_params(self.flat, self.gflat, self.m, self.v, self.hyper, self.sumsq, self.step_count,
                                 None, self.adamw, w2=w2, **book)
        P.adam_skip = ops.adam_params(self.flat, self.gflat, self.m, self.v, self.hyper, self.sumsq, self.step_count,
                                      P.cox_out[1:], self.adamw, w2=w2, **book)
        P.graphs = {}
        self.plans[key] = P
        return P

    def _plan_big(self, P, gmap):
        """LinBigP blocks (include/mmsurv.h) of the large-batch Linear chain.  BatchNorm1d statistics of a buffer live in one
        fp64 array per plan: [sum | sumsq | s1 | s2] x columns, zeroed once per training step."""
        prog, B, dev = self.prog, P.B, self.device
        S, ptr = _S()["LinBigP"], ops.ptr
        widths = {L.src[0]: L.lin.in_features for L in prog["lins"] if L.pro_bn is not None}
        off, o = {}, 0
        for name, k in widths.items():
            off[name] = o
            o += 4 * k
        P.big_stats = torch.zeros(max(o, 1), dtype=torch.float64, device=dev)
        P.big_dbn = torch.zeros(B, max(list(widths.values()) + [4]), device=dev)
        st = lambda name, j: P.big_stats[off[name] + j * widths[name]:]
        P.big_lin = {True: [], False: []}
        for i, L in enumerate(prog["lins"]):
            (xs, xo), (ys, yo) = L.src, L.dst
            if xo or yo:
                raise RuntimeError("large-batch path: column-offset buffers are not supported")
            x, y = P.buf[xs], P.buf[ys]
            K, N = L.lin.in_features, L.lin.out_features
            for train in (True, False):
                q = S()
                q.x, q.ldx, q.M, q.K = x.data_ptr(), x.stride(0), B, K
                q.w, q.bias, q.N = L.lin.weight.data_ptr(), L.lin.bias.data_ptr(), N
                q.y, q.ldy, q.out_relu = y.data_ptr(), y.stride(0), 1 if L.out_relu else 0
                q.train = 1 if train else 0
                q.drop_p = float(L.pro_drop.p) if L.pro_drop is not None else 0.0
                q.drop_mask, q.rng, q.stream_id = ptr(self.dropout_masks.get(i)), self.rng.data_ptr(), i + 1
                bn = L.pro_bn
                if bn is not None:
                    q.has_bn = 1
                    q.bn.sum, q.bn.sumsq = st(xs, 0).data_ptr(), st(xs, 1).data_ptr()
                    q.bn.rmean, q.bn.rvar = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                    q.bn.gamma, q.bn.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
                    q.bn.inv_count, q.bn.eps, q.bn.train, q.bn.nrep, q.bn.rep_stride = 1.0 / B, float(bn.eps), q.train, 1, 0
                    q.rmean, q.rvar, q.nbt = bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr()
                    q.momentum = float(bn.momentum)
                if train and ys in widths:           # a BatchNorm1d follows: the forward accumulates its batch statistics
                    q.osum, q.osumsq = st(ys, 0).data_ptr(), st(ys, 1).data_ptr()
                if train:
                    dy = P.dbuf[ys]
                    q.dy, q.lddy = dy.data_ptr(), dy.stride(0)
                    q.dw, q.dbias = gmap[id(L.lin.weight)].data_ptr(), gmap[id(L.lin.bias)].data_ptr()
                    tiles = ((N + 63) // 64) * ((K + 63) // 64)
                    q.msplit = max(1, min((512 + tiles - 1) // tiles, (B + 63) // 64))
                    if L.need_dx:
                        dx = P.dbuf[xs]
                        if bn is not None:
                            q.dbn, q.lddbn = P.big_dbn.data_ptr(), P.big_dbn.stride(0)
                            q.s1, q.s2 = st(xs, 2).data_ptr(), st(xs, 3).data_ptr()
                            q.dx, q.lddx = dx.data_ptr(), dx.stride(0)
                            q.dgamma, q.dbeta = gmap[id(bn.weight)].data_ptr(), gmap[id(bn.bias)].data_ptr()
                        else:
                            q.dbn, q.lddbn = dx.data_ptr(), dx.stride(0)
                P.big_lin[train].append(q)

    def _prolog(self, L, idx, train):
        p = float(L.pro_drop.p) if (L.pro_drop is not None) else 0.0
        mask = self.dropout_masks.get(idx)
        return ops.inprolog(L.pro_bn, train=train, drop_p=p, drop_mask=mask, rng=self.rng, stream_id=idx + 1)

    # ---- launches (all on torch's current stream; capturable) ---------------------------------------
    def _opts_arg(self, P):
        """`const MmsDnOpts*` of this engine's driver calls on plan P (ops.persistent_opts: the persistent per-block launches are taken
        only where their workgroups can all be co-resident; decided here, at launch = graph-capture time, and passed as an ARGUMENT)."""
        o = ops.persistent_opts(self.dn_opts, self.device, 1, P.B, P.dims)
        self._opts_live = o          # (keeps the block alive for the duration of the call)
        return ctypes.byref(o)

    def _forward(self, P, train):
        self.sync_packs()
        st = ops.stream()
        lib, prog = self.lib, self.prog
        B, (D, H, W) = P.B, (P.dims if P.has_enc else (1, 1, 1))
        if P.has_enc:
            feats = P.buf["feats"]
            out = feats[:, prog["ct_cols"]:]
            if P.fallback:
                _lib.check(lib.mms_fb_forward(P.ws.data_ptr(), B, D, H, W, P.ct.data_ptr(), P.ptab, P.btab, out.data_ptr(),
                                              feats.stride(0), 1 if train else 0, st), "mms_fb_forward")
            else:
                _lib.check(lib.mms_dn121_forward(P.ws.data_ptr(), B, D, H, W, P.ct.data_ptr(), P.ptab, P.btab, out.data_ptr(),
                                                 feats.stride(0), 1 if train else 0, self._opts_arg(P), st), "mms_dn121_forward")
        if P.big:
            if train:
                P.big_stats.zero_()
            for q in P.big_lin[train]:
                _lib.check(lib.mms_linear_big_fwd(ctypes.byref(q), st), "mms_linear_big_fwd")
            return
        lf = P.lin_fwd[train]
        n_pre = prog["n_pre"]
        for i in range(n_pre):
            _lib.check(lib.mms_linear_fwd(ctypes.byref(lf[i]), st), "mms_linear_fwd")
        if P.gate is not None:
            self.entropy.zero_()
            _lib.check(lib.mms_gate_fwd(ctypes.byref(P.gate), st), "mms_gate_fwd")
        if P.mix is not None:
            _lib.check(lib.mms_missing_mix_fwd(ctypes.byref(P.mix), st), "mms_missing_mix_fwd")
        for i in range(n_pre, len(lf)):
            _lib.check(lib.mms_linear_fwd(ctypes.byref(lf[i]), st), "mms_linear_fwd")

    def _backward_from_dhz(self, P):
        """dbuf['hz'] holds dL/dhazard; accumulates every parameter gradient into gflat."""
        self._backward_heads(P)
        self._backward_encoder(P)

    def _backward_heads(self, P):
        """Heads' backward: dbuf['hz'] -> every head parameter gradient and dbuf['feats'] (gradient wrt the encoder's output)."""
        st = ops.stream()
        lib, prog = self.lib, self.prog
        B, (D, H, W) = P.B, (P.dims if P.has_enc else (1, 1, 1))
        n_pre = prog["n_pre"]
        if P.big:
            for L, q in zip(reversed(prog["lins"]), reversed(P.big_lin[True])):
                _lib.check(lib.mms_linear_big_bwd_w(ctypes.byref(q), st), "mms_linear_big_bwd_w")
                if L.need_dx:
                    _lib.check(lib.mms_linear_big_bwd_x(ctypes.byref(q), st), "mms_linear_big_bwd_x")
                    if L.pro_bn is not None:
                        _lib.check(lib.mms_bn1d_bwd_apply(ctypes.byref(q), st), "mms_bn1d_bwd_apply")
            return
        for i in range(len(P.lin_bwd) - 1, n_pre - 1, -1):
            _lib.check(lib.mms_linear_bwd(ctypes.byref(P.lin_bwd[i]), st), "mms_linear_bwd")
        if P.gate is not None:
            _lib.check(lib.mms_gate_bwd(ctypes.byref(P.gate), st), "mms_gate_bwd")
        if P.mix is not None:
            _lib.check(lib.mms_missing_mix_bwd(ctypes.byref(P.mix), st), "mms_missing_mix_bwd")
        for i in range(n_pre - 1, -1, -1):
            _lib.check(lib.mms_linear_bwd(ctypes.byref(P.lin_bwd[i]), st), "mms_linear_bwd")

    def _backward_encoder(self, P, stage=None, hook=None):
        """Encoder backward from dbuf['feats'].  stage = (block_hi, block_lo): one stage of the DenseNet121 backward
        (mms_dn121_backward_stage; the data-parallel step all-reduces each stage's gradient bucket while the next stage runs);
        hook: SyncBN statistics all-reduce (P.bn_world ranks)."""
        if not P.has_enc:
            return
        st = ops.stream()
        lib, prog = self.lib, self.prog
        B, (D, H, W) = P.B, P.dims
        dfe = P.dbuf["feats"]
        dct = dfe[:, prog["ct_cols"]:]
        if stage is not None or hook is not None or P.bn_world > 1:
            hi, lo = stage if stage is not None else (3, 0)
            _lib.check(lib.mms_dn121_backward_stage(P.ws.data_ptr(), B, D, H, W, P.ct.data_ptr(), P.ptab, dct.data_ptr(), dfe.stride(0),
                                                    P.gtab, hi, lo, P.bn_world, hook, None, self._opts_arg(P), st), "mms_dn121_backward_stage")
            return
        # The weight-gradient fork (mms_dn121_backward_mt) is off by default: measured, it neither helps a single chain
        # (graph branches run mostly serially) nor concurrent fold models (it takes hardware queues away from them).
        if os.environ.get("MMS_SIDE_STREAM") != "1" and not P.fallback:
            _lib.check(lib.mms_dn121_backward(P.ws.data_ptr(), B, D, H, W, P.ct.data_ptr(), P.ptab, dct.data_ptr(),
                                              dfe.stride(0), P.gtab, self._opts_arg(P), st), "mms_dn121_backward")
            return
        if P.fallback:
            _lib.check(lib.mms_fb_backward(P.ws.data_ptr(), B, D, H, W, P.ct.data_ptr(), P.ptab, dct.data_ptr(),
                                           dfe.stride(0), P.gtab, st), "mms_fb_backward")
            return
        _lib.check(lib.mms_dn121_backward_mt(P.ws.data_ptr(), B, D, H, W, P.ct.data_ptr(), P.ptab, dct.data_ptr(),
                                             dfe.stride(0), P.gtab, self._opts_arg(P), st, ctypes.c_void_p(self.side_stream.cuda_stream),
                                             ctypes.c_void_p(self.ev_fork.cuda_event), ctypes.c_void_p(self.ev_join.cuda_event)),
                   "mms_dn121_backward_mt")

    def _global_cox_buffers(self, P, world):
        if getattr(P, "gcox", None) is None or P.gcox["world"] != world:
            n, dev = world * P.B, self.device
            G = dict(world=world, h=torch.zeros(n, device=dev), time=torch.zeros(n, device=dev), event=torch.zeros(n, device=dev),
                     valid=torch.ones(n, device=dev), lse=torch.zeros(n, device=dev), dh=torch.zeros(n, device=dev), rank=0,
                     frac=torch.zeros(n, device=dev))
            G["cox"] = _S()["CoxP"](G["h"].data_ptr(), 1, G["time"].data_ptr(), G["event"].data_ptr(), G["valid"].data_ptr(), n, 1.0,
                                    G["lse"].data_ptr(), G["dh"].data_ptr(), 1, P.cox_out.data_ptr(), self.tie_mode, G["frac"].data_ptr())
            P.gcox = G
        return P.gcox

    def _train_body(self, P, skip_if_unusable, part="all"):
        """zero-grad -> forward -> Cox -> backward [-> gradient all-reduce outside] -> clip -> Adam, epoch accumulators.
        part: "all" (single GPU / fold sharding), or "grad" / "update" = the two halves around the DDP all-reduce."""
        st = ops.stream()
        lib = self.lib
        if part == "fwd":                      # DDP with the global risk set: first third
            self.gflat.zero_()
            self.sumsq.zero_()
            self._forward(P, True)
            return
        if part in ("coxbwd", "coxheads"):     # ... loss over the gathered world*B hazards, own slice of dL/dh [, heads only]
            G = P.gcox
            _lib.check(lib.mms_cox_fwd_bwd(ctypes.byref(G["cox"]), st), "mms_cox_fwd_bwd")
            P.dbuf["hz"][:, 0].copy_(G["dh"][G["rank"] * P.B:(G["rank"] + 1) * P.B])
            if part == "coxheads":
                self._backward_heads(P)
            else:
                self._backward_from_dhz(P)
            return
        if part == "gradheads":                # rank-local risk set: zero-grad, forward, Cox, heads' backward
            self.gflat.zero_()
            self.sumsq.zero_()
            self._forward(P, True)
            _lib.check(lib.mms_cox_fwd_bwd(ctypes.byref(P.cox), st), "mms_cox_fwd_bwd")
            self._backward_heads(P)
            return
        if isinstance(part, tuple) and part[0] == "enc":      # one stage of the encoder backward (None = all of it)
            self._backward_encoder(P, stage=None if part[1] is None else (part[1], part[1]))
            return
        if isinstance(part, tuple) and part[0] == "update":   # gradients arrived as a SUM over ranks: scale (mean for rank-local losses)
            if part[1] != 1.0:
                self.gflat.mul_(part[1])
            part = "update"
        if part in ("all", "grad"):
            self.gflat.zero_()
            self.sumsq.zero_()
            self._forward(P, True)
            _lib.check(lib.mms_cox_fwd_bwd(ctypes.byref(P.cox), st), "mms_cox_fwd_bwd")
            self._backward_from_dhz(P)
            if part == "grad":
                return
        if part == "update":
            self.sumsq.zero_()                 # mms_grad_sumsq ADDS into it; the other parts that zero it are not in this graph
        ad = P.adam_skip if skip_if_unusable else P.adam
        _lib.check(lib.mms_grad_sumsq(ctypes.byref(ad), st), "mms_grad_sumsq")
        _lib.check(lib.mms_clip_adam(ctypes.byref(ad), st), "mms_clip_adam")
        # (dropout counter and the epoch accumulators self.acc are advanced inside mms_grad_sumsq: AdamP.acc / .rng)

    # ---- public: fused training step ------------------------------------------------------------------
    def load_batch(self, P, ct=None, rna=None, clinical=None, mask=None, time=None, event=None, valid=None):
        if P.has_enc:
            P.ct.copy_(ct.reshape(P.ct.shape), non_blocking=True)
        P.buf["rna"].copy_(rna, non_blocking=True)
        if clinical is not None and "clin" in P.buf:
            P.buf["clin"].copy_(clinical.reshape(P.buf["clin"].shape), non_blocking=True)
        if mask is not None:
            (P.mask2 if P.mix is not None else P.mask).copy_(mask, non_blocking=True)
        if time is not None:
            P.time.copy_(time.reshape(-1), non_blocking=True)
            P.event.copy_(event.reshape(-1).to(torch.float32), non_blocking=True)
        if valid is None:
            P.valid.fill_(1.0)
        else:
            P.valid.copy_(valid.reshape(-1).to(torch.float32), non_blocking=True)

    def gather_block(self, P, cohort, idx_dev):
        """GatherP (include/mmsurv.h) that assembles this plan's batch from a cohort dict (data.make_cohort) that lives in HBM
        (data.cohort_to) or in pinned host memory (data.cohort_pin: the gather launch then reads the rows over PCIe): image, rnaseq,
        clinical, mask, label[time, event] and, when present, a float per-patient `valid` column.  idx_dev: [B] int64 device tensor
        that the caller refills before each launch.  With a modality mask in the cohort, image / rnaseq rows of patients without that
        modality are zero-filled instead of read -- after checking once that they are all-zero in the cohort, as the reference's
        dataset makes them (partial_modality_training.py:96-141)."""
        G = _S()["GatherP"]()
        G.idx, G.B = idx_dev.data_ptr(), P.B
        flags = {}
        if "mask" in cohort:
            chk = cohort.setdefault("_absent_rows_zero", {})
            for j, key in enumerate(("image", "rnaseq")):
                if key not in chk:
                    gone = cohort["mask"][:, j] == 0
                    chk[key] = bool((cohort[key][gone] == 0).all()) if bool(gone.any()) else True
                if chk[key]:
                    flags[key] = cohort["mask"][:, j:]
        srcs = [(cohort["rnaseq"], P.buf["rna"], None, flags.get("rnaseq"))]
        if P.has_enc:
            srcs.append((cohort["image"].view(cohort["image"].shape[0], -1), P.ct.view(P.B, -1), None, flags.get("image")))
        if "clin" in P.buf:
            srcs.append((cohort["clinical"], P.buf["clin"], None))
        if P.gate is not None:
            srcs.append((cohort["mask"], P.mask, None))
        if P.mix is not None:            # [has_image, has_rnaseq] = the first columns of the cohort's modality mask
            srcs.append((cohort["mask"], P.mask2, P.mask2.shape[1]))
        lab = cohort["label"]
        srcs.append((lab, P.time.view(P.B, 1), 1))
        srcs.append((lab[:, 1:], P.event.view(P.B, 1), 1))
        if "valid" in cohort:
            srcs.append((cohort["valid"].view(-1, 1), P.valid.view(P.B, 1), 1))
        G.nsrc = len(srcs)
        for i, src in enumerate(srcs):
            a, b, w, flag = src if len(src) == 4 else src + (None,)
            if a.dtype != torch.float32 or not (a.is_cuda or a.is_pinned()):
                raise TypeError("gather sources must be fp32 tensors in device or pinned host memory")
            G.src[i], G.dst[i] = a.data_ptr(), b.data_ptr()
            G.src_ld[i], G.dst_ld[i] = a.stride(0), b.stride(0)
            G.width[i] = w if w is not None else a.shape[1]
            if flag is not None:
                G.present[i], G.present_ld[i] = flag.data_ptr(), flag.stride(0)
        return G

    def train_step(self, ct=None, rna=None, clinical=None, mask=None, time=None, event=None, valid=None, skip_if_unusable=True,
                   use_graph=True, ddp_world=1, global_cox=False, sync_bn=False):
        """One optimisation step on one batch (inputs may live on host or device).  Returns nothing: losses are
        accumulated on the device (`epoch_stats()`), exactly one host sync per epoch instead of one per batch.
        ddp_world > 1: data-parallel step on this rank's shard of the global batch (`_ddp_step`: bucketed gradient all-reduce
        overlapped with the backward; rank-local BatchNorm; rank-local or global_cox risk sets), or with sync_bn=True the exact
        global-batch step (`_ddp_step_syncbn`: SyncBN + replicated heads + global risk set; eager launches)."""
        B = rna.shape[0]
        check_train_batch(B, ddp_world if sync_bn else 1)
        P = self.plan(B, tuple(ct.shape[-3:]) if ct is not None else None, bn_world=ddp_world if (sync_bn and ddp_world > 1) else 1)
        self.load_batch(P, ct, rna, clinical, mask, time, event, valid)
        self.sync_packs()
        if ddp_world > 1 and sync_bn:
            self._ddp_step_syncbn(P, ddp_world)
            return
        if ddp_world > 1:
            self._ddp_step(P, ddp_world, global_cox, use_graph)
            return
        if not use_graph:
            self._train_body(P, skip_if_unusable)
            return
        key = ("train", skip_if_unusable)
        if key not in P.graphs:
            # warm-up on a side stream (first-launch attribute calls, lazy module loads), then capture
            state = [self.flat.clone(), self.m.clone(), self.v.clone(), self.step_count.clone(), self.rng.clone(),
                     self.acc.clone(), [b.clone() for b in self.model.buffers()]]
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._train_body(P, skip_if_unusable)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            with torch.no_grad():   # undo the warm-up step
                self.flat.copy_(state[0]); self.m.copy_(state[1]); self.v.copy_(state[2])
                self.step_count.copy_(state[3]); self.rng.copy_(state[4]); self.acc.copy_(state[5])
                for b, b0 in zip(self.model.buffers(), state[6]):
                    b.copy_(b0)
            self.sync_packs()          # (the roll-back changed the weights: rebuild the derived conv2 packs, outside any capture)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._train_body(P, skip_if_unusable)
            P.graphs[key] = g
            with torch.no_grad():   # capture does not execute, but keep state exact regardless
                self.flat.copy_(state[0]); self.m.copy_(state[1]); self.v.copy_(state[2])
                self.step_count.copy_(state[3]); self.rng.copy_(state[4]); self.acc.copy_(state[5])
                for b, b0 in zip(self.model.buffers(), state[6]):
                    b.copy_(b0)
            self.sync_packs()          # (the roll-back changed the weights: rebuild the derived conv2 packs, outside any capture)
        P.graphs[key].replay()

    # ---- data-parallel step (one process per GPU) --------------------------------------------------------
    def _buckets(self, P):
        """Gradient buckets in the order the backward finalises them (distributed.gradient_buckets) -> (staged, buckets)."""
        if getattr(self, "_bucket_cache", None) is None:
            from . import distributed as D
            enc = self.prog["encoder"]
            staged = enc is not None and not isinstance(enc, nn.Sequential)
            self._bucket_cache = (staged, D.gradient_buckets(self.params, list(enc.parameters()) if enc is not None else [], staged))
        return self._bucket_cache

    def _ddp_sequence(self, P, global_cox):
        """The step as a list of ("part", name) compute pieces (each one HIP graph) and communication points."""
        staged, buckets = self._buckets(P)
        seq = [("part", "fwd"), ("gather",), ("part", "coxheads")] if global_cox else [("part", "gradheads")]
        if staged:
            seq.append(("bucket", 0))
            for k, b in enumerate((3, 2, 1, 0)):
                seq += [("part", ("enc", b)), ("bucket", k + 1)]
        else:
            seq += [("part", ("enc", None)), ("bucket", 0)]
        return seq + [("wait",)], buckets

    def _ddp_step(self, P, world, global_cox, use_graph):
        """Rank-local BatchNorm statistics.  The flat gradient buffer is all-reduced (SUM) bucket by bucket, each bucket launched as
        soon as the backward stage that finalises it has been enqueued, so the collective of stage k runs beside the kernels of
        stage k+1 (torch.distributed enqueues an async collective behind the current stream's work and runs it on its own
        stream; the update waits for all of them).  Rank-local risk sets: gradients are averaged (1/world folded into the update
        graph).  global_cox: risk sets over the world*B patients (all-gather of hazard/time/event/valid), gradients summed."""
        from . import distributed as D
        import torch.distributed as dist
        seq, buckets = self._ddp_sequence(P, global_cox)
        upd = ("update", 1.0 if global_cox else 1.0 / world)
        if global_cox:
            G = self._global_cox_buffers(P, world)
            G["rank"] = dist.get_rank() if dist.is_initialized() else 0

        def run(run_part):
            works = []
            for item in seq:
                if item[0] == "part":
                    run_part(item[1])
                elif item[0] == "gather":
                    D.all_gather_into(G["h"], P.buf["hz"][:, 0], world)
                    D.all_gather_into(G["time"], P.time, world); D.all_gather_into(G["event"], P.event, world)
                    D.all_gather_into(G["valid"], P.valid, world)
                elif item[0] == "bucket":
                    works.append(D.allreduce_ranges_async(self.gflat, buckets[item[1]], world))
                else:
                    for w in works:
                        w()
            run_part(upd)
        if not use_graph:
            run(lambda part: self._train_body(P, False, part))
            return
        key0 = ("ddp", global_cox, world)
        if key0 not in P.graphs:
            # the parts depend on each other through the workspace (statistic accumulators are zeroed by the first part only), so
            # they are warmed up as ONE eager step, rolled back, and then captured without being executed
            state = [self.flat.clone(), self.m.clone(), self.v.clone(), self.step_count.clone(), self.rng.clone(),
                     self.acc.clone(), [b.clone() for b in self.model.buffers()]]
            run(lambda part: self._train_body(P, False, part))
            torch.cuda.synchronize()
            with torch.no_grad():
                self.flat.copy_(state[0]); self.m.copy_(state[1]); self.v.copy_(state[2])
                self.step_count.copy_(state[3]); self.rng.copy_(state[4]); self.acc.copy_(state[5])
                for b, b0 in zip(self.model.buffers(), state[6]):
                    b.copy_(b0)
            self.sync_packs()          # (the roll-back changed the weights: rebuild the derived conv2 packs, outside any capture)
            graphs = {}
            for part in [it[1] for it in seq if it[0] == "part"] + [upd]:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._train_body(P, False, part)
                graphs[part] = g
            P.graphs[key0] = graphs
        graphs = P.graphs[key0]
        run(lambda part: graphs[part].replay())

    def _sync_hook(self, P, world):
        """mms_sync_fn (include/mmsurv.h) of a plan: all-reduce freshly written BatchNorm accumulator words over the ranks."""
        if getattr(P, "sync_hook", None) is None:
            from . import distributed as D
            base0 = P.ws.data_ptr()
            ws64 = P.ws[:P.ws.numel() // 8 * 8].view(torch.float64)

            def hook(user, base, nrep, rstride, ncols, pstride, stream):
                try:
                    self.sync_collectives = getattr(self, "sync_collectives", 0) + 1      # (reported by bench.py --mode ddp --sync-bn)
                    t = ws64.as_strided((nrep, 2, ncols), (rstride, pstride, 1), (base - base0) // 8)
                    buf = t.contiguous()
                    D.allreduce_sum_(buf, world)
                    if buf.data_ptr() != t.data_ptr():
                        t.copy_(buf)
                    return 0
                except Exception as e:      # never raise through the C frame
                    print("mmsurv: SyncBN hook failed: %r" % (e,), flush=True)
                    return -2
            P.sync_hook = _lib.SYNC_FN(hook)
        return P.sync_hook

    def _ddp_step_syncbn(self, P, world):
        """Exact single-process semantics for a global batch of world*B patients (SURVEY.md section 8e ii-iii), eager launches.
        Collectives per step: ONE all-reduce per statistic-producing kernel (all replicas and both moments of its channels in one
        message): 121 in the forward (conv0, pool0, conv1 + conv2 of the 58 dense layers, 3 transitions), 121 in the backward, + the
        feature / label all-gathers and the gradient all-reduce -- 2 per BatchNorm layer and pass, the same count torch's SyncBatchNorm
        issues.  It cannot be lower without changing the arithmetic: layer l's conv2 normalises with the statistics of ITS conv1's
        output and layer l+1's conv1 with those of layer l's conv2 output, so the chain y1-stats(l) -> z-stats(l) -> y1-stats(l+1) is
        strictly sequential (the slab's OLDER channels are reduced once, when they are produced, and reused by every later layer).
        `self.sync_collectives` counts them; bench.py --mode ddp --sync-bn reports the count per step.
          * encoder: every BatchNorm3d statistic (forward sums, backward sums) is all-reduced between the kernel that produces it
            and the kernels that consume it (the drivers' hook);
          * heads (BatchNorm1d over the batch, gate, Cox risk set): replicated -- the encoder features and the small per-patient
            inputs are all-gathered and every rank runs the heads on the GLOBAL batch, then back-propagates its own patients'
            feature gradient through its encoder;
          * gradients: encoder weights hold rank-local partial sums -> SUM; head parameters and BatchNorm3d gamma/beta hold the
            global value on every rank -> pre-scaled by 1/world so that the same SUM returns them; then clip + Adam."""
        from . import distributed as D
        import torch.distributed as dist
        if not P.has_enc or P.fallback:
            raise RuntimeError("sync_bn: the DenseNet121-3D imaging models only")
        st = ops.stream()
        lib, prog = self.lib, self.prog
        rank = dist.get_rank() if dist.is_initialized() else 0
        B, (Dd, H, W) = P.B, P.dims
        Pg = self.plan(world * B, P.dims, heads_only=True)
        hook = self._sync_hook(P, world)
        cc = prog["ct_cols"]
        self.gflat.zero_(); self.sumsq.zero_()
        feats = P.buf["feats"]
        _lib.check(lib.mms_dn121_forward_sync(P.ws.data_ptr(), B, Dd, H, W, P.ct.data_ptr(), P.ptab, P.btab, feats[:, cc:].data_ptr(),
                                              feats.stride(0), world, hook, None, self._opts_arg(P), st), "mms_dn121_forward_sync")

        def gather(dst, src):
            tmp = torch.empty(world * src.shape[0], *src.shape[1:], device=self.device)
            D.all_gather_into(tmp.view(-1), src.contiguous().view(-1), world)
            dst.copy_(tmp.view(dst.shape))
        ew = prog.get("enc_width", 128)
        gather(Pg.buf["feats"][:, cc:cc + ew], feats[:, cc:cc + ew])
        for name in ("rna", "clin"):
            if name in P.buf:
                gather(Pg.buf[name], P.buf[name])
        if P.gate is not None:
            gather(Pg.mask, P.mask)
        if P.mix is not None:
            gather(Pg.mask2, P.mask2)
        gather(Pg.time, P.time); gather(Pg.event, P.event); gather(Pg.valid, P.valid)
        self._forward(Pg, True)                                   # heads only (Pg.has_enc is False)
        _lib.check(lib.mms_cox_fwd_bwd(ctypes.byref(Pg.cox), st), "mms_cox_fwd_bwd")
        self._backward_heads(Pg)
        P.dbuf["feats"][:, cc:cc + ew].copy_(Pg.dbuf["feats"][rank * B:(rank + 1) * B, cc:cc + ew])
        self._backward_encoder(P, hook=hook)
        if getattr(self, "_share", None) is None or self._share[0] != world:
            sh = torch.full_like(self.gflat, 1.0 / world)        # heads + BatchNorm3d parameters: global value on every rank
            o = 0
            bn_ids = {id(q) for m in prog["encoder"].modules() if isinstance(m, nn.BatchNorm3d) for q in m.parameters()}
            eids = {id(q) for q in prog["encoder"].parameters()}
            for q in self.params:
                if id(q) in eids and id(q) not in bn_ids:
                    sh[o:o + q.numel()] = 1.0                    # conv / class_layers weights: rank-local partial sums
                o += q.numel()
            self._share = (world, sh)
        self.gflat.mul_(self._share[1])
        D.allreduce_sum_(self.gflat, world)
        ad = Pg.adam
        _lib.check(lib.mms_grad_sumsq(ctypes.byref(ad), st), "mms_grad_sumsq")
        _lib.check(lib.mms_clip_adam(ctypes.byref(ad), st), "mms_clip_adam")

    def forward_eval(self, ct=None, rna=None, clinical=None, mask=None, use_graph=True):
        """Eval-mode forward -> (hazard [B] view of a static buffer, gate [B,3] or None)."""
        B = rna.shape[0]
        P = self.plan(B, tuple(ct.shape[-3:]) if ct is not None else None)
        self.load_batch(P, ct, rna, clinical, mask)
        self.sync_packs()
        if use_graph:
            if "eval" not in P.graphs:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self._forward(P, False)
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._forward(P, False)
                P.graphs["eval"] = g
            P.graphs["eval"].replay()
        else:
            self._forward(P, False)
        return P.buf["hz"][:, 0], (P.gatew if P.gate is not None else None)

    def reset_epoch_stats(self):
        self.acc.zero_()

    def check_b4(self):
        """The sticky time-out word of the block-4 persistent kernels (csrc/dn_b4.hip): a cluster whose workgroups never became
        co-resident (more persistent workgroups in flight than the chip holds) leaves garbage behind and must not pass silently --
        neither in training (epoch_stats) nor in validation / inference (validate_*, validate_lockstep, evaluate_model.py).  One
        device->host read per plan; the word is cleared before raising so that the workspace stays usable."""
        for P in self.plans.values():
            if getattr(P, "has_enc", False) and not P.fallback:
                if getattr(P, "b4_err", None) is None:
                    off, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
                    D, H, W = P.dims
                    _lib.check(self.lib.mms_dn121_region(P.B, D, H, W, b"b4_err", 0, ctypes.byref(off), ctypes.byref(nb)), "mms_dn121_region")
                    P.b4_err = P.ws[off.value:off.value + 4].view(torch.int32)
                if int(P.b4_err.item()) != 0:
                    P.b4_err.zero_()
                    raise RuntimeError("mmsurv: a hand-off of the block-4 persistent kernel timed out (too many persistent launches in "
                                       "flight at once?); the results since the last check are invalid -- rerun with dn_opts={'persist_b4': -1}")

    def epoch_stats(self):
        """-> dict(sum_loss, n_usable, sum_entropy, n_batches) (one device->host sync); checks the block-4 time-out word (check_b4)."""
        a = self.acc.tolist()
        self.check_b4()
        return dict(sum_loss=a[0], n_usable=a[1], sum_entropy=a[2], n_batches=a[3])


def engine_of(model, **kw):
    e = getattr(model, "_mms_engine", None)
    if e is None:
        e = SurvivalEngine(model, **kw)
    return e
