"""Fold groups: the K fold models of the reference's cross-validation loop advanced in lock-step on one GPU.

The reference trains fold after fold (final_multimodal.py:316-402, partial_modality_training.py:482-560,
simple_fusion.py:318-436).  The folds are independent models of identical architecture, and one batch-4 step of one
of them is far too small for 256 CUs (most of its 570 kernels are launch/latency bound).  A `FoldGroupEngine` owns G
`SurvivalEngine`s and issues every launch of the training step ONCE for the whole group through the `*_group` entry
points of include/mmsurv.h: same kernels, same per-model arithmetic, G times the workgroups per launch.

Per-model state stays per model (parameters, Adam moments, BatchNorm buffers, learning rate, dropout counters,
epoch accumulators): a fold trained in a group follows exactly the trajectory it follows alone.
"""
import ctypes

import torch

from . import _lib, ops
from .engine import SurvivalEngine, check_train_batch

_S = _lib.structs


def _arr(blocks):
    T = type(blocks[0])
    return (T * len(blocks))(*blocks)


def _ptrs(vals):
    return (ctypes.c_void_p * len(vals))(*vals)


class _GroupPlan:
    pass


class FoldGroupEngine:
    MAX = 10     # MMS_MAX_GROUP

    def __init__(self, models, **engine_kw):
        """models: 1..10 modules of the same class/shape, already on the GPU and not yet bound to an engine."""
        if not 1 <= len(models) <= self.MAX:
            raise ValueError("a fold group holds 1..%d models" % self.MAX)
        self.lib = _lib.load_library()
        dev = next(models[0].parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("FoldGroupEngine: models must be on the GPU (no CPU fallback)")
        n = sum(p.numel() for p in models[0].parameters())
        n += (-n) % 4
        G = len(models)
        self.device = dev
        # group-wide buffers so that zeroing the step's accumulators is three memsets for the whole group
        self.gflat_all = torch.zeros(G, n, device=dev)
        self.sumsq_all = torch.zeros(G, dtype=torch.float64, device=dev)
        self.entropy_all = torch.zeros(G, device=dev)
        self.engines = []
        for g, m in enumerate(models):
            if getattr(m, "_mms_engine", None) is not None:
                raise RuntimeError("model %d already has an engine; build the group from fresh models" % g)
            if sum(p.numel() for p in m.parameters()) + (-sum(p.numel() for p in m.parameters())) % 4 != n:
                raise ValueError("fold-group models must have identical shapes")
            slots = dict(gflat=self.gflat_all[g], sumsq=self.sumsq_all[g:g + 1], entropy=self.entropy_all[g:g + 1])
            self.engines.append(SurvivalEngine(m, _slots=slots, **engine_kw))
        kinds = {e.prog["kind"] for e in self.engines}
        if len(kinds) != 1:
            raise ValueError("fold-group models must be of one class, got %s" % sorted(kinds))
        if len({bytes(e.dn_opts) for e in self.engines}) != 1:
            raise ValueError("fold-group models must share one encoder shape (class_layers.out width) and launch options")
        self.plans = {}

    def __len__(self):
        return len(self.engines)

    # ---- plans -----------------------------------------------------------------------------------
    def plan(self, B, dims, members=None):
        members = tuple(range(len(self.engines))) if members is None else tuple(members)
        key = (B,) + (tuple(dims) if dims is not None else ()) + (members,)
        if key in self.plans:
            return self.plans[key]
        if B > 32:
            raise RuntimeError("fold groups batch the small-batch head kernels (<= 32 rows per model); train large-batch "
                               "RNA-seq-only folds one engine at a time (SurvivalEngine.train_step)")
        eng = [self.engines[i] for i in members]
        Ps = [e.plan(B, dims) for e in eng]
        if any(P.fallback and P.has_enc for P in Ps):
            raise NotImplementedError("fold groups drive the DenseNet121-3D encoder; the 3-conv fallback encoder has no "
                                      "group entry points (train those folds one at a time)")
        GP = _GroupPlan()
        GP.members, GP.eng, GP.Ps, GP.B, GP.dims = members, eng, Ps, B, (tuple(dims) if dims is not None else ())
        prog = eng[0].prog
        GP.ng = len(eng)
        GP.has_enc = Ps[0].has_enc
        if GP.has_enc:
            GP.ws = _ptrs([P.ws.data_ptr() for P in Ps])
            GP.x = _ptrs([P.ct.data_ptr() for P in Ps])
            GP.params = _ptrs([ctypes.addressof(P.ptab) for P in Ps])
            GP.buffers = _ptrs([ctypes.addressof(P.btab) for P in Ps])
            GP.grads = _ptrs([ctypes.addressof(P.gtab) for P in Ps])
            cc = prog["ct_cols"]
            GP.out = _ptrs([P.buf["feats"][:, cc:].data_ptr() for P in Ps])
            GP.dout = _ptrs([P.dbuf["feats"][:, cc:].data_ptr() for P in Ps])
            GP.ld = Ps[0].buf["feats"].stride(0)
        GP.mix = _arr([P.mix for P in Ps]) if Ps[0].mix is not None else None
        nl = len(prog["lins"])
        GP.lin_fwd = {t: [_arr([P.lin_fwd[t][i] for P in Ps]) for i in range(nl)] for t in (True, False)}
        GP.lin_bwd = [_arr([P.lin_bwd[i] for P in Ps]) for i in range(nl)]
        GP.gate = _arr([P.gate for P in Ps]) if Ps[0].gate is not None else None
        GP.cox = _arr([P.cox for P in Ps])
        GP.cox_eval = _arr([P.cox_eval for P in Ps])
        GP.adam = {True: _arr([P.adam_skip for P in Ps]), False: _arr([P.adam for P in Ps])}
        GP.graphs = {}
        # zeroing: the whole group's buffers when the group is complete and contiguous, else per member
        GP.full = members == tuple(range(len(self.engines)))
        self.plans[key] = GP
        return GP

    # ---- launches ----------------------------------------------------------------------------------
    def _forward(self, GP, train):
        st = ops.stream()
        lib, ng = self.lib, GP.ng
        prog = GP.eng[0].prog
        if GP.has_enc:
            B, (D, H, W) = GP.B, GP.dims
            _lib.check(lib.mms_dn121_forward_group(ng, GP.ws, B, D, H, W, GP.x, GP.params, GP.buffers, GP.out, GP.ld,
                                                   1 if train else 0, self._opts_arg(GP), st), "mms_dn121_forward_group")
        self._forward_heads(GP, train)

    def _forward_heads(self, GP, train):
        st = ops.stream()
        lib, ng = self.lib, GP.ng
        prog = GP.eng[0].prog
        lf = GP.lin_fwd[train]
        n_pre = prog["n_pre"]
        for i in range(n_pre):
            _lib.check(lib.mms_linear_fwd_group(lf[i], ng, st), "mms_linear_fwd_group")
        if GP.gate is not None:
            _lib.check(lib.mms_gate_fwd_group(GP.gate, ng, st), "mms_gate_fwd_group")
        if GP.mix is not None:
            _lib.check(lib.mms_missing_mix_fwd_group(GP.mix, ng, st), "mms_missing_mix_fwd_group")
        for i in range(n_pre, len(lf)):
            _lib.check(lib.mms_linear_fwd_group(lf[i], ng, st), "mms_linear_fwd_group")

    @staticmethod
    def _sync_packs(GP):
        """Derived conv2 packs of every member current (SurvivalEngine.sync_packs: a no-op unless the weights were changed outside the
        fused step since the last call)."""
        for e in GP.eng:
            e.sync_packs()

    def _zero(self, GP):
        if GP.full:
            self.gflat_all.zero_(); self.sumsq_all.zero_(); self.entropy_all.zero_()
        else:
            for e in GP.eng:
                e.gflat.zero_(); e.sumsq.zero_(); e.entropy.zero_()

    def _train_body(self, GP, skip_if_unusable):
        """zero-grad -> forward -> Cox -> backward -> clip -> Adam for every member, one launch sequence."""
        st = ops.stream()
        lib, ng = self.lib, GP.ng
        prog = GP.eng[0].prog
        self._zero(GP)
        self._forward(GP, True)
        _lib.check(lib.mms_cox_fwd_bwd_group(GP.cox, ng, st), "mms_cox_fwd_bwd_group")
        n_pre = prog["n_pre"]
        for i in range(len(GP.lin_bwd) - 1, n_pre - 1, -1):
            _lib.check(lib.mms_linear_bwd_group(GP.lin_bwd[i], ng, st), "mms_linear_bwd_group")
        if GP.gate is not None:
            _lib.check(lib.mms_gate_bwd_group(GP.gate, ng, st), "mms_gate_bwd_group")
        if GP.mix is not None:
            _lib.check(lib.mms_missing_mix_bwd_group(GP.mix, ng, st), "mms_missing_mix_bwd_group")
        for i in range(n_pre - 1, -1, -1):
            _lib.check(lib.mms_linear_bwd_group(GP.lin_bwd[i], ng, st), "mms_linear_bwd_group")
        if GP.has_enc:
            B, (D, H, W) = GP.B, GP.dims
            _lib.check(lib.mms_dn121_backward_group(ng, GP.ws, B, D, H, W, GP.x, GP.params, GP.dout, GP.ld, GP.grads, self._opts_arg(GP), st),
                       "mms_dn121_backward_group")
        ad = GP.adam[bool(skip_if_unusable)]
        _lib.check(lib.mms_grad_sumsq_group(ad, ng, st), "mms_grad_sumsq_group")
        _lib.check(lib.mms_clip_adam_group(ad, ng, st), "mms_clip_adam_group")

    # ---- state snapshot around graph warm-up ---------------------------------------------------------
    def _snapshot(self, eng):
        return [[e.flat.clone(), e.m.clone(), e.v.clone(), e.step_count.clone(), e.rng.clone(), e.acc.clone(),
                 [b.clone() for b in e.model.buffers()], e.acc_eval.clone()] for e in eng]

    def _restore(self, eng, snap):
        with torch.no_grad():
            for e, s in zip(eng, snap):
                e.flat.copy_(s[0]); e.m.copy_(s[1]); e.v.copy_(s[2]); e.step_count.copy_(s[3]); e.rng.copy_(s[4])
                e.acc.copy_(s[5]); e.acc_eval.copy_(s[7])
                for b, b0 in zip(e.model.buffers(), s[6]):
                    b.copy_(b0)
                e.sync_packs()         # (the roll-back changed the weights: rebuild the derived conv2 packs, outside any capture)

    def _opts_arg(self, GP):
        """`const MmsDnOpts*` of the group's driver calls (the members share one block: same class, same options).  The persistent
        per-block kernels (csrc/dn_cl.hip, dn_b4.hip) hand data between co-resident workgroups and one such launch per worker stream
        may be in flight: where those workgroups could outnumber the CUs, the per-layer path is taken (ops.persistent_opts) -- decided
        at launch (= graph-capture) time from the number of worker streams the process has created, and passed to the drivers as an
        ARGUMENT."""
        o = ops.persistent_opts(GP.eng[0].dn_opts, self.device, GP.ng, GP.B, GP.dims)
        GP.opts_live = o
        return ctypes.byref(o)

    def _graph(self, GP, key, body):
        if key not in GP.graphs:
            snap = self._snapshot(GP.eng)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                body()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self._restore(GP.eng, snap)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            GP.graphs[key] = g
            self._restore(GP.eng, snap)
        return GP.graphs[key]

    # ---- public ------------------------------------------------------------------------------------------
    def train_step(self, batches, members=None, skip_if_unusable=True, use_graph=True):
        """One optimisation step of every member on its own batch.  batches: one dict per member with the keyword
        arguments of SurvivalEngine.load_batch (ct, rna, clinical, mask, time, event, valid); all of one batch size."""
        members = tuple(range(len(self.engines))) if members is None else tuple(members)
        if len(batches) != len(members):
            raise ValueError("one batch per member")
        has_enc = self.engines[0].prog["encoder"] is not None
        B = batches[0]["rna"].shape[0]
        check_train_batch(B)
        dims = tuple(batches[0]["ct"].shape[-3:]) if has_enc else None
        for b in batches:
            if b["rna"].shape[0] != B or (has_enc and tuple(b["ct"].shape[-3:]) != dims):
                raise ValueError("fold-group batches must share one shape; split ragged tails into their own step")
        GP = self.plan(B, dims, members)
        for e, P, b in zip(GP.eng, GP.Ps, batches):
            e.load_batch(P, **b)
        self._sync_packs(GP)
        if not use_graph:
            self._train_body(GP, skip_if_unusable)
            return
        self._graph(GP, ("train", bool(skip_if_unusable)), lambda: self._train_body(GP, skip_if_unusable)).replay()

    def _gather_indexed(self, GP, cohort, idx):
        """Index copy (pinned ring -> device) + the ONE gather launch that assembles the group's batches in the plans' input buffers."""
        members, B = GP.members, GP.B
        key = id(cohort)
        cache = GP.__dict__.setdefault("gather", {})
        if key not in cache:
            dev_idx = torch.zeros(len(members), B, dtype=torch.int64, device=self.device)
            # ring of pinned staging buffers: the async copy reads the buffer when the stream gets there, so a buffer is
            # only rewritten after the event recorded behind its previous copy has completed
            pin = dict(bufs=[torch.zeros(len(members), B, dtype=torch.int64).pin_memory() for _ in range(4)],
                       evs=[None] * 4, k=0)
            blocks = _arr([e.gather_block(P, cohort, dev_idx[g]) for g, (e, P) in enumerate(zip(GP.eng, GP.Ps))])
            for e, P in zip(GP.eng, GP.Ps):
                if "valid" not in cohort:
                    P.valid.fill_(1.0)
            cache[key] = (dev_idx, pin, blocks, cohort)       # the cohort reference keeps the source pointers alive
        dev_idx, pin, blocks, _ = cache[key]
        k = pin["k"]
        pin["k"] = (k + 1) % len(pin["bufs"])
        if pin["evs"][k] is not None:
            pin["evs"][k].synchronize()
        pin["bufs"][k].copy_(idx)
        dev_idx.copy_(pin["bufs"][k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pin["evs"][k] = ev
        _lib.check(self.lib.mms_gather_rows_group(blocks, GP.ng, ops.stream()), "mms_gather_rows_group")

    def train_step_indexed(self, cohort, indices, members=None, skip_if_unusable=True, use_graph=True):
        """Same step with the batches named by patient indices into a cohort that lives in HBM (data.cohort_to) or in pinned host
        memory (data.cohort_pin: the gather launch reads the rows over PCIe): indices: [len(members)][B] integer array-like (host).  The batch assembly of the whole group is one small
        host-to-device copy of the indices plus ONE gather launch (mms_gather_rows_group) instead of ~10 torch
        indexing/copy kernels per member."""
        members = tuple(range(len(self.engines))) if members is None else tuple(members)
        idx = torch.as_tensor(indices, dtype=torch.int64)
        if idx.dim() != 2 or idx.shape[0] != len(members):
            raise ValueError("indices must be [len(members)][B]")
        B = idx.shape[1]
        check_train_batch(B)
        dims = tuple(cohort["image"].shape[-3:]) if self.engines[0].prog["encoder"] is not None else None
        GP = self.plan(B, dims, members)
        self._gather_indexed(GP, cohort, idx)
        self._sync_packs(GP)
        if not use_graph:
            self._train_body(GP, skip_if_unusable)
            return
        self._graph(GP, ("train", bool(skip_if_unusable)), lambda: self._train_body(GP, skip_if_unusable)).replay()

    def _eval_loss_body(self, GP):
        """Eval-mode forward of every member + the Cox value of its batch (no gradient) + the validation accumulators."""
        self._forward(GP, False)
        _lib.check(self.lib.mms_cox_fwd_bwd_group(GP.cox_eval, GP.ng, ops.stream()), "mms_cox_fwd_bwd_group")
        for e, P in zip(GP.eng, GP.Ps):
            e.acc_eval[:2] += P.cox_eval_out         # (the loss is 0 when the batch is unusable)
            e.acc_eval[3] += 1.0

    def eval_loss_step_indexed(self, cohort, indices, members=None):
        """validate_*'s step with the batches named by patient indices (as train_step_indexed): gather + ONE graph (eval-mode forward,
        Cox value of each member's batch, accumulators SurvivalEngine.acc_eval).  -> per member (hazard [B], (loss | usable) [2]):
        views of static buffers, valid until the next step of this plan."""
        members = tuple(range(len(self.engines))) if members is None else tuple(members)
        idx = torch.as_tensor(indices, dtype=torch.int64)
        if idx.dim() != 2 or idx.shape[0] != len(members):
            raise ValueError("indices must be [len(members)][B]")
        dims = tuple(cohort["image"].shape[-3:]) if self.engines[0].prog["encoder"] is not None else None
        GP = self.plan(idx.shape[1], dims, members)
        self._gather_indexed(GP, cohort, idx)
        self._sync_packs(GP)
        self._graph(GP, "evalloss", lambda: self._eval_loss_body(GP)).replay()
        return [(P.buf["hz"][:, 0], P.cox_eval_out) for P in GP.Ps]

    def forward_eval(self, batches, members=None, use_graph=True):
        """Eval-mode forward of every member -> list of (hazard [B] view, gate [B,3] or None) per member."""
        members = tuple(range(len(self.engines))) if members is None else tuple(members)
        B = batches[0]["rna"].shape[0]
        dims = tuple(batches[0]["ct"].shape[-3:]) if self.engines[0].prog["encoder"] is not None else None
        GP = self.plan(B, dims, members)
        for e, P, b in zip(GP.eng, GP.Ps, batches):
            e.load_batch(P, **b)
        self._sync_packs(GP)
        if use_graph:
            self._graph(GP, "eval", lambda: self._forward(GP, False)).replay()
        else:
            self._forward(GP, False)
        return [(P.buf["hz"][:, 0], (P.gatew if P.gate is not None else None)) for P in GP.Ps]

    def reset_epoch_stats(self):
        for e in self.engines:
            e.reset_epoch_stats()

    def epoch_stats(self):
        return [e.epoch_stats() for e in self.engines]
