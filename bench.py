#!/usr/bin/env python3
"""Headline benchmark: patients/sec of the multimodal survival training hot path on MI355X.

Default workload `c3` = the configuration BASELINE.json's metric is quoted on, N = 1 leg (BASELINE configs[2]):
  608-patient synthetic cohort with modality masks (142 CT / 427 RNA-seq / 587 clinical / 348 labelled), PartialModalityNet
  (DenseNet121-3D CT 64x64x32 + RNA-seq 5005 + clinical, softmax gate, gate-entropy term), 5-fold cross-validation over the
  348 labelled patients with all 260 unlabelled patients added to every training split
  (R/scripts/training/partial_modality_training.py:496-532), batch 4, Adam(lr 1e-4, wd 1e-4), clip_grad_norm_(1.0).
  The K = 5 fold models are advanced in lock-step by `training.train_epoch_lockstep` (what scripts/training/
  partial_modality_training.py runs): ONE launch sequence per lock-step step carries all fold models of a sub-group
  (default: sub-groups of 2 + 2 + 1 on three HIP streams).
  A "step" is one lock-step step: every fold model that still has a batch takes one optimisation step on it (zero-grad,
  forward, Cox partial likelihood on the labelled patients + gate entropy, backward, clip, Adam -- one replayed HIP graph per
  sub-group).  The timed region is WHOLE EPOCHS of the 5-fold job (135 steps each: 134 full batches + the ragged tail of every
  fold, entropy-only batches included) -- at least one epoch, more if --steps asks for more; `value` = training patients
  processed / wall time.  The cohort is resident in HBM before the timed region; each step's batches are gathered
  device-to-device by one launch fed by one small index copy.  `--h2d` adds the host-resident variant.

Other workloads: `--workload c2` (BASELINE configs[1]: MultiModalSurvivalNet, 109 complete patients, fold models in flight
configurable -- round 1's headline), `--workload c5` (BASELINE config 5: RNA-seq-only model, batch 2048).

  python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1: under torch.distributed.run (RANK / WORLD_SIZE set) every process is one rank; started plainly, bench.py launches its own N
ranks as children (torch.distributed.run --standalone-style on 127.0.0.1) BEFORE touching the GPU and relays rank 0's JSON line and the
exit code -- it never prints an n_gpus = 1 line for a --gpus N request.  `value` at N > 1 is BASELINE config 3's literal layout: ONE
5-fold cross-validation's folds dealt over the ranks (fold k -> rank k mod N, no data-path collective; "scaling": "strong" -- the job
is fixed, at N = 4 the ranks hold 2/1/1/1 fold models, at N = 8 three ranks idle).  The repeated-K-fold figure (rank r = its own
repetition of the 5-fold CV, per-GPU work fixed: weak scaling) is reported beside it as `repeated_kfold_patients_per_s`.
`--mode ddp` = BASELINE config 4's data-parallel step (one model, global batch N*B, gradient all-reduce over RCCL).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     dominant kernel family, algorithmic FLOPs / measured average launch duration vs the fp32 MFMA peak
  "cpu_baseline": the CPU oracle (torch fp32 restatement of the same model/loop) timed on this box's host cores.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
BLOCKS = ((6, 64), (12, 128), (24, 256), (16, 512))     # (layers, first-layer input channels) of DenseNet121


# ---- roofline leg: the dense-layer 3x3x3 convolution family, timed live ------------------------------------------------
def _stat_reps(M):                 # dn_net.hip make_plan: statistic-accumulator replicas of a level
    r, rows = 1, 2048
    while r < 8 and M // (2 * r) >= rows:
        r *= 2
    return r


def _conv3_nsplit(M, ng, gd):      # dn_net.hip conv3_nsplit (scratch assumed large enough)
    if 16 + 2 * (gd[1] * gd[2] + gd[2] + 1) <= 120:
        return 1                   # dn_ops.h mms_conv3_small_jn: the all-tap 16-row kernels of dn_c3s.hip, no tap split
    target = 256                   # (default launch-shape options: MmsDnOpts all zero)
    tiles = ((M + 31) // 32) * ng
    if tiles >= 256 and tiles >= target:
        return 1
    ns = max(1, min(27, (target + tiles - 1) // tiles))
    tpw = (27 + ns - 1) // ns
    return (27 + tpw - 1) // tpw


def _bwdw_msplit(M, members):      # the driver's own rule (csrc/dn_ops.h mms_conv3w_msplit), asked of the library
    from multimodal_survival_prediction_amd import _lib
    ms = _lib.load_library().mms_conv3_bwd_weight_msplit(int(M), int(members), None)
    assert ms > 0
    return ms


def _bwdw_members(G, block, layers):
    """(model, layer) members per weight-gradient launch (dn_net.hip: defer / ngw_blk): the launches are deferred to the end of their
    dense block and batched over layers when 2 G <= 10 -- blocks 2-4 greedily (10 // G * G members), block 1's layers split evenly
    over the fewest launches."""
    if 2 * G > 10:
        return G
    if block > 0:
        return (10 // G) * G
    nl = -(-layers * G // 10)
    return -(-layers // nl) * G


def _bwdw_multitap(M, G):          # dn_bwd.hip conv3w_mt_ok: which of the two weight-gradient kernels a launch runs on
    ms = _bwdw_msplit(M, G)
    chunk = (-(-M // ms) + 31) & ~31
    w = ms * G * 9
    return chunk >= 512 and w * 10 >= -(-w // 768) * 768 * 9


def measure_conv2_family(B, dims, device, G, reps=20, manifest=None):
    """The three kernels of the dense layers' 3x3x3 convolution (norm2/relu2/conv2 of MONAI's _DenseLayer): forward
    (mms_conv3_fwd_group), backward-data (mms_conv3_bwd_data_group) and weight gradient (mms_conv3_bwd_weight_group) -- 58
    launches each per lock-step step, together the largest share of GPU time in profiles/.  Each is timed live with HIP
    events on the launch stream (torch's current stream), launched exactly as the step launches it: one launch carries the G
    models of a sub-group, shape by shape (the four dense blocks) with the driver's own split factors, weighted by the launch
    counts.  Algorithmic FLOPs per launch of any of the three = members * 2 * M * 27 * 128 * 32: members = the G models for the forward,
    and the backward-data; the weight-gradient launches are DEFERRED to the end of their block and batched over layers by the driver
    (dn_net.hip flush_w: up to 10 // G * G (model, layer) members per launch in blocks 2-4, block 1's six layers split evenly over the
    fewest launches) -- timed that way.  (The per-layer launches of a dense block that
    runs as one persistent launch per pass, csrc/dn_cl.hip / dn_b4.hip, do not exist in the step: they are left out of the forward /
    backward-data ops' averages; the weight-gradient launches remain.)
    -> {op: (avg seconds per launch, avg FLOPs per launch, launches per step)}"""
    from multimodal_survival_prediction_amd import _lib, ops
    lib, S = _lib.load_library(), _lib.structs()
    D, H, W = dims
    gam, bet = torch.ones(128, device=device), torch.zeros(128, device=device)
    tot = {k: [0.0, 0.0, 0] for k in ("fwd", "bwd_data", "bwd_weight")}
    # dense blocks that run as ONE persistent launch per pass have no per-layer conv2 launches (dn_net.hip cluster_block; the residency
    # rule of fold_group.FoldGroupEngine._opts_arg)
    po, cw = ops.persistent_opts(ops.dn_opts(), device, G, B, dims), ops.cluster_workgroups(B, dims)
    cl_fwd = {2: cw[0] > 0 and po.persist_b3 > 0, 3: cw[1] > 0 and po.persist_b4 >= 0}
    cl_bwd = {2: False, 3: cw[1] == 8 and po.persist_b4 == 0}
    w = torch.randn(32, 128, 3, 3, 3, device=device) * 0.03
    wpf, wpb = ops.pack_conv3(w)
    wff, wfb = ops.pack_conv3_frag(w)
    for i, (layers, _) in enumerate(BLOCKS):
        gd = (D // 4 >> i, H // 4 >> i, W // 4 >> i)
        M = B * gd[0] * gd[1] * gd[2]
        R = _stat_reps(M)
        ns = _conv3_nsplit(M, G, gd)
        frag = i < 3 and 16 + 2 * (gd[1] * gd[2] + gd[2] + 1) <= 120 and not cl_fwd.get(i, False)      # dn_net.hip conv3_frag_block
        coords = ops.init_coords(B, gd, device)
        keep, fw, bd, bw = [], [], [], []
        for _ in range(G):
            y1 = torch.randn(M, 128, device=device)
            s, q = y1.double().sum(0), (y1.double() ** 2).sum(0)
            bn = ops.bnsrc(gam, bet, M, True, s, q)
            slab = torch.zeros(M, 256, device=device)
            dslab = torch.randn(M, 256, device=device)
            ost = torch.zeros(R, 2, 256, dtype=torch.float64, device=device)
            bst = torch.zeros(R, 2, 128, dtype=torch.float64, device=device)
            dbn = torch.zeros(M, 128, device=device)
            part = torch.zeros(max(ns, 1) * M * 128, device=device) if ns > 1 else None
            dwp = torch.zeros(27 * 32 * 128, device=device)
            out, dz = slab[:, 64:96], dslab[:, 64:96]
            keep.append((y1, s, q, slab, dslab, ost, bst, dbn, part, dwp))
            f = S["Conv3FwdP"](y1.data_ptr(), coords.data_ptr(), ops.dims3(gd), M, (wff if frag else wpf).data_ptr(), out.data_ptr(), out.stride(0), bn,
                               ost[0, 0, 64:].data_ptr(), ost[0, 1, 64:].data_ptr(), ops.ptr(part), ns, R, 2 * 256, 1 if frag else 0)
            d = S["Conv3BwdDataP"](dz.data_ptr(), dz.stride(0), coords.data_ptr(), ops.dims3(gd), M, (wfb if frag else wpb).data_ptr(), y1.data_ptr(), bn,
                                   dbn.data_ptr(), bst[0, 0].data_ptr(), bst[0, 1].data_ptr(), ops.ptr(part), ns, R, 2 * 128, 1 if frag else 0)
            fw.append(f); bd.append(d)
        # weight gradient: one launch carries `nw` (model, layer) members -- G for block 1, the deferred + batched count for blocks 2-4
        nw = _bwdw_members(G, i, layers)
        for k in range(nw):           # (every member has its own activations and gradients, as the layers of a block have in the step)
            if k < G:
                y1, s_, q_, dslab = keep[k][0], keep[k][1], keep[k][2], keep[k][4]
            else:
                y1 = torch.randn(M, 128, device=device)
                s_, q_ = y1.double().sum(0), (y1.double() ** 2).sum(0)
                dslab = torch.randn(M, 256, device=device)
            bn = ops.bnsrc(gam, bet, M, True, s_, q_)
            dz = dslab[:, 64:96]
            dwk = torch.zeros(27 * 32 * 128, device=device)
            keep.append((y1, s_, q_, dslab, dwk))
            bw.append(S["Conv3BwdWP"](y1.data_ptr(), coords.data_ptr(), ops.dims3(gd), M, bn, dz.data_ptr(), dz.stride(0),
                                      dwk.data_ptr(), _bwdw_msplit(M, nw), 2))
        arrs = {"fwd": ((S["Conv3FwdP"] * G)(*fw), lib.mms_conv3_fwd_group, G, layers),
                "bwd_data": ((S["Conv3BwdDataP"] * G)(*bd), lib.mms_conv3_bwd_data_group, G, layers),
                "bwd_weight": ((S["Conv3BwdWP"] * nw)(*bw), lib.mms_conv3_bwd_weight_group, nw, layers * G / nw)}      # launches per step: members / members per launch
        for op, (arr, fn, nm, nlaunch) in arrs.items():
            if (op == "fwd" and cl_fwd.get(i, False)) or (op == "bwd_data" and cl_bwd.get(i, False)):
                continue
            for _ in range(3):
                _lib.check(fn(arr, nm, None, ops.stream()), op)          # (launch-shape options: NULL = the defaults the step runs with)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if manifest is None:
                # as the step issues them: from a captured HIP graph.  Issued one by one from Python the 8-12 us launches of blocks 2-4 are
                # HOST-bound (ctypes call + launch path per kernel): the same kernels read 12.1 .. 15.3 us on different boxes of the pool.
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for _ in range(reps):
                        _lib.check(fn(arr, nm, None, ops.stream()), op)
                graph.replay()                                           # (first replay: upload)
                e0.record()
                graph.replay()
                e1.record()
            else:                                                        # --roofline-only (the rocprofv3 kernel-trace / PMC passes): plain launches
                e0.record()
                for _ in range(reps):
                    _lib.check(fn(arr, nm, None, ops.stream()), op)
                e1.record()
            torch.cuda.synchronize()
            tot[op][0] += e0.elapsed_time(e1) * 1e-3 / reps * nlaunch
            tot[op][1] += nm * 2.0 * M * 27 * 128 * 32 * nlaunch
            tot[op][2] += nlaunch
            if manifest is not None:      # what this leg launched, in order: 3 warm-up + `reps` timed calls each (tools/pmc_conv2.py reads it)
                manifest.append(dict(G=G, block=i, op=op, M=M, members=nm, launches_per_step=nlaunch, calls=3 + reps))
    return {op: (t / n, f / n, n) for op, (t, f, n) in tot.items()}


def _roof_name(op, B, dims, group_sizes):
    """The kernels rocprofv3 shows for the launches of an op (what the launchers in csrc/dn_fwd.hip / dn_bwd.hip pick at these sizes)."""
    if op == "fwd":
        return "mms_conv3_fwd_group = conv3_fwd_mt_kernel (block 1) / conv3s_fwd_kernel (blocks 2-4)"
    if op == "bwd_data":
        return "mms_conv3_bwd_data_group = conv3_bwd_data_mt_kernel (block 1) / conv3s_bwd_data_kernel (blocks 2-4)"
    D, H, W = dims
    ms = [B * (D // 4 >> i) * (H // 4 >> i) * (W // 4 >> i) for i in range(4)]
    mt = sorted({_bwdw_members(G, i, BLOCKS[i][0]) for G in group_sizes for i, M in enumerate(ms) if _bwdw_multitap(M, _bwdw_members(G, i, BLOCKS[i][0]))})
    name = "mms_conv3_bwd_weight_group = tile_gemm_kernel<Conv3BwdWOp>"
    if mt:
        name += " (conv3_bwdw_mt_kernel for the %s-member launches of block 1)" % "/".join(map(str, mt))
    return name


def roofline_block(B, dims, dev, group_sizes, with_manifest=False):
    """`roofline` object of the JSON line.  group_sizes: models per launch of the sub-groups the timed region ran (e.g. (3, 2));
    per op the launches of all sub-groups are pooled (time and FLOPs summed).  The dominant kernel = the op with the largest
    time per lock-step step; the other two are listed beside it."""
    fam = {}
    manifest = [] if with_manifest else None
    for G in group_sizes:
        for op, (t, f, nl) in measure_conv2_family(B, dims, dev, G, manifest=manifest).items():
            a = fam.setdefault(op, [0.0, 0.0, 0, 0])
            a[0] += t; a[1] += f; a[2] += 1; a[3] = nl
    dom = max(fam, key=lambda k: fam[k][0] * fam[k][3])          # largest time per lock-step step = average x launches
    t, f, n, _ = fam[dom]
    traffic = None      # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside bench.py)
    try:
        with open(os.path.join(ROOT, "profiles", "r04_pmc_conv2_traffic.json")) as fh:
            j = json.load(fh)
        if tuple(j.get("sub_groups", ())) == tuple(group_sizes):          # the committed passes ran on this launch configuration
            traffic = j[dom]["avg_hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError, TypeError):
        pass
    extra = {"manifest": manifest} if with_manifest else {}
    return {**extra, "bound": "mfma", "kernel": _roof_name(dom, B, dims, group_sizes) + f"; {fam[dom][3]:.1f} launches per lock-step step and sub-group (the weight-gradient launches of dense blocks 2-4 carry up to 10 (model, layer) members, as the step batches them), sub-groups of {'+'.join(map(str, group_sizes))} fold models",
            "achieved": f / t / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": f / t / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "traffic": traffic, "avg_launch_us": t / n * 1e6, "avg_flops_per_launch": f / n,
            "family": {op: {"avg_launch_us": v[0] / v[2] * 1e6, "launches_per_step_and_sub_group": v[3], "achieved": v[1] / v[0] / 1e12,
                            "frac": v[1] / v[0] / 1e12 / PEAK_FP32_MFMA_TFLOPS} for op, v in fam.items()}}


def _host_cores():
    """-> (physical cores available to this process, torch intra-op threads actually used)"""
    threads = torch.get_num_threads()
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or threads
        logical = psutil.cpu_count(logical=True) or phys
        avail = len(os.sched_getaffinity(0))
        phys = max(1, min(phys, avail * phys // max(logical, 1) if avail < logical else phys))
    except Exception:
        phys = threads
    return phys, threads


# ---- CPU baselines (the oracle = torch fp32 restatement; bench.py's cpu_baseline leg is one of the three places allowed to use it)
def cpu_baseline_partial(cohort, train_idx, steps, B):
    """The CPU oracle's train_epoch_partial loop body (oracle/loops.py, partial_modality_training.py:382-435) on the SAME cohort
    and fold split, on this box's host cores: bounded sample of `steps` batches after one warm-up batch."""
    from oracle import loops as OLP
    from oracle import models as OM
    torch.manual_seed(0)
    model = OM.PartialModalityNet(rna_dim=cohort["rnaseq"].shape[1], use_monai=True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4)
    cores, threads = _host_cores()

    def loader(lo, hi):
        for i in range(lo, hi):
            j = train_idx[i * B:(i + 1) * B]
            yield dict(image=cohort["image"][j], rnaseq=cohort["rnaseq"][j], clinical=cohort["clinical"][j], label=cohort["label"][j],
                       mask=cohort["mask"][j], has_survival=cohort["has_survival"][j].tolist())
    OLP.train_epoch_partial(model, loader(0, 1), opt, "cpu")
    t0 = time.perf_counter()
    OLP.train_epoch_partial(model, loader(1, steps + 1), opt, "cpu")
    dt = time.perf_counter() - t0
    return dict(value=steps * B / dt, unit="patients/s", cores=cores, threads=threads, kind="port", full_epoch=False,
                sample=f"{steps} batches (batch {B}) of the torch-fp32 CPU oracle's train_epoch_partial on fold 1's training split "
                       f"of the same cohort, after 1 warm-up batch, {dt:.1f} s (a bounded sample, not BASELINE.md's full epoch)")


def cpu_baseline_final(cohort, train_idx, steps, B):
    """Config 2: the CPU oracle's final_multimodal loop body."""
    from oracle import losses as OL
    from oracle import models as OM
    torch.manual_seed(0)
    model = OM.MultiModalSurvivalNet(rna_dim=cohort["rnaseq"].shape[1], use_monai=True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4)
    model.train()
    cores, threads = _host_cores()

    def one(i):
        j = train_idx[i * B:(i + 1) * B]
        hz = model(cohort["image"][j], cohort["rnaseq"][j], cohort["clinical"][j])
        loss = OL.cox_loss(hz, cohort["label"][j, 1], cohort["label"][j, 0])
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()

    one(0)
    t0 = time.perf_counter()
    for i in range(1, steps + 1):
        one(i)
    dt = time.perf_counter() - t0
    return dict(value=steps * B / dt, unit="patients/s", cores=cores, threads=threads, kind="port", full_epoch=False,
                sample=f"{steps} training steps (batch {B}) of the torch-fp32 CPU oracle after 1 warm-up step, {dt:.1f} s")


# ---- workload c3: BASELINE config 3 (the metric's configuration) -------------------------------------------------------------
def run_config3(args, world, rank, dev):
    import numpy as np
    from multimodal_survival_prediction_amd import data, distributed as D, models
    from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
    from multimodal_survival_prediction_amd.training import train_epoch_lockstep, validate_lockstep

    B, dims, K = args.batch, tuple(args.volume), args.folds
    cohort_cpu = data.make_cohort(n=args.patients, dims=dims, rna_dim=5005, seed=608, complete=False)
    cohort = data.cohort_to(cohort_cpu, dev)                      # resident in HBM before the timed region
    has = cohort_cpu["has_survival"].numpy()
    survival, non_survival = np.nonzero(has)[0], np.nonzero(~has)[0]

    def cv_loaders(random_state):
        folds = data.kfold_indices(len(survival), K, seed=random_state)
        tr = [np.concatenate([survival[f[0]], non_survival]) for f in folds]          # partial_modality_training.py:508-513
        va = [survival[f[1]] for f in folds]
        return ([data.BatchLoader(cohort, t, B, shuffle=True, seed=random_state + k, lazy=True, with_valid=True) for k, t in enumerate(tr)],
                [data.BatchLoader(cohort, v, B, shuffle=False, lazy=True, with_valid=True) for v in va], tr, folds)

    # N = 1: the 5-fold CV on this GPU.  N > 1: the SAME cross-validation (KFold random_state 42, fold k's model seeded 42 + k on every
    # rank), its folds dealt over the ranks: rank r steps the fold models k with k mod N == r (distributed.folds_of_rank) -- BASELINE
    # config 3's literal layout, no data-path collective.  Every rank builds all K models (the repeated-K-fold side leg uses them).
    train_loaders, val_loaders, train_sets, folds = cv_loaders(42)
    ms = []
    for k in range(K):
        torch.manual_seed(42 + k)
        ms.append(models.PartialModalityNet(rna_dim=5005).to(dev).train())
    group = FoldGroupEngine(ms, lr=1e-4, weight_decay=1e-4, adamw=False, gate_entropy_weight=0.01)
    conc = args.lockstep_streams
    steps_per_epoch = max(len(l) for l in train_loaders)
    patients_per_epoch = sum(len(t) for t in train_sets)
    mine = tuple(D.folds_of_rank(K, world, rank)) if world > 1 else tuple(range(K))

    def epoch(members=None, loaders=None):
        mem = tuple(range(K)) if members is None else tuple(members)
        if not mem:
            return []
        ld = train_loaders if loaders is None else loaders
        return train_epoch_lockstep(group, [ld[g] for g in mem], "partial", members=mem, concurrent=conc)

    def timed(fn):
        torch.cuda.synchronize(); D.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(); D.barrier(); torch.cuda.synchronize()
        return D.max_over_ranks(time.perf_counter() - t0, dev), r

    n_warm = max(1, math.ceil(args.warmup / steps_per_epoch))       # whole epochs: every sub-group / ragged-tail graph gets captured
    for _ in range(n_warm):
        epoch(mine)
    n_ep = max(1, math.ceil(args.steps / steps_per_epoch))
    last = []

    def run_epochs():
        for _ in range(n_ep):
            last[:] = epoch(mine)
    dt, _ = timed(run_epochs)
    value = n_ep * patients_per_epoch / dt          # the whole job's training patients (all folds, all ranks) per second
    if args.timed_only:
        if rank == 0:
            print(f"timed-only: {value:.1f} patients/s, {dt / (n_ep * steps_per_epoch) * 1e3:.3f} ms per lock-step step, "
                  f"{dt / n_ep:.3f} s per epoch", flush=True)
        D.barrier()
        return
    from multimodal_survival_prediction_amd.training import subgroup_sizes
    sub_groups = subgroup_sizes(len(mine) if mine else 1, conc)       # what training.train_epoch_lockstep stepped on this GPU
    import torch.distributed as tdist
    backend = tdist.get_backend() if tdist.is_initialized() else None

    # side leg at N > 1: repeated K-fold (rank r = repetition r of the 5-fold CV on the same cohort, KFold random_state 42 + r, all K
    # fold models of this rank in lock-step): per-GPU work fixed as N grows = weak scaling, no collective
    repeated = None
    if world > 1:
        rep_loaders = cv_loaders(42 + rank)[0]
        epoch(None, rep_loaders)                                     # capture the full-group graphs
        dtr, _ = timed(lambda: epoch(None, rep_loaders))
        repeated = world * patients_per_epoch / dtr
    # one fold model alone on the GPU (what a rank with a single fold of config 3 runs)
    n1 = max(20, min(steps_per_epoch, args.steps // 3))
    idx1 = train_loaders[0].idx

    def chain(n):
        for i in range(n):
            group.train_step_indexed(train_loaders[0].view, idx1[(i % (len(idx1) // B)) * B:][:B].view(1, B), members=(0,),
                                     skip_if_unusable=False)
    chain(3)
    dt1, _ = timed(lambda: chain(n1))
    dt1 /= n1
    # validation (reported separately): validate_lockstep over the K validation splits
    validate_lockstep(group, val_loaders, "partial", dev, concurrent=args.validation_streams)
    dtv, val = timed(lambda: validate_lockstep(group, val_loaders, "partial", dev, concurrent=args.validation_streams))
    n_val = sum(len(l.idx) for l in val_loaders)
    for e_ in group.engines:
        e_.model.train()
    # host-resident variant (the reference moves every batch host -> device, final_multimodal.py:244-247)
    h2d = run_h2d_epoch(args, group, cohort_cpu, train_sets, B, dev, timed, conc) if args.h2d else None
    # the same kernels with the chip filled: 2 concurrent groups x 10 lock-step fold models (four cross-validations in flight)
    many = None
    if args.many_folds:
        many = run_many(args, cohort, train_sets, B, dev, timed, models, FoldGroupEngine)

    if rank == 0:
        out = {
            "metric": "patients/sec per epoch (training epoch of the K-fold job: fwd + Cox + gate entropy + bwd + clip + Adam; HBM-resident "
                      "cohort, batches gathered on the device inside the timed region -- the same epoch fed from pinned host memory over PCIe: config.h2d)",
            "value": value, "unit": "patients/s", "n_gpus": world, "steps": n_ep * steps_per_epoch,
            "warmup": n_warm * steps_per_epoch, "ms_per_step": dt / (n_ep * steps_per_epoch) * 1e3, "higher_is_better": True,
            "scaling": None if world == 1 else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": ("BASELINE config 3, N=1 leg" if world == 1 else "BASELINE config 3, one 5-fold CV sharded over %d GPUs" % world) +
                            " (the metric's configuration): %d synthetic patients with modality masks "
                            "(%d CT / %d RNA-seq / %d clinical / %d labelled), PartialModalityNet (DenseNet121-3D CT %dx%dx%d + RNA-seq 5005 + "
                            "clinical, softmax gate + 0.01 x gate entropy), %d-fold CV over the labelled patients with all %d unlabelled "
                            "patients in every training split (%d training patients per epoch over the folds), batch %d, Adam lr 1e-4 wd "
                            "1e-4, clip 1.0; %d whole epoch(s) timed, ragged tails and entropy-only batches included"
                            % (args.patients, int(cohort_cpu["mask"][:, 0].sum()), int(cohort_cpu["mask"][:, 1].sum()),
                               int(cohort_cpu["mask"][:, 2].sum()), len(survival), dims[0], dims[1], dims[2], K, len(non_survival),
                               patients_per_epoch, B, n_ep),
                "step": "one lock-step step of the K-fold job: every fold model takes one batch (<= %d x %d patients)" % (K, B),
                "steps_requested": args.steps, "warmup_requested": args.warmup, "epochs_timed": n_ep, "steps_per_epoch": steps_per_epoch,
                "train_patients_per_epoch": patients_per_epoch, "seconds_per_epoch": dt / n_ep,
                "global_batch": K * B,
                "parallelism": (f"1 GPU: the {K} fold models in lock-step as sub-groups of {'+'.join(map(str, sub_groups))} on "
                                f"{len(sub_groups)} HIP stream(s), one step graph per sub-group" if world == 1 else
                                f"kfold-shard x{world}: ONE {K}-fold CV, fold k on rank k mod {world} (rank 0 holds {len(mine)} fold model(s), "
                                f"{max(0, world - K)} rank(s) idle), no data-path collective (torch.distributed backend {backend}); value = the "
                                f"CV's training patients / max-over-ranks epoch time"),
                "fold_models_in_flight": K, "lockstep_streams": conc,
                "rccl_ranks_seen": (tdist.get_world_size() if backend == "nccl" else 0),
                "repeated_kfold_patients_per_s": repeated,
                "single_chain_patients_per_s": B / dt1, "single_chain_ms_per_step": dt1 * 1e3,
                "validation_patients_per_s": n_val / dtv,
                "h2d": h2d, "many_folds": many,
                "hip_graph": True, "bn": "per-model batch statistics", "cox": "per-batch risk sets, Efron ties = torchsurv's rule (identical to Breslow on the cohort's distinct times)",
                "train_loss_last_epoch": [[float(a), float(b)] for a, b in last],
                "val_loss_cindex": [[float(a), float(b)] for a, b in val]},
        }
        out["roofline"] = roofline_block(B, dims, dev, sub_groups)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_partial(cohort_cpu, torch.as_tensor(train_sets[0]), args.cpu_steps, B)
        print(json.dumps(out), flush=True)
    D.barrier()


def run_h2d_epoch(args, group, cohort_cpu, train_sets, B, dev, timed, conc):
    """One epoch of the same K-fold job with the cohort in PINNED HOST memory (the reference keeps its data on the host and does
    `.to(device)` per batch, final_multimodal.py:244-247): batches are named by index exactly as in the headline leg, and the one
    gather launch per sub-group step reads the patients' rows over PCIe straight into the step graph's inputs (rows of a missing
    modality are zero-filled, not read) -- the host-to-device copy is inside the timed region."""
    from multimodal_survival_prediction_amd import data
    from multimodal_survival_prediction_amd.training import train_epoch_lockstep
    pinned = data.cohort_pin(cohort_cpu)
    loaders = [data.BatchLoader(pinned, t, B, shuffle=True, seed=142 + k, lazy=True, with_valid=True) for k, t in enumerate(train_sets)]
    train_epoch_lockstep(group, loaders, "partial", concurrent=conc)
    dt, _ = timed(lambda: train_epoch_lockstep(group, loaders, "partial", concurrent=conc))
    return {"patients_per_s": sum(len(t) for t in train_sets) / dt,
            "what": "same epoch, cohort in pinned host memory, every batch read over PCIe by the step's gather launch inside the timed region"}


def run_many(args, cohort, train_sets, B, dev, timed, models, FoldGroupEngine):
    """2 concurrent groups x 10 lock-step PartialModalityNet fold models (four cross-validations in flight): the throughput of the
    same kernels when the launches fill the chip -- a capability figure, not the metric."""
    F, G = 2, 10
    groups, streams, orders = [], [], []
    for f in range(F):
        ms = []
        for g in range(G):
            torch.manual_seed(1000 + f * G + g)
            ms.append(models.PartialModalityNet(rna_dim=5005).to(dev).train())
            gen = torch.Generator().manual_seed(7 + f * G + g)
            t = torch.as_tensor(train_sets[(f * G + g) % len(train_sets)])
            orders.append(t[torch.randperm(len(t), generator=gen)])
        groups.append(FoldGroupEngine(ms, lr=1e-4, weight_decay=1e-4, adamw=False, gate_entropy_weight=0.01))
    from multimodal_survival_prediction_amd import data, ops
    streams = ops.worker_streams(dev, F)                 # the process-wide streams the K-fold epoch's sub-groups stepped on
    view = data.gather_view(cohort, True)

    def run(n):
        for u in range(n):
            f, k = u % F, u // F
            with torch.cuda.stream(streams[f]):
                idx = [orders[f * G + g][(k % (len(orders[f * G + g]) // B)) * B:][:B] for g in range(G)]
                groups[f].train_step_indexed(view, torch.stack(idx), skip_if_unusable=False)
    run(3 * F)
    n = 12
    dt, _ = timed(lambda: run(n))
    return {"patients_per_s": n * G * B / dt, "fold_models_in_flight": F * G}


# ---- workload c5 --------------------------------------------------------------------------------------------------------------
def run_config5(args, dev):
    """BASELINE config 5 (`--workload c5`, single GPU): RNA-seq-only model 5005 -> 1024 -> 512 -> 256 -> 1 at batch 2048, one full
    training step per replay (zero-grad, forward, O(B^2) Cox partial likelihood, backward, AdamW; train_rnaseq_only.py:153-176)
    as a captured HIP graph.  roofline: the first-layer forward GEMM (2 * B * 5005 * 1024 FLOP per launch; the launch sequence
    timed live with HIP events includes its 13 us of output zeroing + column statistics); cpu_baseline: the oracle's loop body."""
    import ctypes
    import numpy as np
    from multimodal_survival_prediction_amd import _lib, models
    from multimodal_survival_prediction_amd.training import FusedOptimizer
    B = 2048 if args.batch == 4 else args.batch
    torch.manual_seed(0)
    net = models.RNASeqSurvivalModel(input_dim=5005).to(dev).train()
    fo = FusedOptimizer(net, lr=1e-4, weight_decay=1e-3, adamw=True, max_norm=0.0)
    rng = np.random.default_rng(0)
    rna = torch.tensor(rng.normal(0, 1, (B, 5005)).astype(np.float32), device=dev)
    t = torch.tensor((rng.exponential(1000, B) + 1 + np.arange(B) * 1e-3).astype(np.float32), device=dev)
    e = torch.tensor((rng.random(B) < 0.6).astype(np.float32), device=dev)

    def step():
        fo.engine.train_step(None, rna, time=t, event=e, skip_if_unusable=False, use_graph=not args.no_graph)
    for _ in range(max(args.warmup, 3)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    if args.timed_only:
        print(f"timed-only (config 5): {B / dt:.1f} patients/s, {dt * 1e3:.3f} ms/step", flush=True)
        return
    out = {"metric": "patients/sec per epoch (training: fwd + Cox + bwd + AdamW)", "value": B / dt, "unit": "patients/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": None,
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"BASELINE config 5: RNASeqSurvivalModel (5005-1024-512-256-1), batch {B}, Cox over the {B}-patient risk set, "
                                  "AdamW lr 1e-4 wd 1e-3", "global_batch": B, "parallelism": "single GPU, one step graph",
                      "hip_graph": not args.no_graph, "mean_train_loss": fo.engine.epoch_stats()["sum_loss"] / (args.steps + max(args.warmup, 3))}}
    P = fo.engine.plans[(B,)]
    q, lib = P.big_lin[True][0], _lib.load_library()
    P.big_stats.zero_()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    from multimodal_survival_prediction_amd import ops
    for i in range(3 + reps):
        if i == 3:
            e0.record()
        _lib.check(lib.mms_linear_big_fwd(ctypes.byref(q), ops.stream()), "mms_linear_big_fwd")
    e1.record()
    torch.cuda.synchronize()
    avg_t, flops = e0.elapsed_time(e1) * 1e-3 / reps, 2.0 * B * 5005 * 1024
    out["roofline"] = {"bound": "mfma", "kernel": "mms_linear_big_fwd of the first layer = lin_zero_y_kernel + lin_fwd_wide_kernel + lin_colstats_kernel",
                       "achieved": flops / avg_t / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                       "frac": flops / avg_t / 1e12 / PEAK_FP32_MFMA_TFLOPS, "traffic": None, "avg_launch_us": avg_t * 1e6,
                       "avg_flops_per_launch": flops}
    if not args.no_cpu_baseline:
        from oracle import losses as OL
        from oracle import models as OM
        ref = OM.RNASeqSurvivalModel(input_dim=5005).train()
        opt = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=1e-3)
        x, tc, ec = rna.cpu(), t.cpu(), e.cpu().bool()
        for i in range(args.cpu_steps + 1):
            if i == 1:
                c0 = time.perf_counter()
            opt.zero_grad(); OL.neg_partial_log_likelihood(ref(x).squeeze(), ec, tc).backward(); opt.step()
        cdt = time.perf_counter() - c0
        out["cpu_baseline"] = dict(value=args.cpu_steps * B / cdt, unit="patients/s", cores=_host_cores()[0], threads=torch.get_num_threads(), kind="port", full_epoch=False,
                                   sample=f"{args.cpu_steps} training steps (batch {B}) of the torch-fp32 CPU oracle after 1 warm-up step, {cdt:.1f} s")
    print(json.dumps(out), flush=True)


# ---- workload c2 (round 1's headline) and --mode ddp (config 4's step) ---------------------------------------------------------
def run_config2(args, world, rank, dev):
    """BASELINE configs[1]: MultiModalSurvivalNet, 109 complete patients; a step = one batch of 4 patients of ONE fold model, the K
    timed steps dealt over --concurrent-folds groups of --fold-group lock-step fold models.  --mode ddp: one model, this rank's
    shard of the global batch per step, gradient all-reduce (RCCL) between backward and clip+Adam."""
    from multimodal_survival_prediction_amd import data, distributed as D, models
    from multimodal_survival_prediction_amd.training import FusedOptimizer

    B, dims, rna_dim = args.batch, tuple(args.volume), 5005
    cohort_cpu = data.make_cohort(n=109, dims=dims, rna_dim=rna_dim, seed=608, complete=True)
    cohort = data.cohort_to(cohort_cpu, dev)                      # resident in HBM before the timed region
    folds = data.kfold_indices(cohort["n"], 5, seed=42)
    ddp = args.mode == "ddp" and world > 1
    one = args.mode == "ddp"               # config 4's per-rank problem: ONE model (at N = 1: the same step without its collectives)
    F = 1 if one else max(1, args.concurrent_folds)
    G = 1 if one else max(1, min(args.fold_group, 10))
    engines, groups, streams, orders = [], [], [], []          # engines/orders: one per fold model, index f * G + g
    for f in range(F):
        ms = []
        for g in range(G):
            unit = rank * F * G + f * G + g
            fold = 0 if ddp else unit % 5
            train_idx = torch.as_tensor(folds[fold][0])
            torch.manual_seed(42 if ddp else 42 + unit)           # ddp: identical initial weights on every rank
            model = models.MultiModalSurvivalNet(rna_dim=rna_dim).to(dev)
            model.train()
            ms.append(model)
            gen = torch.Generator().manual_seed(7 + f * G + g)
            orders.append(train_idx[torch.randperm(len(train_idx), generator=gen)].to(dev if G == 1 else "cpu"))
        if G > 1:
            from multimodal_survival_prediction_amd.fold_group import FoldGroupEngine
            groups.append(FoldGroupEngine(ms, lr=1e-4, weight_decay=1e-4, adamw=False))
            engines.extend(groups[-1].engines)
        else:
            engines.append(FusedOptimizer(ms[0], lr=1e-4, weight_decay=1e-4, adamw=False).engine)
    from multimodal_survival_prediction_amd import ops as _ops
    streams = _ops.worker_streams(dev, F)
    gb = B * world if ddp else B                                  # patients per step handled by one model

    def batch_of(m, k):
        order = orders[m]
        nb = len(order) // gb                                     # full (global) batches of the fold's train split
        o = (k % nb) * gb + (rank * B if ddp else 0)              # ddp: this rank's shard of the global batch
        j = order[o:o + B]
        lab = cohort["label"][j]
        return dict(ct=cohort["image"][j], rna=cohort["rnaseq"][j], clinical=cohort["clinical"][j], time=lab[:, 0], event=lab[:, 1])

    def launch(u, nf, members):
        """launch unit u: `members` (<= G) fold models of group u % nf each take one step (one launch sequence)"""
        f = u % nf
        k = u // nf
        with torch.cuda.stream(streams[f]):
            if G > 1:     # batches named by patient index; the group gathers them from the HBM-resident cohort in one launch
                idx = [orders[f * G + g][(k % (len(orders[f * G + g]) // B)) * B:][:B] for g in range(members)]
                groups[f].train_step_indexed(cohort, torch.stack(idx), members=tuple(range(members)),
                                             skip_if_unusable=True, use_graph=not args.no_graph)
            else:
                engines[f].train_step(skip_if_unusable=True, use_graph=not args.no_graph, ddp_world=world if ddp else 1,
                                      global_cox=bool(ddp and args.global_cox), sync_bn=bool(ddp and args.sync_bn), **batch_of(f, k))

    def run(nsteps, nf):
        """exactly nsteps steps (one step = one batch of one fold model), dealt over nf groups of G lock-step models"""
        u, left = 0, nsteps
        while left > 0:
            m = min(G, left)
            launch(u, nf, m)
            u += 1; left -= m

    def timed(nsteps, nf):
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(nsteps, nf)
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        return D.max_over_ranks(time.perf_counter() - t0, dev)

    run(max(args.warmup, 3 * F * G), F)                           # includes each fold model's / group's graph capture
    if G > 1 and args.steps % (F * G):
        run(args.steps, F)                                        # the exact timed schedule once, untimed: captures the ragged tail's sub-group graph
    dt = timed(args.steps, F)
    if args.timed_only:
        if rank == 0:
            print(f"timed-only: {world * args.steps * B / dt:.1f} patients/s, {dt / args.steps * 1e3:.3f} ms/step", flush=True)
        D.barrier()
        return
    stats = engines[0].epoch_stats()
    sync_per_step = getattr(engines[0], "sync_collectives", 0) / max(1, args.steps + max(args.warmup, 3 * F * G)) if ddp and args.sync_bn else None
    if rank == 0:
        out = {
            "metric": "patients/sec per epoch (training: fwd + Cox + bwd + clip + Adam)",
            "value": world * args.steps * B / dt, "unit": "patients/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if world > 1 else None,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (("BASELINE config 4's per-rank problem (--mode ddp: ONE model, this rank's shard of the global batch per step)" if world == 1 else
                                     "BASELINE config 4's step (--mode ddp: ONE model data-parallel over %d ranks, global batch %d)" % (world, world * B))
                                    if one else "BASELINE config 2") +
                                   ": MultiModalSurvivalNet (DenseNet121-3D CT %dx%dx%d + RNA-seq 5005 + clinical), "
                                   "109 synthetic complete patients, 5-fold split, batch %d%s, Adam lr 1e-4 wd 1e-4, clip 1.0" % (dims + (B, " per rank" if one else "")),
                       "global_batch": world * B,
                       "parallelism": (f"ddp x{world} (gradient all-reduce per step in {args.ddp_buckets} bucket(s) overlapped with backward, "
                                       f"{'Sync' if args.sync_bn else 'local'} BN + {'global' if args.global_cox else 'local'} Cox risk set"
                                       + (f"; {sync_per_step:.0f} statistic all-reduces per step = 2 per BatchNorm layer and pass, the data-dependence floor" if sync_per_step else "") + ")" if ddp else
                                       "ddp x1: config 4's step on one rank without its collectives (N = 1: nothing to all-reduce)" if one else
                                       f"kfold-shard x{world} ranks x {F} concurrent groups x {G} lock-step fold models per GPU (one stream + step graph per group, no collective)"),
                       "concurrent_folds": F, "fold_group": G, "fold_models_in_flight": F * G,
                       "hip_graph": not args.no_graph, "mean_train_loss": stats["sum_loss"] / max(stats["n_batches"], 1)},
        }
        out["roofline"] = roofline_block(B, dims, dev, (G,))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_final(cohort_cpu, torch.as_tensor(folds[0][0]), args.cpu_steps, B)
        print(json.dumps(out), flush=True)
    D.barrier()


def _visible_gpus():
    """GPUs this process could use, counted WITHOUT initialising HIP in the launcher parent (its children are started with fork + exec):
    KFD topology nodes with SIMDs, capped by the *_VISIBLE_DEVICES lists.  0 when the box has no KFD (a CPU box)."""
    import re
    base, n = "/sys/class/kfd/kfd/topology/nodes", 0
    try:
        nodes = os.listdir(base)
    except OSError:
        nodes = []
    for d in nodes:
        try:
            with open(os.path.join(base, d, "properties")) as fh:
                m = re.search(r"^simd_count\s+(\d+)", fh.read(), re.M)
        except OSError:
            continue
        if m and int(m.group(1)) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip()]))
    return n


def _spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children through torch.distributed.run (one process per GPU,
    rendezvous on 127.0.0.1) and return their exit code.  Refuses when fewer than N GPUs are visible -- a --gpus N request never
    yields an n_gpus = 1 line."""
    n = args.gpus
    if not (args.rendezvous_check or args.dry_launch):
        ndev = _visible_gpus()
        if ndev < n and os.environ.get("MMS_ALLOW_SHARED_GPU") == "1" and os.environ.get("MMS_DIST_BACKEND") == "gloo":
            print(f"bench.py: REHEARSAL -- {n} ranks share {ndev} GPU(s) through gloo; the line is a liveness check, not a measurement", file=sys.stderr)
        elif ndev < n:
            print(f"bench.py: --gpus {n} but only {ndev} GPU(s) are visible; not running on fewer GPUs than asked", file=sys.stderr)
            return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if a != "--dry-launch"]
    if args.dry_launch:
        print(json.dumps({"launch": cmd}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="c3: lock-step steps (rounded up to whole epochs, default one epoch); "
                                                            "c2/c5: optimisation steps (default 240 / 200)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--volume", type=int, nargs=3, default=(64, 64, 32), metavar=("D", "H", "W"),
                    help="CT volume (default: the headline 64 64 32; BASELINE config 4 uses 128 128 64)")
    ap.add_argument("--workload", choices=["c3", "c2", "c5"], default="c3",
                    help="c3 (default): BASELINE config 3's N=1 leg, the metric's configuration; c2: BASELINE configs[1] (round 1's "
                         "headline; also hosts --mode ddp); c5: BASELINE config 5 (RNA-seq-only model, batch 2048, single GPU)")
    ap.add_argument("--patients", type=int, default=608)
    ap.add_argument("--folds", type=int, default=5)
    ap.add_argument("--lockstep-streams", type=int, default=3,
                    help="c3: the folds step as this many lock-step sub-groups on as many HIP streams (default 3: 5 folds = 2 + 2 + 1; round 2 ran 3 + 2); "
                         "more than 3 need GPU_MAX_HW_QUEUES >= streams + 1 in the environment, else two streams share a hardware queue")
    ap.add_argument("--validation-streams", type=int, default=2, help="c3: sub-groups / streams of the validation pass")
    ap.add_argument("--h2d", action="store_true", default=True, help="c3: also time the epoch with the cohort in pinned host memory")
    ap.add_argument("--no-h2d", dest="h2d", action="store_false")
    ap.add_argument("--many-folds", action="store_true", default=True, help="c3: also time 2 x 10 fold models in flight")
    ap.add_argument("--no-many-folds", dest="many_folds", action="store_false")
    ap.add_argument("--cpu-steps", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--timed-only", action="store_true", help="profiling aid: stop after the timed region (no extra legs, no JSON)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="profiling aid: run only the roofline leg (the conv2 family's isolated group launches), so that a "
                         "rocprofv3 --stats summary of this command holds exactly the launches the live measurement times")
    ap.add_argument("--concurrent-folds", type=int, default=2, help="c2: fold groups trained concurrently per GPU")
    ap.add_argument("--fold-group", type=int, default=10, help="c2: fold models advanced in lock-step by ONE launch sequence")
    ap.add_argument("--global-cox", action="store_true", default=True,
                    help="mode ddp (default): Cox risk set over the whole global batch (all-gather of hazards/times/events, gradients "
                         "summed) -- with rank-local BatchNorm the mode that can scale; --sync-bn is the exact-global-batch option")
    ap.add_argument("--local-cox", dest="global_cox", action="store_false", help="mode ddp: rank-local risk sets (gradients averaged)")
    ap.add_argument("--sync-bn", action="store_true", help="mode ddp: BatchNorm statistics over the global batch (SyncBN)")
    ap.add_argument("--ddp-buckets", type=int, default=6, help="mode ddp: gradient all-reduce buckets (reverse-layer order)")
    ap.add_argument("--mode", choices=["fold", "ddp"], default="fold",
                    help="c2, N>1: 'fold' = K-fold units sharded over ranks, no collective; 'ddp' = one model, global "
                         "batch N*B, bucketed gradient all-reduce (RCCL) per step")
    ap.add_argument("--dry-launch", action="store_true", help="--gpus N > 1 started without a launcher: print the rank-launch command as JSON and exit")
    ap.add_argument("--rendezvous-check", action="store_true",
                    help="ranks only rendezvous (torch.distributed init + one all-gather) and rank 0 prints {rendezvous: N, ranks: [...]}: "
                         "checks the self-launch path on a box without N GPUs (CPU: gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become the launcher.  Nothing in this process touches the GPU (the devices are counted from the KFD topology in
        # sysfs, not through HIP), so the ranks are plain children; this process only relays their output (rank 0's JSON line) and exit code.
        raise SystemExit(_spawn_ranks(args, sys.argv[1:]))

    from multimodal_survival_prediction_amd import distributed as D
    world, rank, local = D.init()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs")
    if args.rendezvous_check:
        import torch.distributed as tdist
        seen = [None] * world
        if world > 1:
            tdist.all_gather_object(seen, rank)
        else:
            seen = [0]
        if rank == 0:
            print(json.dumps({"rendezvous": world, "ranks": sorted(seen), "backend": tdist.get_backend() if world > 1 else None}), flush=True)
        D.barrier()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from multimodal_survival_prediction_amd import _build, _lib
    if not os.path.exists(_lib.lib_path()):
        _build.build()
    if _lib.load_library().mms_ablation_build() and not args.timed_only:
        raise SystemExit("bench.py: libmmsurv_hip.so is a timing-ablation build (MMS_CXXFLAGS=-DMMS_ABLATE_* / *_TIMING): only --timed-only "
                         "diagnostics may run on it, never a bench line")
    if args.mode == "ddp":
        args.workload = "c2"
    dflt = {"c3": (1, 1), "c2": (240, 20), "c5": (200, 10)}[args.workload]
    args.steps = dflt[0] if args.steps is None else args.steps
    args.warmup = dflt[1] if args.warmup is None else args.warmup
    if args.roofline_only:
        from multimodal_survival_prediction_amd.training import subgroup_sizes
        sub = (subgroup_sizes(args.folds, args.lockstep_streams) if args.workload == "c3"
               else (max(1, min(args.fold_group, 10)),))
        print(json.dumps(roofline_block(args.batch, tuple(args.volume), dev, sub, with_manifest=True)), flush=True)
        return
    if args.workload == "c5":
        if world > 1:
            raise SystemExit("--workload c5 is a single-GPU run")
        run_config5(args, dev)
    elif args.workload == "c2":
        run_config2(args, world, rank, dev)
    else:
        run_config3(args, world, rank, dev)


if __name__ == "__main__":
    main()
