#!/usr/bin/env python3
"""Headline benchmark: patients/sec of the multimodal survival training hot path on MI355X.

Workload (BASELINE.json configs[1]): full MultiModalSurvivalNet (DenseNet121-3D CT 64x64x32 + RNA-seq 5005 +
clinical), 109 synthetic complete-modality patients, batch 4, 5-fold split, Adam(lr 1e-4, wd 1e-4),
clip_grad_norm_(1.0).  A "step" is one pass of the hot path over one batch of 4 patients of one fold's model:
zero-grad, forward, Cox partial likelihood, backward, clip, Adam -- all inside one replayed HIP graph.  The cohort is
resident in HBM before the timed region; per step the batch is gathered device-to-device into the graph's static buffers.

K-fold cross-validation trains independent models of one shape, and at batch 4 one model's step is a chain of ~560 small
dependent kernels that leaves most of the 256 CUs idle.  Fold models are therefore advanced in lock-step as FOLD GROUPS
(DESIGN.md section 3a): every launch of the step carries the parameter blocks of all G models of a group, and F groups run
concurrently on F streams (one step graph each).  Default: F = 2 groups x G = 10 models = 20 fold models in flight (four
5-fold cross-validations; the reference's own experiment set is 3-5 scripts x 5 folds), the K timed steps dealt over
them.  Per-model semantics are untouched (tests/test_gpu_fold_group.py).  The same run also reports, in `config`, one
5-fold CV alone on the GPU (5 lock-step models, `one_cv_5_lockstep_patients_per_s`) and ONE model alone
(`single_chain_patients_per_s`).

  python bench.py [--gpus N] [--steps K] [--warmup W]            (N>1: launched by torch.distributed.run)
N>1 shards K-fold units over ranks (fold k -> rank k mod N, no data-path collective): weak scaling.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     dominant kernel, algorithmic FLOPs / measured average launch duration vs the fp32 MFMA peak
  "cpu_baseline": the CPU oracle (torch fp32 restatement of the same model/loop) timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
BLOCKS = ((6, 64), (12, 128), (24, 256), (16, 512))     # (layers, first-layer input channels) of DenseNet121


def measure_dominant_kernel(B, dims, device, G=1, reps=20):
    """Average launch duration of the dominant kernel -- the weight gradient of the dense-layer 3x3x3 conv
    (mms_conv3_bwd_weight_group: conv3_bwdw_mt_kernel for the block-1 launches of groups, tile_gemm_kernel<Conv3BwdWOp>
    otherwise; 58 launches per group step, the largest share of GPU time in profiles/r01_*) -- timed live
    with HIP events on the launch stream (torch's current stream), launched exactly as the timed region launches it:
    one launch carries the G models of a fold group (mms_conv3_bwd_weight_group), shape by shape with the driver's own
    split factors, weighted by the launch counts.  Algorithmic FLOPs per launch = G * 2 * M * 27 * 128 * 32."""
    import ctypes
    from multimodal_survival_prediction_amd import _lib, ops
    lib, S = _lib.load_library(), _lib.structs()
    tot_t, tot_f, n = 0.0, 0.0, 0
    D, H, W = dims
    g, b = torch.ones(128, device=device), torch.zeros(128, device=device)
    for i, (layers, _) in enumerate(BLOCKS):
        gd = (D // 4 >> i, H // 4 >> i, W // 4 >> i)
        M = B * gd[0] * gd[1] * gd[2]
        rows, rows_s = (1024, 256) if G >= 4 else (512, 128)                     # dn_net.hip: ms3
        fills = lambda w: w * 10 >= -(-w // 768) * 768 * 9                     # dn_ops.h: mms_conv3w_mt_fills
        if G >= 4 and M > 1024 and not fills(-(-M // 1024) * G * 9) and fills(-(-M // 512) * G * 9):
            rows = 512                                                           # ... e.g. 5 models: multi-tap kernel on 512-row chunks
        ms = (M + rows - 1) // rows if M > 1024 else max((M + rows_s - 1) // rows_s, 1)
        coords = ops.init_coords(B, gd, device)
        keep, blocks = [], []
        for _ in range(G):
            y1 = torch.randn(M, 128, device=device)
            s, q = y1.double().sum(0), (y1.double() ** 2).sum(0)
            bn = ops.bnsrc(g, b, M, True, s, q)
            dslab = torch.randn(M, 256, device=device)
            dwp = torch.zeros(27 * 32 * 128, device=device)
            dz = dslab[:, 64:96]
            keep.append((y1, s, q, dslab, dwp))
            blocks.append(S["Conv3BwdWP"](y1.data_ptr(), coords.data_ptr(), ops.dims3(gd), M, bn, dz.data_ptr(), dz.stride(0),
                                          dwp.data_ptr(), ms, 1))
        arr = (S["Conv3BwdWP"] * G)(*blocks)

        def launch():
            _lib.check(lib.mms_conv3_bwd_weight_group(arr, G, ops.stream()), "mms_conv3_bwd_weight_group")
        for _ in range(3):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            launch()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / reps
        tot_t += t * layers
        tot_f += G * 2.0 * M * 27 * 128 * 32 * layers
        n += layers
    return tot_t / n, tot_f / n


def cpu_baseline(cohort, train_idx, steps, B):
    """The CPU oracle (same model, same loop body) on this box's host cores: bounded sample of `steps` batches."""
    from oracle import losses as OL
    from oracle import models as OM
    torch.manual_seed(0)
    model = OM.MultiModalSurvivalNet(rna_dim=cohort["rnaseq"].shape[1], use_monai=True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4)
    model.train()
    cores = torch.get_num_threads()

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
    return dict(value=steps * B / dt, unit="patients/s", cores=cores, kind="port",
                sample=f"{steps} training steps (batch {B}) of the torch-fp32 CPU oracle after 1 warm-up step, {dt:.1f} s")


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
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak",
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
        out["cpu_baseline"] = dict(value=args.cpu_steps * B / cdt, unit="patients/s", cores=torch.get_num_threads(), kind="port",
                                   sample=f"{args.cpu_steps} training steps (batch {B}) of the torch-fp32 CPU oracle after 1 warm-up step, {cdt:.1f} s")
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--volume", type=int, nargs=3, default=(64, 64, 32), metavar=("D", "H", "W"),
                    help="CT volume (default: the headline 64 64 32; BASELINE config 4 uses 128 128 64)")
    ap.add_argument("--workload", choices=["c2", "c5"], default="c2",
                    help="c2 (default): BASELINE configs[1], the headline; c5: BASELINE config 5 (RNA-seq-only model, batch 2048, single GPU)")
    ap.add_argument("--cpu-steps", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--timed-only", action="store_true", help="profiling aid: stop after the timed region (no extra legs, no JSON)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="profiling aid: run only the roofline leg (the dominant kernel's isolated group launches), so that a "
                         "rocprofv3 --stats summary of this command holds exactly the launches the live measurement times")
    ap.add_argument("--concurrent-folds", type=int, default=2,
                    help="fold groups trained concurrently per GPU, one HIP stream + step graph each (mode fold)")
    ap.add_argument("--fold-group", type=int, default=10,
                    help="fold models advanced in lock-step by ONE launch sequence (FoldGroupEngine, *_group entry points); "
                         "--concurrent-folds then counts concurrent groups")
    ap.add_argument("--global-cox", action="store_true",
                    help="mode ddp: Cox risk set over the whole global batch (all-gather of hazards/times/events, gradients "
                         "summed) instead of rank-local risk sets")
    ap.add_argument("--mode", choices=["fold", "ddp"], default="fold",
                    help="N>1: 'fold' = K-fold units sharded over ranks, no collective (default); 'ddp' = one model, global "
                         "batch N*B, flat gradient all-reduce (RCCL) per step")
    args = ap.parse_args()

    from multimodal_survival_prediction_amd import distributed as D
    world, rank, local = D.init()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from multimodal_survival_prediction_amd import _build, _lib
    if not os.path.exists(_lib.lib_path()):
        _build.build()
    if args.workload == "c5":
        if world > 1:
            raise SystemExit("--workload c5 is a single-GPU run")
        if args.steps == 240 and args.warmup == 20:
            args.steps, args.warmup = 200, 10
        run_config5(args, dev)
        return
    from multimodal_survival_prediction_amd import data, models
    from multimodal_survival_prediction_amd.training import FusedOptimizer

    B, dims, rna_dim = args.batch, tuple(args.volume), 5005
    if args.roofline_only:
        Gr = max(1, min(args.fold_group, 10))
        avg_t, avg_f = measure_dominant_kernel(B, dims, dev, Gr)
        print(json.dumps({"kernel": "mms_conv3_bwd_weight_group (conv3_bwdw_mt_kernel / tile_gemm_kernel<Conv3BwdWOp>)", "models_per_launch": Gr, "avg_launch_us": avg_t * 1e6,
                          "avg_flops_per_launch": avg_f, "achieved_tflops": avg_f / avg_t / 1e12}), flush=True)
        return
    cohort_cpu = data.make_cohort(n=109, dims=dims, rna_dim=rna_dim, seed=608, complete=True)
    cohort = data.cohort_to(cohort_cpu, dev)                      # resident in HBM before the timed region
    folds = data.kfold_indices(cohort["n"], 5, seed=42)
    ddp = args.mode == "ddp" and world > 1
    F = 1 if ddp else max(1, args.concurrent_folds)
    G = 1 if ddp else max(1, min(args.fold_group, 10))
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
        streams.append(torch.cuda.Stream(device=dev))
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
                                      global_cox=bool(ddp and args.global_cox), **batch_of(f, k))

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
    # single chain: ONE fold model alone on the GPU (no grouping, no concurrency), same graph-replayed step
    if F * G > 1:
        n1 = max(args.steps // (3 * F * G), 10)
        if G > 1:
            def one(i):
                nb0 = len(orders[0]) // B
                groups[0].train_step_indexed(cohort, orders[0][(i % nb0) * B:][:B].view(1, B), members=(0,),
                                             skip_if_unusable=True, use_graph=not args.no_graph)
            for i in range(3):
                one(i)
            torch.cuda.synchronize(); D.barrier()
            t0 = time.perf_counter()
            for i in range(n1):
                one(i)
            torch.cuda.synchronize(); D.barrier()
            dt1 = D.max_over_ranks(time.perf_counter() - t0, dev) / n1
        else:
            dt1 = timed(n1, 1) / n1
    else:
        dt1 = dt / args.steps
    # one 5-fold cross-validation alone on the GPU: its 5 models and nothing else (members 0-4 of group 0, stepped as two
    # concurrent lock-step sub-groups of 3 + 2)
    dt5 = None
    if G >= 5:
        cv5_split = os.environ.get("MMS_CV5_SPLIT", "1") == "1" and F >= 2      # sub-groups (0,1,2) and (3,4) on two streams (+4.5 % over one group of 5)

        def cv5(i):
            idx = [orders[g][(i % (len(orders[g]) // B)) * B:][:B] for g in range(5)]
            if not cv5_split:
                groups[0].train_step_indexed(cohort, torch.stack(idx), members=(0, 1, 2, 3, 4), skip_if_unusable=True,
                                             use_graph=not args.no_graph)
                return
            for sidx, mem in ((0, (0, 1, 2)), (1 % F, (3, 4))):
                with torch.cuda.stream(streams[sidx]):
                    groups[0].train_step_indexed(cohort, torch.stack([idx[m] for m in mem]), members=mem, skip_if_unusable=True,
                                                 use_graph=not args.no_graph)
        n5 = max(args.steps // (2 * F * G), 6)
        for i in range(3):
            cv5(i)
        torch.cuda.synchronize(); D.barrier()
        t0 = time.perf_counter()
        for i in range(n5):
            cv5(i)
        torch.cuda.synchronize(); D.barrier()
        dt5 = D.max_over_ranks(time.perf_counter() - t0, dev) / (5 * n5)
    # validation (reported separately): eval-mode forwards of the same models, same grouping/concurrency
    val_rate = None
    if G > 1:
        for e_ in engines:
            e_.model.eval()
        nv = max(args.steps // (F * G), 4)

        def vrun(n):
            for u in range(n):
                f = u % F
                with torch.cuda.stream(streams[f]):
                    bs = [batch_of(f * G + g, u // F) for g in range(G)]
                    groups[f].forward_eval([dict(ct=b["ct"], rna=b["rna"], clinical=b["clinical"]) for b in bs])
        vrun(2 * F)
        torch.cuda.synchronize(); D.barrier()
        t0 = time.perf_counter()
        vrun(nv)
        torch.cuda.synchronize(); D.barrier()
        val_rate = world * nv * G * B / D.max_over_ranks(time.perf_counter() - t0, dev)
        for e_ in engines:
            e_.model.train()
    stats = engines[0].epoch_stats()

    if rank == 0:
        out = {
            "metric": "patients/sec per epoch (training: fwd + Cox + bwd + clip + Adam)",
            "value": world * args.steps * B / dt, "unit": "patients/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "MultiModalSurvivalNet (DenseNet121-3D CT %dx%dx%d + RNA-seq 5005 + clinical), "
                                   "109 synthetic complete patients, 5-fold split, batch %d, Adam lr 1e-4 wd 1e-4, clip 1.0" % (dims + (B,)),
                       "global_batch": world * B, "parallelism": (f"ddp x{world} (flat gradient all-reduce per step, local BN + {'global' if args.global_cox else 'local'} Cox risk set)" if ddp else
                                       f"kfold-shard x{world} ranks x {F} concurrent groups x {G} lock-step fold models per GPU (one stream + step graph per group, no collective)"),
                       "concurrent_folds": F, "fold_group": G, "fold_models_in_flight": F * G,
                       "one_cv_5_lockstep_patients_per_s": (world * B / dt5) if dt5 else None,
                       "validation_patients_per_s": val_rate, "single_chain_patients_per_s": world * B / dt1, "single_chain_ms_per_step": dt1 * 1e3,
                       "hip_graph": not args.no_graph, "mean_train_loss": stats["sum_loss"] / max(stats["n_batches"], 1)},
        }
        avg_t, avg_f = measure_dominant_kernel(B, dims, dev, G)
        traffic = None      # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside bench.py)
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_conv3bwdw_traffic.json")) as f:
                traffic = json.load(f)["avg_hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        out["roofline"] = {"bound": "mfma", "kernel": f"mms_conv3_bwd_weight_group = conv3_bwdw_mt_kernel (block-1 launches of well-filled groups) / tile_gemm_kernel<Conv3BwdWOp> "
                                                       f"(weight gradient of the dense-layer 3x3x3 conv; 58 launches per group step, {G} fold models per launch)",
                           "achieved": avg_f / avg_t / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": avg_f / avg_t / 1e12 / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                           "avg_launch_us": avg_t * 1e6, "avg_flops_per_launch": avg_f}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cohort_cpu, torch.as_tensor(folds[0][0]), args.cpu_steps, B)
        print(json.dumps(out), flush=True)
    D.barrier()


if __name__ == "__main__":
    main()
