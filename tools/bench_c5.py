"""BASELINE config 5: RNA-seq-only model (5005 -> 1024 -> 512 -> 256 -> 1), batch 2048, one full training step per replay
(zero-grad, forward, O(B^2) Cox partial likelihood, backward, AdamW) as a captured HIP graph.  Prints one JSON line;
--cpu-steps N also times the CPU oracle's loop body (train_rnaseq_only.py:153-176) on the host cores."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--cpu-steps", type=int, default=0)
    a = ap.parse_args()
    from multimodal_survival_prediction_amd import models
    from multimodal_survival_prediction_amd.training import FusedOptimizer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = models.RNASeqSurvivalModel(input_dim=5005).to(dev).train()
    fo = FusedOptimizer(net, lr=1e-4, weight_decay=1e-3, adamw=True, max_norm=0.0)
    B = a.batch
    rng = np.random.default_rng(0)
    rna = torch.tensor(rng.normal(0, 1, (B, 5005)).astype(np.float32), device=dev)
    t = torch.tensor((rng.exponential(1000, B) + 1 + np.arange(B) * 1e-3).astype(np.float32), device=dev)
    e = torch.tensor((rng.random(B) < 0.6).astype(np.float32), device=dev)
    for _ in range(a.warmup):
        fo.engine.train_step(None, rna, time=t, event=e, skip_if_unusable=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fo.engine.train_step(None, rna, time=t, event=e, skip_if_unusable=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    flops = 3 * 2 * B * (5005 * 1024 + 1024 * 512 + 512 * 256 + 256) - 2 * B * 5005 * 1024      # no dX for the first layer
    out = dict(workload="C5 rnaseq-only B=%d" % B, ms_per_step=dt * 1e3, patients_per_s=B / dt, gemm_tflops=flops / dt / 1e12,
               loss=fo.engine.epoch_stats()["sum_loss"] / (a.steps + a.warmup))
    if a.cpu_steps:
        from oracle import losses as OL, models as OM
        ref = OM.RNASeqSurvivalModel(input_dim=5005).train()
        opt = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=1e-3)
        x, tc, ec = rna.cpu(), t.cpu(), e.cpu().bool()
        for i in range(a.cpu_steps + 1):
            if i == 1:
                c0 = time.perf_counter()
            opt.zero_grad(); OL.neg_partial_log_likelihood(ref(x).squeeze(), ec, tc).backward(); opt.step()
        cdt = (time.perf_counter() - c0) / a.cpu_steps
        out["cpu_patients_per_s"], out["cpu_threads"] = B / cdt, torch.get_num_threads()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
