"""Soak test of the fence-free split-K fixup (Conv1FwdOp KSPLIT): many launches, NaN-poisoned partial buffer, several
concurrent streams, every result compared with the unsplit kernel.  A lost or stale partial shows up as NaN / mismatch."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_survival_prediction_amd import ops
dev = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
torch.manual_seed(0)
cases = []
for M, K, ks in ((128, 640, 5), (16, 992, 8), (128, 1024, 8), (64, 384, 3)):
    slab = torch.randn(M, 1024, device=dev) * 1.5 + 0.3
    g, b = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
    w = torch.randn(128, K, device=dev) / K ** 0.5
    s, q = slab[:, :K].double().sum(0).contiguous(), (slab[:, :K].double() ** 2).sum(0).contiguous()
    bn = ops.bnsrc(g, b, M, True, s, q)
    y0 = torch.zeros(M, 128, device=dev)
    ops.conv1_fwd(slab, K, w, y0, bn, M)
    cases.append(dict(M=M, K=K, ks=ks, slab=slab, w=w, bn=bn, y0=y0, keep=(g, b, s, q),
                      partial=torch.empty(ks * M * 128, device=dev), counters=torch.zeros(64, dtype=torch.int32, device=dev),
                      y=torch.zeros(M, 128, device=dev), stream=torch.cuda.Stream()))
torch.cuda.synchronize()
bad = 0
for it in range(iters):
    for c in cases:                      # the four cases run concurrently on four streams
        with torch.cuda.stream(c["stream"]):
            c["partial"].fill_(float("nan")); c["y"].zero_()
            ops.conv1_fwd(c["slab"], c["K"], c["w"], c["y"], c["bn"], c["M"], partial=c["partial"], ksplit=c["ks"], counters=c["counters"])
    if it % 50 == 49 or it == iters - 1:
        torch.cuda.synchronize()
        for c in cases:
            err = float((c["y"] - c["y0"]).abs().max())
            if not (err <= 2e-5 * float(c["y0"].abs().max())) or int(c["counters"].abs().sum()) != 0:
                bad += 1
                print("MISMATCH it", it, "case", c["M"], c["K"], c["ks"], "err", err, flush=True)
print("soak done:", iters, "iterations x", len(cases), "cases, mismatches:", bad)
sys.exit(1 if bad else 0)
