"""Microbench of conv0 backward-weight for a fold group of G models (event-timed)."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_survival_prediction_amd import ops, _lib
dev = "cuda:0"
G = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nrep = int(sys.argv[3]) if len(sys.argv) > 3 else 8      # Conv0BwdWP.nrep: gradient replicas (0: atomics straight into dw)
B, (D, H, W) = 4, (64, 64, 32)
g0 = (D // 2, H // 2, W // 2)
M = B * g0[0] * g0[1] * g0[2]
lib, S = _lib.load_library(), _lib.structs()
keep, blocks = [], []
coords = ops.init_coords(B, g0, dev)
for g in range(G):
    x = torch.randn(B, D, H, W, device=dev)
    y0 = torch.randn(M, 64, device=dev); dbn = torch.randn(M, 64, device=dev)
    s, q = y0.double().sum(0), (y0.double() ** 2).sum(0)
    bn = ops.bnsrc(torch.ones(64, device=dev), torch.zeros(64, device=dev), M, True, s, q)
    s1, s2 = dbn.double().sum(0), (dbn.double() * y0.double()).sum(0)
    bb = ops.bnbwd(s1, s2)
    dw = torch.zeros(64 * 343, device=dev); dg = torch.zeros(64, device=dev); db = torch.zeros(64, device=dev)
    rep = torch.zeros(max(nrep, 1), 64 * 343, device=dev)
    keep.append((x, y0, dbn, s, q, s1, s2, dw, dg, db, rep))
    blocks.append(S["Conv0BwdWP"](dbn.data_ptr(), y0.data_ptr(), bn, bb, x.data_ptr(), ops.dims3((D, H, W)), ops.dims3(g0),
                                  coords.data_ptr(), M, dw.data_ptr(), 64, dg.data_ptr(), db.data_ptr(), rep.data_ptr() if nrep else None, nrep))
arr = (S["Conv0BwdWP"] * G)(*blocks)
def launch():
    _lib.check(lib.mms_conv0_bwd_weight_group(arr, G, None, ops.stream()), "conv0bw")
for _ in range(3): launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): launch()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e3 / reps
print(f"G={G} conv0 bwd-weight avg {t:.1f} us  ({G * 2.0 * M * 343 * 64 / t / 1e6:.1f} TFLOP/s)")
