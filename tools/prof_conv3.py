"""Microbench of the conv3 forward kernel at the four DenseNet block shapes (for rocprofv3 --pmc runs)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_survival_prediction_amd import ops
dev = "cuda:0"
B, dims = 4, (64, 64, 32)
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
D, H, W = dims
gd = (D // 4 >> which, H // 4 >> which, W // 4 >> which)
M = B * gd[0] * gd[1] * gd[2]
wp = torch.randn(32 * 27 * 128, device=dev) * 0.02
g, b = torch.ones(128, device=dev), torch.zeros(128, device=dev)
y1 = torch.randn(M, 128, device=dev)
s, q = y1.double().sum(0), (y1.double() ** 2).sum(0)
bn = ops.bnsrc(g, b, M, True, s, q)
coords = ops.init_coords(B, gd, dev)
slab = torch.zeros(M, 256, device=dev)
os_, oq = torch.zeros(32, dtype=torch.float64, device=dev), torch.zeros(32, dtype=torch.float64, device=dev)
for _ in range(reps):
    ops.conv3_fwd(y1, coords, gd, wp, slab[:, 64:96], bn, os_, oq)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.conv3_fwd(y1, coords, gd, wp, slab[:, 64:96], bn, os_, oq)
e1.record(); torch.cuda.synchronize()
print("block", which, "M", M, "avg us", e0.elapsed_time(e1) * 1e3 / reps)
