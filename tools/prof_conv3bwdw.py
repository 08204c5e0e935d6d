"""Microbench of the dominant kernel (conv3 weight gradient) at one DenseNet block shape, for rocprofv3 --pmc passes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_survival_prediction_amd import ops
dev = "cuda:0"
B, (D, H, W) = 4, (64, 64, 32)
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
gd = (D // 4 >> which, H // 4 >> which, W // 4 >> which)
M = B * gd[0] * gd[1] * gd[2]
ms = (M + 511) // 512 if M > 1024 else max((M + 127) // 128, 1)
g, b = torch.ones(128, device=dev), torch.zeros(128, device=dev)
y1 = torch.randn(M, 128, device=dev)
s, q = y1.double().sum(0), (y1.double() ** 2).sum(0)
bn = ops.bnsrc(g, b, M, True, s, q)
coords = ops.init_coords(B, gd, dev)
dslab = torch.randn(M, 256, device=dev)
dwp = torch.zeros(27 * 32 * 128, device=dev)
for _ in range(reps):
    ops.conv3_bwd_weight(y1, coords, gd, bn, dslab[:, 64:96], dwp, ms, tapmajor=True)
torch.cuda.synchronize()
print("block", which, "M", M, "msplit", ms, "algorithmic bytes", M * 128 * 4 + M * 32 * 4 + 27 * 32 * 128 * 4)
