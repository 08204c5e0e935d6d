"""Per-phase wall-clock sums of the persistent per-block forward kernels (csrc/dn_cl.hip; build with MMS_CXXFLAGS=-DB4_TIMING).  GPU box.
usage: python tools/b4_timing.py        (block 4: one cluster of 16 rows; block 3: four clusters of 32 rows, cluster 0 is reported)"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from multimodal_survival_prediction_amd import _lib
from multimodal_survival_prediction_amd.densenet import DenseNet121
lib = _lib.load_library()
B, dims = 4, (64, 64, 32)
net = DenseNet121().to("cuda").train()
net.dn_opts = dict(persist_b3=1)
x = torch.rand(B, 1, *dims, device="cuda")
for _ in range(3):
    y = net(x)
torch.cuda.synchronize()
w = net.workspace_region("b4_err", 0, torch.int32).cpu().tolist()
names = ["loop top", "as build", "conv1+stats(+exchange)+publish", "sweep A", "conv2 + publish", "sweep B", "prefetch issue + z + stats(+exchange)"]
for blk, off, nl in (("block 4 (RT = 1)", 8, 16), ("block 3 (RT = 2)", 72, 24)):
    print(blk)
    print("%-40s " % "us per layer, workgroup:" + " ".join("%6d" % i for i in range(8)))
    for i, n in enumerate(names):
        print("%-40s " % n + " ".join("%6.2f" % (w[off + 8 * g + i] / 100.0 / nl) for g in range(8)))
    print("%-40s " % "sum" + " ".join("%6.2f" % (sum(w[off + 8 * g:off + 8 * g + 7]) / 100.0 / nl) for g in range(8)))
print("err word", w[0])
