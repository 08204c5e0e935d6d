"""Per-phase wall-clock sums of the block-4 persistent forward (build with MMS_CXXFLAGS=-DB4_TIMING).  GPU box."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from multimodal_survival_prediction_amd import _lib
from multimodal_survival_prediction_amd.densenet import DenseNet121
from multimodal_survival_prediction_amd.engine import engine_of
lib = _lib.load_library()
B, dims = 4, (64, 64, 32)
net = DenseNet121().to("cuda").train()
x = torch.rand(B, 1, *dims, device="cuda")
for _ in range(3):
    y = net(x)
torch.cuda.synchronize()
w = net.workspace_region("b4_err", 0, torch.int32).cpu().tolist()
names = ["loop top", "as build", "conv1+stats+publish", "sync A (+w1 prefetch issue)", "gather a2 + conv2 + publish", "sync B (+tap prefetch issue)", "gather z + stats"]
print("%-32s " % "us per layer, workgroup:" + " ".join("%6d" % i for i in range(8)))
for i, n in enumerate(names):
    print("%-32s " % n + " ".join("%6.2f" % (w[8 + 8 * g + i] / 100.0 / 16) for g in range(8)))
print("%-32s " % "sum" + " ".join("%6.2f" % (sum(w[8 + 8 * g:8 + 8 * g + 7]) / 100.0 / 16) for g in range(8)))
print("err word", w[0])
