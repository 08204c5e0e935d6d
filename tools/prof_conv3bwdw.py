"""Microbench of the dominant kernel (conv3 weight gradient) at one DenseNet block shape, launched as the training step
launches it (one launch = the G models of a fold group), for rocprofv3 --pmc passes.
usage: prof_conv3bwdw.py <block 0..3> <reps> [G=5] [rows per chunk] [mt: -1 = one-tap GEMM form, 2 = multi-tap kernel, 0 = by launch size]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_survival_prediction_amd import ops, _lib
dev = "cuda:0"
B, (D, H, W) = 4, (64, 64, 32)
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
G = int(sys.argv[3]) if len(sys.argv) > 3 else 5
lib, S = _lib.load_library(), _lib.structs()
gd = (D // 4 >> which, H // 4 >> which, W // 4 >> which)
M = B * gd[0] * gd[1] * gd[2]
rows, rows_s = (1024, 256) if G >= 4 else (512, 128)            # dn_net.hip: ms3
fills = lambda w: w * 10 >= -(-w // 768) * 768 * 9                     # dn_ops.h: mms_conv3w_mt_fills
if G >= 4 and M > 1024 and not fills(-(-M // 1024) * G * 9) and fills(-(-M // 512) * G * 9):
    rows = 512
if len(sys.argv) > 4:
    rows = rows_s = int(sys.argv[4])
ms = (M + rows - 1) // rows if M > 1024 else max((M + rows_s - 1) // rows_s, 1)
g, b = torch.ones(128, device=dev), torch.zeros(128, device=dev)
coords = ops.init_coords(B, gd, dev)
keep, blocks = [], []
for _ in range(G):
    y1 = torch.randn(M, 128, device=dev)
    s, q = y1.double().sum(0), (y1.double() ** 2).sum(0)
    bn = ops.bnsrc(g, b, M, True, s, q)
    dslab = torch.randn(M, 256, device=dev)
    dwp = torch.zeros(27 * 32 * 128, device=dev)
    dz = dslab[:, 64:96]
    keep.append((y1, s, q, dslab, dwp))
    blocks.append(S["Conv3BwdWP"](y1.data_ptr(), coords.data_ptr(), ops.dims3(gd), M, bn, dz.data_ptr(), dz.stride(0), dwp.data_ptr(), ms, 1))
arr = (S["Conv3BwdWP"] * G)(*blocks)
import ctypes
opt = ops.dn_opts(conv3w_mt=int(sys.argv[5]) if len(sys.argv) > 5 else 0)
for _ in range(3):
    _lib.check(lib.mms_conv3_bwd_weight_group(arr, G, ctypes.byref(opt), ops.stream()), "conv3_bwd_weight_group")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    _lib.check(lib.mms_conv3_bwd_weight_group(arr, G, ctypes.byref(opt), ops.stream()), "conv3_bwd_weight_group")
e1.record()
torch.cuda.synchronize()
print("us per launch %.1f" % (e0.elapsed_time(e1) * 1e3 / reps))
print("block", which, "M", M, "msplit", ms, "G", G, "algorithmic bytes per launch", G * (M * 128 * 4 + M * 32 * 4 + 27 * 32 * 128 * 4))
