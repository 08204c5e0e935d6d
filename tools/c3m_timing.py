"""In-kernel phase times of the multi-tap conv2 forward kernel (conv3_fwd_mt_kernel, dn_fwd.hip) at the block-1 shape, GPU box.
Needs a library built with -DC3M_TIMING in place of libmmsurv_hip.so (tools/build_variant.sh c3m_timing "-DC3M_TIMING" dn_fwd.hip, copied over
the library in the box's scratch tree).  Launches rotate over 8 operand sets (cold L2, as in the step).  usage: c3m_timing.py [G]"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multimodal_survival_prediction_amd import ops, _lib
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev, nsets = "cuda:0", 8
B, gd = 4, (16, 16, 8)
M = B * gd[0] * gd[1] * gd[2]
lib, S = _lib.load_library(), _lib.structs()
raw = ctypes.CDLL(_lib.lib_path())
buf = torch.zeros(8 * 4096, dtype=torch.int64, device=dev)
assert raw.mms_c3m_timing_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
coords = ops.init_coords(B, gd, dev)
keep, arrs = [], []
for k in range(nsets):
    blocks = []
    for g in range(G):
        y1 = torch.randn(M, 128, device=dev); wp = torch.randn(32 * 27 * 128, device=dev) * 0.02
        s, q = y1.double().sum(0), (y1.double() ** 2).sum(0)
        bn = ops.bnsrc(torch.ones(128, device=dev), torch.zeros(128, device=dev), M, True, s, q)
        slab = torch.zeros(M, 256, device=dev)
        os_, oq = torch.zeros(32, dtype=torch.float64, device=dev), torch.zeros(32, dtype=torch.float64, device=dev)
        keep.append((y1, wp, s, q, slab, os_, oq))
        blocks.append(S["Conv3FwdP"](y1.data_ptr(), coords.data_ptr(), ops.dims3(gd), M, wp.data_ptr(), slab[:, 64:96].data_ptr(), 256, bn,
                                     os_.data_ptr(), oq.data_ptr(), None, 27))
    arrs.append((S["Conv3FwdP"] * G)(*blocks))
for i in range(12):
    _lib.check(lib.mms_conv3_fwd_group(arrs[i % nsets], G, None, ops.stream()), "conv3_fwd_group")
torch.cuda.synchronize()
n = (M // 32) * G
t = buf[:8 * n].view(n, 8).cpu().numpy().astype("float64")
d = t[:, 1:8] - t[:, 0:7]
names = ["prologue loads issued", "first window staged + barrier", "taps 0-9 (+ window 2)", "taps 10-17", "taps 18-26 (+ window 3)", "channel splits added", "tile + statistics written"]
print("conv3_fwd_mt_kernel<32>, block-1 shape, G = %d: %d workgroups, hundreds of shader-clock cycles (s_memtime; mean / max over workgroups)" % (G, n))
for i, nm in enumerate(names):
    print("  %-34s %7.2f / %7.2f" % (nm, d[:, i].mean() / 100, d[:, i].max() / 100))
print("  %-34s %7.2f / %7.2f" % ("total", (t[:, 7] - t[:, 0]).mean() / 100, (t[:, 7] - t[:, 0]).max() / 100))
