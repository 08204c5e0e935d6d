"""Microbench of the conv2 backward-data launch at a DenseNet block shape for a fold group of G models (event-timed).
usage: prof_conv3bwdd_group.py <block> <G> [<sets> [<conv3_mt>]]   sets > 1: launches rotate over that many operand sets (cold L2, as in the
step); conv3_mt: MmsDnOpts.conv3_mt (-1 = the tile-GEMM form, 0 = default choice)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_survival_prediction_amd import ops, _lib
dev = "cuda:0"
blk = int(sys.argv[1]) if len(sys.argv) > 1 else 0
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nsets = int(sys.argv[3]) if len(sys.argv) > 3 else 1
o = ops.dn_opts(conv3_mt=int(sys.argv[4]) if len(sys.argv) > 4 else 0)
reps = 20
B, (D, H, W) = 4, (64, 64, 32)
gd = (D // 4 >> blk, H // 4 >> blk, W // 4 >> blk)
M = B * gd[0] * gd[1] * gd[2]
lib, S = _lib.load_library(), _lib.structs()
coords = ops.init_coords(B, gd, dev)
keep, arrs = [], []
for k in range(nsets):
    blocks = []
    for g in range(G):
        y1 = torch.randn(M, 128, device=dev)
        wpb = torch.randn(128 * 27 * 32, device=dev) * 0.02
        s, q = y1.double().sum(0), (y1.double() ** 2).sum(0)
        bn = ops.bnsrc(torch.ones(128, device=dev), torch.zeros(128, device=dev), M, True, s, q)
        dslab = torch.randn(M, 256, device=dev)
        dbn = torch.zeros(M, 128, device=dev)
        bst = torch.zeros(2, 128, dtype=torch.float64, device=dev)
        dz = dslab[:, 64:96]
        keep.append((y1, wpb, s, q, dslab, dbn, bst))
        blocks.append(S["Conv3BwdDataP"](dz.data_ptr(), dz.stride(0), coords.data_ptr(), ops.dims3(gd), M, wpb.data_ptr(), y1.data_ptr(), bn,
                                         dbn.data_ptr(), bst[0].data_ptr(), bst[1].data_ptr(), None, 1, 1, 2 * 128, 0))
    arrs.append((S["Conv3BwdDataP"] * G)(*blocks))
def launch(i):
    _lib.check(lib.mms_conv3_bwd_data_group(arrs[i % nsets], G, ops.opts_ref(o), ops.stream()), "conv3_bwd_data_group")
for i in range(3): launch(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(reps): launch(i)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e3 / reps
print(f"block {blk + 1} M={M} G={G} sets={nsets} conv3_mt={o.conv3_mt}: conv3 bwd-data avg {t:.1f} us ({G * 2.0 * M * 27 * 128 * 32 / t / 1e6:.1f} TFLOP/s)")
