"""Microbench of conv3 forward at a DenseNet block shape for a fold group of G models (event-timed).
usage: prof_conv3fwd_group.py <block> <G> [<sets>]   sets > 1: the launches rotate over that many distinct (activations, weights) sets, so that
every launch meets its operands as cold in L2 as the step's launches do (default 1: every launch re-reads the same, L2-warm operands)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_survival_prediction_amd import ops, _lib
dev = "cuda:0"
blk = int(sys.argv[1]) if len(sys.argv) > 1 else 0
G = int(sys.argv[2]) if len(sys.argv) > 2 else 10
reps = 20
nsets = int(sys.argv[3]) if len(sys.argv) > 3 else 1
frag = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # 1: fragment-ordered weights (Conv3FwdP.wfrag: what the step feeds the small-grid kernels)
o = ops.dn_opts()
B, (D, H, W) = 4, (64, 64, 32)
gd = (D // 4 >> blk, H // 4 >> blk, W // 4 >> blk)
M = B * gd[0] * gd[1] * gd[2]
lib, S = _lib.load_library(), _lib.structs()
coords = ops.init_coords(B, gd, dev)
keep, arrs = [], []
for g in range(G * nsets):
    if g % G == 0:
        blocks = []
    y1 = torch.randn(M, 128, device=dev)
    wp = ops.pack_conv3_frag(torch.randn(32, 128, 3, 3, 3, device=dev) * 0.02)[0] if frag else torch.randn(32 * 27 * 128, device=dev) * 0.02
    s, q = y1.double().sum(0), (y1.double() ** 2).sum(0)
    bn = ops.bnsrc(torch.ones(128, device=dev), torch.zeros(128, device=dev), M, True, s, q)
    slab = torch.zeros(M, 256, device=dev)
    os_, oq = torch.zeros(32, dtype=torch.float64, device=dev), torch.zeros(32, dtype=torch.float64, device=dev)
    out = slab[:, 64:96]
    keep.append((y1, wp, s, q, slab, os_, oq))
    blocks.append(S["Conv3FwdP"](y1.data_ptr(), coords.data_ptr(), ops.dims3(gd), M, wp.data_ptr(), out.data_ptr(), out.stride(0), bn,
                                 os_.data_ptr(), oq.data_ptr(), None, 27, 1, 64, frag))
    if g % G == G - 1:
        arrs.append((S["Conv3FwdP"] * G)(*blocks))
def launch(i):
    _lib.check(lib.mms_conv3_fwd_group(arrs[i % nsets], G, ops.opts_ref(o), ops.stream()), "conv3_fwd_group")
for i in range(3): launch(i)
torch.cuda.synchronize()
# timed from a captured graph (as the step issues its launches): issued one by one from Python, launches below ~12 us are host-bound
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    for i in range(reps): launch(i)
graph.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
graph.replay()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e3 / reps
print(f"block {blk + 1} M={M} G={G} sets={nsets} frag={frag}: conv3 fwd avg {t:.1f} us ({G * 2.0 * M * 27 * 128 * 32 / t / 1e6:.1f} TFLOP/s)")
