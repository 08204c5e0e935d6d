"""One look at the observation recorded in DESIGN.md section 3 (round 1): two consecutive hipMemsetAsync nodes in a captured graph
followed by kernels that read both buffers.  Captures [memset A; memset B; C = A + B (kernel); A += 1; B += 1 (kernels)] on torch's
capture stream, replays it several times and checks that every replay sees zeroed A and B (C == 0)."""
import ctypes, sys, torch
hip = ctypes.CDLL("libamdhip64.so")           # the runtime torch already loaded
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
A, B, C = torch.ones(n, device=dev), torch.ones(n, device=dev), torch.full((n,), 7.0, device=dev)


MODE = sys.argv[2] if len(sys.argv) > 2 else "two"      # two: memset A, memset B | one: memset A only (B zeroed by a kernel) | gap: a kernel between them


def body():
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert hip.hipMemsetAsync(A.data_ptr(), 0, A.numel() * 4, st) == 0
    if MODE == "gap":
        C.add_(0.0)
    if MODE == "one":
        B.zero_()
    else:
        assert hip.hipMemsetAsync(B.data_ptr(), 0, B.numel() * 4, st) == 0
    torch.add(A, B, out=C)
    A.add_(1.0); B.add_(1.0)


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    g.enable_debug_mode()
except Exception as e:
    print("no debug mode:", e)
with torch.cuda.graph(g):
    body()
bad = 0
for it in range(6):
    g.replay()
    torch.cuda.synchronize()
    c = float(C.abs().max())
    nz = int((C != 0).sum())
    print("replay", it, "max |C| =", c, "nonzero elements of C:", nz, "first at", int((C != 0).nonzero()[0]) if nz else -1)
    bad += c != 0.0
print("RESULT n=%d mode=%s:" % (n, MODE), "stale reads in %d of 6 replays" % bad if bad else "all replays saw zeroed buffers")
