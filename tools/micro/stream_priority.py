"""Does HIP stream priority let a chain of small kernels get CU slots ahead of another stream's chip-filling kernels?
Stream A: back-to-back fp32 GEMMs (each fills the chip for ~100 us).  Stream B: a dependent chain of tiny kernels.
Reports B's chain time alone, beside A at equal priority, and beside A with B on a high-priority stream (and A's rate)."""
import time
import torch

dev = torch.device("cuda", 0)
lo, hi = -1, 0
try:
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    a, b = ctypes.c_int(0), ctypes.c_int(0)
    hip.hipDeviceGetStreamPriorityRange(ctypes.byref(a), ctypes.byref(b))
    print("priority range (least, greatest):", a.value, b.value)
except Exception as e:
    print("priority range query failed:", e)

X = torch.randn(4096, 2048, device=dev)
W = torch.randn(2048, 2048, device=dev)
small = torch.zeros(64, 256, device=dev)


def heavy(n):
    for _ in range(n):
        torch.mm(X, W)


def chain(n):
    t = small
    for _ in range(n):
        t = t + 1.0
    return t


def graph_of(fn, n, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn(3)
        stream.synchronize()
        with torch.cuda.graph(g, stream=stream):
            fn(n)
    return g


def run(prio_b):
    sa = torch.cuda.Stream(device=dev, priority=0)
    sb = torch.cuda.Stream(device=dev, priority=prio_b)
    ga = graph_of(heavy, 200, sa)
    gb = graph_of(chain, 400, sb)
    torch.cuda.synchronize()
    # B alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sb):
        e0.record(sb); gb.replay(); e1.record(sb)
    torch.cuda.synchronize()
    alone = e0.elapsed_time(e1)
    # A alone
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sa):
        a0.record(sa); ga.replay(); a1.record(sa)
    torch.cuda.synchronize()
    a_alone = a0.elapsed_time(a1)
    # together
    with torch.cuda.stream(sa):
        a0.record(sa); ga.replay(); a1.record(sa)
    with torch.cuda.stream(sb):
        e0.record(sb); gb.replay(); e1.record(sb)
    torch.cuda.synchronize()
    print(f"B priority {prio_b:2d}: chain of 400 tiny kernels alone {alone:7.3f} ms, beside A {e0.elapsed_time(e1):7.3f} ms | "
          f"A (200 GEMMs) alone {a_alone:7.3f} ms, beside B {a0.elapsed_time(a1):7.3f} ms")


for p in (0, -1, 0, -1):
    run(p)
