"""Premise check for a CU-partitioned schedule: stream A (chip-filling GEMMs) restricted to most of the CUs, stream B (a dependent
chain of tiny kernels) on the rest.  Does B then run at its stand-alone speed beside A, what does A lose, and do graph replays on
CU-masked streams keep the mask?  (hipExtStreamCreateWithCUMask; torch sees the streams as ExternalStream.)"""
import ctypes
import sys
import torch

dev = torch.device("cuda", 0)
torch.cuda.init()
torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count
print("CUs:", NCU)


def masked_stream(bits):
    """bits: iterable of CU indices enabled"""
    nwords = (NCU + 31) // 32
    arr = (ctypes.c_uint32 * nwords)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(nwords), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value, device=dev)


X = torch.randn(4096, 2048, device=dev)
W = torch.randn(2048, 2048, device=dev)
small = torch.zeros(64, 256, device=dev)
mid = torch.zeros(2048, 2048, device=dev)      # a "medium" kernel: 4 M elements


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


def timed(stream, g):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record(stream); g.replay(); e1.record(stream)
    return e0, e1


def run(name, sa, sb):
    ga = graph_of(heavy, 200, sa)
    gb = graph_of(chain, 400, sb)
    torch.cuda.synchronize()
    e = timed(sb, gb); torch.cuda.synchronize(); b_alone = e[0].elapsed_time(e[1])
    e = timed(sa, ga); torch.cuda.synchronize(); a_alone = e[0].elapsed_time(e[1])
    ea = timed(sa, ga); eb = timed(sb, gb); torch.cuda.synchronize()
    print(f"{name:42s}: B chain alone {b_alone:7.3f} ms, beside A {eb[0].elapsed_time(eb[1]):7.3f} ms | A alone {a_alone:7.3f} ms, beside B {ea[0].elapsed_time(ea[1]):7.3f} ms", flush=True)


run("no masks", torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
for nb in (32, 64):
    # B gets the LAST nb mask bits, A the rest
    run(f"A: bits 0..{NCU - nb - 1}, B: last {nb} bits", masked_stream(range(NCU - nb)), masked_stream(range(NCU - nb, NCU)))
    # B gets every (NCU/nb)-th bit
    step = NCU // nb
    bsel = set(range(0, NCU, step))
    run(f"A: the rest, B: every {step}th bit ({nb})", masked_stream([i for i in range(NCU) if i not in bsel]), masked_stream(sorted(bsel)))
run("A masked to 224, B unmasked", masked_stream(range(NCU - 32)), torch.cuda.Stream(device=dev))
