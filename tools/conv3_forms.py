"""Block-1 conv2 launches (B=4, 64x64x32 volumes: 8192 rows per model) timed per kernel FORM and sub-group size (GPU box).
usage: python tools/conv3_forms.py [fwd|bwd_data|bwd_weight] [G ...]
Forms are selected through the launch-shape options (MmsDnOpts, include/mmsurv.h): fwd: conv3_mt = -1 (per-tap GEMM form),
3 (multi-tap, 32-row tiles), 2 (multi-tap, 64-row tiles); bwd_weight: conv3w_mt = -1 / 2.  Prints microseconds per launch and the fraction
of the fp32 MFMA peak (157.3 TFLOP/s)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import _stat_reps, _bwdw_msplit
from multimodal_survival_prediction_amd import _lib, ops

FORMS = {"fwd": ("conv3_mt", (-1, 3, 2, 0)), "bwd_data": ("conv3_mt", (0,)), "bwd_weight": ("conv3w_mt", (-1, 2, 0))}


def main():
    op = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    Gs = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 5]
    dev = torch.device("cuda:0")
    lib, S = _lib.load_library(), _lib.structs()
    blk = int(os.environ.get("BLOCK", "0"))            # dense block whose grid the launch has (0: 16x16x8 ... 3: 2x2x1)
    gd, B = (16 >> blk, 16 >> blk, 8 >> blk), 4
    nsp = int(os.environ.get("NSPLIT", "0"))           # forward / backward-data: tap split (0 = none)
    M = B * gd[0] * gd[1] * gd[2]
    R = _stat_reps(M)
    gam, bet = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    w = torch.randn(32, 128, 3, 3, 3, device=dev) * 0.03
    wpf, wpb = ops.pack_conv3(w)
    coords = ops.init_coords(B, gd, dev)
    env, forms = FORMS[op]
    for G in Gs:
        keep, ps = [], []
        for _ in range(G):
            y1 = torch.randn(M, 128, device=dev)
            bn = ops.bnsrc(gam, bet, M, True, y1.double().sum(0), (y1.double() ** 2).sum(0))
            slab, dslab = torch.zeros(M, 256, device=dev), torch.randn(M, 256, device=dev)
            ost = torch.zeros(R, 2, 256, dtype=torch.float64, device=dev)
            bst = torch.zeros(R, 2, 128, dtype=torch.float64, device=dev)
            dbn, dwp = torch.zeros(M, 128, device=dev), torch.zeros(27 * 32 * 128, device=dev)
            out, dz = slab[:, 64:96], dslab[:, 64:96]
            part = torch.zeros(nsp * M * 128, device=dev) if nsp else None
            keep.append((y1, bn, slab, dslab, ost, bst, dbn, dwp, part))
            if op == "fwd":
                ps.append(S["Conv3FwdP"](y1.data_ptr(), coords.data_ptr(), ops.dims3(gd), M, wpf.data_ptr(), out.data_ptr(), out.stride(0), bn,
                                         ost[0, 0, 64:].data_ptr(), ost[0, 1, 64:].data_ptr(), ops.ptr(part), nsp or 27, R, 2 * 256))
            elif op == "bwd_data":
                ps.append(S["Conv3BwdDataP"](dz.data_ptr(), dz.stride(0), coords.data_ptr(), ops.dims3(gd), M, wpb.data_ptr(), y1.data_ptr(), bn,
                                             dbn.data_ptr(), bst[0, 0].data_ptr(), bst[0, 1].data_ptr(), ops.ptr(part), nsp or 27, R, 2 * 128))
            else:
                ps.append(S["Conv3BwdWP"](y1.data_ptr(), coords.data_ptr(), ops.dims3(gd), M, bn, dz.data_ptr(), dz.stride(0),
                                          dwp.data_ptr(), int(os.environ.get("MSPLIT", _bwdw_msplit(M, G))), 1))
        name = {"fwd": "Conv3FwdP", "bwd_data": "Conv3BwdDataP", "bwd_weight": "Conv3BwdWP"}[op]
        fn = getattr(lib, "mms_conv3_%s_group" % op)
        arr = (S[name] * G)(*ps)
        import ctypes
        for form in forms:
            o = ops.dn_opts(**{env: form})
            for _ in range(3):
                _lib.check(fn(arr, G, ctypes.byref(o), ops.stream()), op)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                _lib.check(fn(arr, G, ctypes.byref(o), ops.stream()), op)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            tf = G * 2.0 * M * 27 * 128 * 32 / us * 1e-6
            print("%-10s G=%d %s=%-7s %7.1f us  %5.1f TFLOP/s  %4.1f %% of peak%s" % (op, G, env, form or "default", us, tf, tf / 157.3 * 100,
                                                                                  "  (msplit %d)" % ps[0].msplit if op == "bwd_weight" else ""), flush=True)


if __name__ == "__main__":
    main()
