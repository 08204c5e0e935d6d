"""rocprofv3 counter/trace CSVs of `bench.py --roofline-only` -> per-op figures of the conv2 family (GPU box).
usage: pmc_conv2.py trace  <kernel_trace.csv>                    -> live-vs-trace agreement JSON (durations)
       pmc_conv2.py pmc    <FETCH counter csv> <WRITE counter csv> -> profiles/r03_pmc_conv2_traffic.json content
The leg launches, for every sub-group size G of the timed region, block 0..3, op in (fwd, bwd_data, bwd_weight): 3 warm-up + 20
timed calls; a call is one main kernel (+ its tap-split reduce kernel where the driver splits)."""
import csv, json, sys

LAYERS = (6, 12, 24, 16)
OPS = ("fwd", "bwd_data", "bwd_weight")
B, DIMS = 4, (64, 64, 32)


def kind(name):
    if "conv3_fwd_reduce" in name: return ("fwd", True)
    if "conv3_bwd_data_reduce" in name: return ("bwd_data", True)
    if "conv3_fwd_mt" in name or "Conv3FwdOp" in name or "conv3s_fwd_kernel" in name: return ("fwd", False)
    if "Conv3BwdDataOp" in name or "conv3s_bwd_data_kernel" in name: return ("bwd_data", False)
    if "conv3_bwdw_mt" in name or "Conv3BwdWOp" in name: return ("bwd_weight", False)
    return None


def calls(rows, value):
    """-> list of (op, value summed over the call's kernels) in dispatch order"""
    out = []
    for r in rows:
        k = kind(r["Kernel_Name"])
        if k is None:
            continue
        if k[1]:
            assert out and out[-1][0] == k[0], "reduce kernel without its main kernel"
            out[-1][1] += value(r)
        else:
            out.append([k[0], value(r)])
    return out


B4_ONE = True        # block 4's forward and backward data path run as one persistent launch each (csrc/dn_b4.hip): bench.py's leg skips (fwd | bwd_data, block 4)


def per_op(cs, groups):
    """mean per timed call, layer-weighted, pooled over the sub-group sizes (as bench.roofline_block pools them)"""
    combos = [(blk, op) for blk in range(4) for op in OPS if not (B4_ONE and blk == 3 and op in ("fwd", "bwd_data"))]
    assert len(cs) == len(groups) * len(combos) * 23, (len(cs), len(groups))
    acc = {op: [0.0, 0] for op in OPS}
    i = 0
    for _ in groups:
        for blk, op in combos:
            if True:
                seg = cs[i:i + 23]; i += 23
                assert all(c[0] == op for c in seg), (op, seg[0][0])
                acc[op][0] += sum(c[1] for c in seg[3:]) / 20.0 * LAYERS[blk]
                acc[op][1] += LAYERS[blk]
    return {op: a / n for op, (a, n) in acc.items()}


def main():
    mode, groups = sys.argv[1], tuple(int(g) for g in sys.argv[-1].split("+"))
    if mode == "trace":
        rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
        d = per_op(calls(rows, lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3), groups)
        print(json.dumps({"rocprofv3_kernel_trace_of_bench_roofline_only": {"avg_launch_us": {k: round(v, 2) for k, v in d.items()},
                          "sub_groups": groups, "note": "kernel durations (main kernel + tap-split reduce kernel) of the timed calls, "
                          "layer-weighted and pooled over the sub-group sizes exactly as bench.py's roofline leg pools its HIP-event timings"}}))
        return
    key = lambda r: int(r["Dispatch_Id"])
    f = sorted((r for r in csv.DictReader(open(sys.argv[2])) if r["Counter_Name"] == "FETCH_SIZE"), key=key)
    w = sorted((r for r in csv.DictReader(open(sys.argv[3])) if r["Counter_Name"] == "WRITE_SIZE"), key=key)
    fe = per_op(calls(f, lambda r: float(r["Counter_Value"])), groups)
    wr = per_op(calls(w, lambda r: float(r["Counter_Value"])), groups)
    out = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE over `python3 bench.py --roofline-only` "
                  "(KB per dispatch; main kernel + tap-split reduce kernel summed per call; layer-weighted, pooled over the sub-group sizes "
                  "%s); gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE x2 for wide coalesced reads, WRITE_SIZE exact" % (groups,),
           "sub_groups": groups}
    gavg = sum(groups) / len(groups)
    for op in OPS:
        alg, nl = 0.0, 0
        for blk in range(4):
            if B4_ONE and blk == 3 and op in ("fwd", "bwd_data"):
                continue
            nl += LAYERS[blk]
            gd = [d // 4 >> blk for d in DIMS]
            M = B * gd[0] * gd[1] * gd[2]
            per_model = {"fwd": M * 128 * 4 + M * 32 * 4 + 27 * 32 * 128 * 4, "bwd_data": M * 32 * 4 + 2 * M * 128 * 4 + 27 * 32 * 128 * 4,
                         "bwd_weight": M * 128 * 4 + M * 32 * 4 + 27 * 32 * 128 * 4}[op]
            alg += gavg * per_model * LAYERS[blk]
        out[op] = {"FETCH_SIZE_KB": round(fe[op], 1), "WRITE_SIZE_KB": round(wr[op], 1),
                   "avg_hbm_bytes_per_launch": round((2 * fe[op] + wr[op]) * 1024),
                   "avg_algorithmic_bytes_per_launch": round(alg / nl)}
    print(json.dumps(out, indent=1))


main()
