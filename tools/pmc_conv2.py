"""rocprofv3 counter/trace CSVs of `bench.py --roofline-only` -> per-op figures of the conv2 family (GPU box).
usage: pmc_conv2.py trace  <kernel_trace.csv> <roofline_leg.json>                       -> live-vs-trace agreement JSON (durations)
       pmc_conv2.py pmc    <FETCH counter csv> <WRITE counter csv> <roofline_leg.json>  -> profiles/rNN_pmc_conv2_traffic.json content
roofline_leg.json = the line `bench.py --roofline-only` printed: its "manifest" lists, in launch order, what the leg launched --
per sub-group size G, dense block and op: `calls` = 3 warm-up + 20 timed calls of ONE launch with `members` (model, layer) members
(a call is one main kernel + its tap-split reduce kernel where the driver splits), weighted by `launches_per_step`."""
import csv, json, sys


def kind(name):
    if "conv3_fwd_reduce" in name: return ("fwd", True)
    if "conv3_bwd_data_reduce" in name: return ("bwd_data", True)
    if "conv3_fwd_mt" in name or "Conv3FwdOp" in name or "conv3s_fwd_kernel" in name: return ("fwd", False)
    if "Conv3BwdDataOp" in name or "conv3s_bwd_data_kernel" in name or "conv3_bwd_data_mt" in name: return ("bwd_data", False)
    if "conv3_bwdw_mt" in name or "Conv3BwdWOp" in name: return ("bwd_weight", False)
    return None


def calls(rows, value):
    """-> list of [op, value summed over the call's kernels] in dispatch order"""
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


def per_op(cs, manifest):
    """mean per timed call, weighted by the launches per step and pooled over the sub-group sizes (as bench.roofline_block pools them)"""
    assert len(cs) == sum(e["calls"] for e in manifest), (len(cs), sum(e["calls"] for e in manifest))
    acc, i = {}, 0
    for e in manifest:
        seg = cs[i:i + e["calls"]]; i += e["calls"]
        assert all(c[0] == e["op"] for c in seg), (e, seg[0][0])
        a = acc.setdefault(e["op"], [0.0, 0.0])
        a[0] += sum(c[1] for c in seg[3:]) / (e["calls"] - 3) * e["launches_per_step"]
        a[1] += e["launches_per_step"]
    return {op: a / n for op, (a, n) in acc.items()}


def main():
    mode = sys.argv[1]
    leg = json.loads([l for l in open(sys.argv[-1]) if l.startswith("{")][0])
    manifest = leg["manifest"]
    groups = sorted({e["G"] for e in manifest}, reverse=True)
    if mode == "trace":
        rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
        d = per_op(calls(rows, lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3), manifest)
        print(json.dumps({"rocprofv3_kernel_trace_of_bench_roofline_only": {"avg_launch_us": {k: round(v, 2) for k, v in d.items()},
                          "live_hip_events_avg_launch_us": {k: round(v["avg_launch_us"], 2) for k, v in leg["family"].items()},
                          "sub_group_sizes": groups, "note": "kernel durations (main kernel + tap-split reduce kernel) of the timed calls, "
                          "weighted by launches per step and pooled over the sub-group sizes exactly as bench.py's roofline leg pools its HIP-event timings"}}))
        return
    key = lambda r: int(r["Dispatch_Id"])
    f = sorted((r for r in csv.DictReader(open(sys.argv[2])) if r["Counter_Name"] == "FETCH_SIZE"), key=key)
    w = sorted((r for r in csv.DictReader(open(sys.argv[3])) if r["Counter_Name"] == "WRITE_SIZE"), key=key)
    fe = per_op(calls(f, lambda r: float(r["Counter_Value"])), manifest)
    wr = per_op(calls(w, lambda r: float(r["Counter_Value"])), manifest)
    sub = leg["kernel"].split("sub-groups of ")[-1].split(" ")[0]
    out = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE over `python3 bench.py --roofline-only` "
                  "(KB per dispatch; main kernel + tap-split reduce kernel summed per call; weighted by launches per step, pooled over the "
                  "sub-groups %s); gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE x2 for wide coalesced reads, WRITE_SIZE exact" % sub,
           "sub_groups": [int(x) for x in sub.split("+")]}
    alg = {}
    for e in manifest:
        M = e["M"]
        per_member = {"fwd": M * 128 * 4 + M * 32 * 4 + 27 * 32 * 128 * 4, "bwd_data": M * 32 * 4 + 2 * M * 128 * 4 + 27 * 32 * 128 * 4,
                      "bwd_weight": M * 128 * 4 + M * 32 * 4 + 27 * 32 * 128 * 4}[e["op"]]
        a = alg.setdefault(e["op"], [0.0, 0.0])
        a[0] += e["members"] * per_member * e["launches_per_step"]; a[1] += e["launches_per_step"]
    for op in fe:
        out[op] = {"FETCH_SIZE_KB": round(fe[op], 1), "WRITE_SIZE_KB": round(wr[op], 1),
                   "avg_hbm_bytes_per_launch": round((2 * fe[op] + wr[op]) * 1024),
                   "avg_algorithmic_bytes_per_launch": round(alg[op][0] / alg[op][1]),
                   "ratio": round((2 * fe[op] + wr[op]) * 1024 / (alg[op][0] / alg[op][1]), 2)}
    print(json.dumps(out, indent=1))


main()
