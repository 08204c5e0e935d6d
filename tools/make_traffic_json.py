"""pmc_conv3bwdw.txt (tools/collect_profiles.sh) -> profiles/r01_pmc_conv3bwdw_traffic.json (what bench.py reads as roofline.traffic)."""
import json, re, sys
txt, G = sys.argv[1], int(sys.argv[2])
B, dims = 4, (64, 64, 32)
launches = (6, 12, 24, 16)
vals = {}
for line in open(txt):
    m = re.match(r"block (\d) (\w+) dispatches (\d+) mean ([\d.]+)", line)
    if m:
        vals[(int(m.group(1)), m.group(2))] = float(m.group(4))
per, tot_h, tot_a, n = [], 0.0, 0.0, 0
for blk in range(4):
    gd = [d // 4 >> blk for d in dims]
    M = B * gd[0] * gd[1] * gd[2]
    f, w = vals[(blk, "FETCH_SIZE")], vals[(blk, "WRITE_SIZE")]
    hbm = (2 * f + w) * 1024           # gfx950: FETCH_SIZE counts half of wide coalesced reads (MI355X_MICROARCH.md); KB units
    alg = G * (M * 128 * 4 + M * 32 * 4 + 27 * 32 * 128 * 4)
    per.append(dict(block=blk + 1, M=M, launches_per_step=launches[blk], FETCH_SIZE_KB=f, WRITE_SIZE_KB=w, hbm_bytes=round(hbm),
                    algorithmic_bytes=alg))
    tot_h += hbm * launches[blk]; tot_a += alg * launches[blk]; n += launches[blk]
print(json.dumps({
    "kernel": "mms_conv3_bwd_weight_group (conv3_bwdw_mt_kernel at the block-1 shape, tile_gemm_kernel<Conv3BwdWOp> at the others)", "models_per_launch": G,
    "how": "tools/prof_conv3bwdw.py <block> 10 %d under rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE "
           "(mean of the dispatches; KB); gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE x2 for wide coalesced 16-B/lane reads, "
           "WRITE_SIZE exact (float atomics)" % G,
    "per_block": per, "avg_hbm_bytes_per_launch": round(tot_h / n), "avg_algorithmic_bytes_per_launch": round(tot_a / n),
    "note": "fabric-side bytes (Infinity-Cache hits included).  Writes are the fp32 atomics of the tap-major gradient scratch (27 x msplit "
            "x 4096 floats per model); reads dropped ~10x when the tap workgroups of a row chunk were placed on one XCD."},
    indent=1))
