#!/bin/bash
# Collects the round's judged profile artifacts on the GPU box into gpurun_out/prof_final/ (copy what you keep to profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_final; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
# 1. kernel-trace summary of the default bench
rm -rf /tmp/pf1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf1 -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(ls /tmp/pf1/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
# 1b. the roofline leg alone: the dominant kernel's isolated group launches (what bench.py times live with HIP events)
rm -rf /tmp/pf2; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf2 -- python3 $R/bench.py --roofline-only > $O/roofline_leg.json 2> /dev/null
cp $(ls /tmp/pf2/*/*_kernel_stats.csv | head -1) $O/roofline_leg_kernel_stats.csv
python3 - $(ls /tmp/pf2/*/*_kernel_trace.csv | head -1) >> $O/roofline_leg.json <<'PY'
import csv, sys, json
# the leg launches the weight-gradient kernel block by block: 3 warm-up + 20 timed launches per block shape, in order
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'Conv3BwdWOp' in r['Kernel_Name'] or 'conv3_bwdw_mt' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
assert len(rows) == 4 * 23, len(rows)
layers, avg, names = (6, 12, 24, 16), [], []
for b in range(4):
    seg = rows[23 * b + 3:23 * (b + 1)]
    avg.append(sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / len(seg) / 1e3)
    names.append(seg[0]['Kernel_Name'].split('(')[0][-40:] + ' z=' + seg[0]['Grid_Size_Z'])
w = sum(a * l for a, l in zip(avg, layers)) / 58.0
print(json.dumps({"rocprofv3_kernel_trace_of_this_command": {"per_block_avg_us": [round(a, 2) for a in avg], "kernels": names,
                  "layer_weighted_avg_us": round(w, 2)}}))
PY
# 2. PMC passes (kernel-trace only) on the dominant kernel, group launches, per block shape
for blk in 0 1 2 3; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pm -- python3 $R/tools/prof_conv3bwdw.py $blk 10 10 > /dev/null 2>&1
    f=$(ls /tmp/pm/*/*_counter_collection.csv | head -1)
    python3 - "$f" $blk $c >> $O/pmc_conv3bwdw.txt <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if ('Conv3BwdWOp' in r['Kernel_Name'] or 'conv3_bwdw_mt' in r['Kernel_Name']) and r['Counter_Name']==sys.argv[3]]
v=[float(r['Counter_Value']) for r in rows]
print('block',sys.argv[2],sys.argv[3],'dispatches',len(v),'mean',sum(v)/max(len(v),1))
PY
  done
done
# 3. plain default bench line (no profiler)
cd $R && python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
# 4. per-grid breakdown of one group step
TOPN=400 GROUP=10 STEPS=80 bash $R/tools/prof_step.sh > $O/group10_step_breakdown.txt 2>&1
TOPN=400 GROUP=5 STEPS=40 bash $R/tools/prof_step.sh > $O/group5_step_breakdown.txt 2>&1
python3 $R/tools/make_traffic_json.py $O/pmc_conv3bwdw.txt 10 > $O/pmc_conv3bwdw_traffic.json
