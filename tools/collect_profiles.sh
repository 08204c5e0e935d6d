#!/bin/bash
# Collects the round's judged profile artifacts on the GPU box into gpurun_out/prof_final/ (copy what you keep to profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_final; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
# 1. kernel-trace summary of the default bench (config 3 headline; extra legs off so that the table is the K-fold epoch)
rm -rf /tmp/pf1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf1 -- python3 $R/bench.py --no-cpu-baseline --no-h2d --no-many-folds > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(ls /tmp/pf1/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
echo "step 1 done"
# 2. the roofline leg alone: exactly the launches bench.py times live with HIP events
rm -rf /tmp/pf2; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf2 -- python3 $R/bench.py --roofline-only > $O/roofline_leg.json 2> /dev/null
cp $(ls /tmp/pf2/*/*_kernel_stats.csv | head -1) $O/roofline_leg_kernel_stats.csv
python3 $R/tools/pmc_conv2.py trace $(ls /tmp/pf2/*/*_kernel_trace.csv | head -1) $O/roofline_leg.json >> $O/roofline_leg.json
echo "step 2 done"
# 3. PMC passes (kernel-trace only, one counter per pass) over the same command
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c; rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pm_$c -- python3 $R/bench.py --roofline-only > /dev/null 2>&1
done
python3 $R/tools/pmc_conv2.py pmc $(ls /tmp/pm_FETCH_SIZE/*/*_counter_collection.csv | head -1) $(ls /tmp/pm_WRITE_SIZE/*/*_counter_collection.csv | head -1) $O/roofline_leg.json > $O/pmc_conv2_traffic.json
cp $O/pmc_conv2_traffic.json $R/profiles/r04_pmc_conv2_traffic.json      # bench.py's roofline.traffic reads the round's file (same tree, same run)
echo "step 3 done"
# 4. plain default bench line (no profiler)
cd $R && python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "step 4 done"
# 5. per-grid breakdown of one lock-step step, one sub-group alone on the GPU
for G in 1 2 3 5; do TOPN=400 GROUP=$G STEPS=$((12 * G)) bash $R/tools/prof_step.sh > $O/group${G}_step_breakdown.txt 2>&1; done
echo "step 5 done"
# 6. BASELINE config 4's per-rank problem at N = 1 (one model, 128x128x64 volumes, 2 patients per step): a full bench line
cd $R && python3 bench.py --mode ddp --volume 128 128 64 --batch 2 --cpu-steps 3 > $O/c4_rank_bench.json 2> $O/c4_rank_bench.err
echo "step 6 done"
