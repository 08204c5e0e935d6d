#!/bin/bash
# A/B of the small-grid conv2 kernels' launch knobs inside a 3-model lock-step step (GPU box): per-kernel times from rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03; mkdir -p $O
run() { name=$1; shift; env "$@" TOPN=400 GROUP=${GROUP:-3} STEPS=$((12 * ${GROUP:-3})) bash $R/tools/prof_step.sh > $O/ab_$name.txt 2>&1; echo "== $name: $(head -1 $O/ab_$name.txt)"; grep -E "conv3s|Conv3Fwd|Conv3BwdData|reduce" $O/ab_$name.txt | cut -c1-120; }
run d9_6 MMS_C3S_D1=9 MMS_C3S_D2=6
run d14_9 MMS_C3S_D1=14 MMS_C3S_D2=9
run d27_13 MMS_C3S_D1=27 MMS_C3S_D2=13
run jn1_d27 MMS_CONV3_SMALL=1 MMS_C3S_D1=27
run jn1_d14 MMS_CONV3_SMALL=1 MMS_C3S_D1=14
