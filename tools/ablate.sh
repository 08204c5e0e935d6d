#!/bin/bash
# Timing-only ablation builds (GPU box): rebuild the library with -DMMS_ABLATE_* / kernel-local switches, time one kernel in
# isolation, restore the normal build.  These are diagnostic builds -- their RESULTS are wrong by construction (loads, MFMAs or
# epilogues removed); they only answer "which phase bounds this kernel".   usage: tools/ablate.sh <target>
#   conv3fwd    tile-GEMM conv2 forward, blocks ${BLOCKS:-0 3} (rocprofv3 kernel durations)
#   c3m         multi-tap conv2 forward, taps executed 27 / 19 / 9 / 1
#   c3w         multi-tap conv2 weight gradient: loads / MFMA / LDS reads / barriers
#   conv3bwdw   tile-GEMM conv2 weight gradient: atomic flush, MFMA, loaders
#   conv0bw     conv0 weight gradient: epilogue, MFMA, staging
#   conv1fwd    1x1x1 forward at block-3/4 shapes;   conv1fwd_b1: at block-1/2 shapes of a 10-model group
#   stats       cost of the statistic epilogue of the conv2 forward
#   c5          config 5's two 21-GFLOP Linear launches
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
build() { (cd $R && MMS_CXXFLAGS="$1" python -m multimodal_survival_prediction_amd._build --force > /dev/null 2>&1); }
conv3_trace() { rm -rf /tmp/abl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -- python3 $R/tools/prof_conv3.py $1 30 > /dev/null 2>&1
                grep 'Conv3FwdOp' $(ls /tmp/abl/*/*_kernel_stats.csv | head -1) | awk -F, '{printf "avg %.1f us min %.1f", $4/1000, $6/1000}'; }
ALL="-DMMS_ABLATE_MMA -DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE"
case "$1" in
  conv3fwd)    V=("" "-DMMS_ABLATE_MMA" "-DMMS_ABLATE_ALOAD" "-DMMS_ABLATE_BLOAD" "-DMMS_ABLATE_ALOAD -DMMS_ABLATE_BLOAD" "$ALL")
               run() { for w in ${BLOCKS:-0 3}; do echo "block $w: $(conv3_trace $w)"; done; } ;;
  stats)       V=("$ALL" "$ALL -DMMS_ABLATE_STATS"); run() { conv3_trace 0; } ;;
  c3m)         V=("-DC3M_TAPS=26" "-DC3M_TAPS=18" "-DC3M_TAPS=8" "-DC3M_TAPS=0"); run() { python3 $R/tools/prof_conv3fwd_group.py 0 10 2>/dev/null; } ;;
  c3w)         V=("" "-DC3W_NO_LOAD" "-DC3W_NO_MFMA" "-DC3W_NO_READ" "-DC3W_NO_LOAD -DC3W_NO_READ" "-DC3W_NO_LOAD -DC3W_NO_READ -DC3W_NO_BARRIER" "-DC3W_NO_LOAD -DC3W_NO_MFMA")
               run() { echo "b0x10 $(python3 $R/tools/prof_conv3bwdw.py 0 40 10 1024 2 2>/dev/null | head -1) | b0x5/512 $(python3 $R/tools/prof_conv3bwdw.py 0 40 5 512 2 2>/dev/null | head -1)"; } ;;
  conv3bwdw)   V=("" "-DMMS_ABLATE_FLUSH" "-DMMS_ABLATE_FLUSH -DMMS_ABLATE_MMA" "-DMMS_ABLATE_FLUSH -DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE")
               run() { for c in "0 40" "1 100" "2 200"; do echo -n "b${c%% *} $(python3 $R/tools/prof_conv3bwdw.py $c 10 2>/dev/null | head -1) | "; done; echo; } ;;
  conv0bw)     V=("" "-DC0_NO_EPI" "-DC0_NO_MMA" "-DC0_NO_STAGE" "-DC0_NO_MMA -DC0_NO_EPI" "-DC0_NO_STAGE -DC0_NO_EPI")
               run() { echo "$(python3 $R/tools/prof_conv0bw.py 5 20 2>/dev/null) | $(python3 $R/tools/prof_conv0bw.py 1 20 2>/dev/null)"; } ;;
  conv1fwd)    V=("" "-DMMS_ABLATE_STATS" "-DMMS_ABLATE_MMA" "-DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE" "$ALL" "$ALL -DMMS_ABLATE_STATS" "$ALL -DMMS_ABLATE_STATS -DMMS_ABLATE_SETUP")
               run() { echo "$(python3 $R/tools/prof_conv1fwd.py 2 640 5 2>/dev/null) | $(python3 $R/tools/prof_conv1fwd.py 3 768 5 2>/dev/null) | $(python3 $R/tools/prof_conv1fwd.py 1 320 5 2>/dev/null)"; } ;;
  conv1fwd_b1) V=("" "-DMMS_ABLATE_STATS" "-DMMS_ABLATE_SETUP" "-DMMS_ABLATE_MMA" "-DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE" "$ALL" "$ALL -DMMS_ABLATE_STATS -DMMS_ABLATE_SETUP")
               run() { echo "$(python3 $R/tools/prof_conv1fwd.py 0 64 10 2>/dev/null) | $(python3 $R/tools/prof_conv1fwd.py 0 224 10 2>/dev/null) | $(python3 $R/tools/prof_conv1fwd.py 1 320 10 2>/dev/null)"; } ;;
  c5)          V=("" "-DMMS_ABLATE_MMA" "-DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE" "-DMMS_ABLATE_GLOAD")
               run() { STEPS=30 bash $R/tools/prof_c5.sh 2>/dev/null | grep -E "g32x16x1|g16x79x1|g8x16x4|g32x8x1"; } ;;
  *) sed -n 2,14p $0; exit 1 ;;
esac
for v in "${V[@]}"; do build "$v"; echo "variant [$v]: $(run)"; done
build ""
