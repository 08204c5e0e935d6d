#!/bin/bash
# Timing ablation of the K = 5 epoch: leave one kernel class out of the step (MMS_DEBUG_SKIP in a -DMMS_ABLATE_STEP build, csrc/dn_net.hip -- results are wrong,
# only the clock is read) and report what the epoch gains.  The gain bounds what optimising that class can buy in the overlapped
# three-stream schedule (where a kernel's duration and its cost to the step are different things).
# usage: bash tools/ablate_step.sh [out-file]     (on the GPU box)
out=${1:-gpurun_out/r04/ablate_step.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
# needs a timing-ablation BUILD (the shipped library has no such switch): MMS_CXXFLAGS=-DMMS_ABLATE_STEP python -m multimodal_survival_prediction_amd._build --force
# bench.py refuses to print a bench line from such a build; --timed-only prints the epoch rate only.
run() {   # name mask
    MMS_DEBUG_SKIP=$2 timeout -k 10 240 python bench.py --timed-only > /tmp/abl.txt 2> /tmp/abl.err || { echo "$1: bench failed" >> "$out"; return 1; }
    printf '%-34s mask %12s  %s\n' "$1" "$2" "$(grep timed-only /tmp/abl.txt | tail -1)" >> "$out"
}
run "baseline"                         0x0 &&
run "block-1 conv2 forward"            0x10 &&
run "block-1 conv2 backward-data"      0x100 &&
run "block-1 conv2 weight gradient"    0x1000 &&
run "block-1 conv1 fwd+bwd+apply"      0x1110001 &&
run "block 2 (all dense-layer work)"   0x2222222 &&
run "block 3 (all dense-layer work)"   0x4444444 &&
run "block 4 (persistent + wgrads)"    0x300088000 &&
run "stem fwd+bwd"                     0x30000000 &&
run "transitions fwd+bwd"              0xC0000000 &&
run "block-3 conv1 bwd-data (c1s)"     0x400000
cat "$out"
