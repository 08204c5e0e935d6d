#!/bin/bash
# As tools/ablate_step.sh, by kernel kind inside dense blocks 2 and 3.
out=${1:-gpurun_out/r04/ablate_step_kinds.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
# needs a timing-ablation BUILD (the shipped library has no such switch): MMS_CXXFLAGS=-DMMS_ABLATE_STEP python -m multimodal_survival_prediction_amd._build --force
# bench.py refuses to print a bench line from such a build; --timed-only prints the epoch rate only.
run() {   # name mask
    MMS_DEBUG_SKIP=$2 timeout -k 10 240 python bench.py --timed-only > /tmp/abl.txt 2> /tmp/abl.err || { echo "$1: bench failed" >> "$out"; return 1; }
    printf '%-34s mask %12s  %s\n' "$1" "$2" "$(grep timed-only /tmp/abl.txt | tail -1)" >> "$out"
}
run "baseline" 0x0 &&
for b in 1 2; do
  bn=$((b+1))
  run "block $bn conv1 forward"          $(printf '0x%x' $((1 << (0 + b)))) &&
  run "block $bn conv2 forward"          $(printf '0x%x' $((1 << (4 + b)))) &&
  run "block $bn conv2 backward-data"    $(printf '0x%x' $((1 << (8 + b)))) &&
  run "block $bn conv2 weight gradient"  $(printf '0x%x' $((1 << (12 + b)))) &&
  run "block $bn conv1 weight gradient"  $(printf '0x%x' $((1 << (16 + b)))) &&
  run "block $bn conv1 backward-data"    $(printf '0x%x' $((1 << (20 + b)))) &&
  run "block $bn bn_bwd_apply"           $(printf '0x%x' $((1 << (24 + b)))) || break
done
run "baseline (again)" 0x0
cat "$out"
