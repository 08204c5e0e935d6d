#!/bin/bash
# As tools/ablate_step.sh, by kernel kind inside dense blocks 2 and 3.
out=${1:-gpurun_out/r03/ablate_step_kinds.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
run() {
    MMS_DEBUG_SKIP_ACK=results-are-wrong MMS_DEBUG_SKIP=$2 timeout -k 10 240 python bench.py --no-cpu-baseline --no-many-folds --no-h2d > /tmp/abl.json 2> /tmp/abl.err || { echo "$1: bench failed" >> "$out"; return 1; }
    python - "$1" "$2" >> "$out" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/abl.json") if l.startswith("{")][-1])
print(f"{sys.argv[1]:34s} mask {sys.argv[2]:>12s}  epoch {d['value']:7.0f} patients/s  {d['ms_per_step']:.3f} ms/step  single model {d['config'].get('single_chain_patients_per_s', 0):6.0f}")
PY
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
