#!/bin/bash
# Timing ablation of the K = 5 epoch: leave one kernel class out of the step (MMS_DEBUG_SKIP, csrc/dn_net.hip -- results are wrong,
# only the clock is read) and report what the epoch gains.  The gain bounds what optimising that class can buy in the overlapped
# three-stream schedule (where a kernel's duration and its cost to the step are different things).
# usage: bash tools/ablate_step.sh [out-file]     (on the GPU box)
out=${1:-gpurun_out/r03/ablate_step.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
run() {   # name mask
    MMS_DEBUG_SKIP_ACK=results-are-wrong MMS_DEBUG_SKIP=$2 timeout -k 10 240 python bench.py --no-cpu-baseline --no-many-folds --no-h2d > /tmp/abl.json 2> /tmp/abl.err || { echo "$1: bench failed" >> "$out"; return 1; }
    python - "$1" "$2" >> "$out" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/abl.json") if l.startswith("{")][-1])
print(f"{sys.argv[1]:34s} mask {sys.argv[2]:>12s}  epoch {d['value']:7.0f} patients/s  {d['ms_per_step']:.3f} ms/step  single model {d['config'].get('single_chain_patients_per_s', 0):6.0f}")
PY
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
run "weight pack + gradient unpack"    0xC00000000 &&
run "block-3 conv1 bwd-data (c1s)"     0x400000
cat "$out"
