#!/bin/bash
# Diagnostic library variants built HERE (hipcc cross-compiles gfx950 without a GPU) so that a GPU call only spends time running them:
#   tools/build_variant.sh <name> "<extra hipcc flags>" <file.hip> [<file.hip> ...]
# compiles the named sources of csrc/ with the extra flags and links them with the standard objects of the other sources into
# multimodal_survival_prediction_amd/csrc/build/variants/<name>.so (git-ignored; ships with the gpurun snapshot).  On the GPU box a
# diagnostic script selects one by copying it over multimodal_survival_prediction_amd/libmmsurv_hip.so of its SCRATCH copy of the tree.
# Variants built with -DC3M_* / -DMMS_ABLATE_* switches compute wrong results by construction: timing only.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/multimodal_survival_prediction_amd; B=$P/csrc/build; V=$B/variants
name=$1; flags=$2; shift 2
python -m multimodal_survival_prediction_amd._build > /dev/null
mkdir -p $V/$name
objs=""
for f in $(ls $P/csrc/*.hip); do
  b=$(basename $f .hip); o=$B/$b.o
  for v in "$@"; do if [ "$v" = "$b.hip" ]; then o=$V/$name/$b.o; /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -fPIC -std=c++17 -Wno-unused-value $flags -c $f -o $o; fi; done
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$name.so $objs
echo $V/$name.so
