#!/bin/bash
# A/B of COMPILE-TIME variants on the GPU box by per-launch device time: bash scratch/ab_layers.sh <dtype> <grep pattern> "flags1" "flags2" ...
dtype=$1; pat=$2; shift 2
for flags in "$@"; do
  export BSR_EXTRA_FLAGS="$flags"
  python -c "from blindshadowremoval_amd.build import build_library; build_library(force=True)" 2>&1 | grep -E "error" | head -3
  echo "== ${flags:-(default)}"
  python scratch/layer_times.py $dtype | grep -E "$pat"
done
unset BSR_EXTRA_FLAGS
