#!/bin/bash
# A/B of COMPILE-TIME variants on the GPU box: for each "flags" argument rebuild the library with BSR_EXTRA_FLAGS, run bench.py for the
# given dtype and print the kernel-group times.   bash scratch/ab_build.sh f16 "" "-DBSR_H16_RING_F16=12" ...
dtype=$1; shift
i=0
for flags in "$@"; do
  i=$((i+1))
  export BSR_EXTRA_FLAGS="$flags"
  python -c "from blindshadowremoval_amd.build import build_library; build_library(force=True)" || { echo "build failed: $flags"; continue; }
  python bench.py --dtype $dtype --no-cpu-baseline --no-secondary --streams 1 --steps 20 > gpurun_out/abb_$i.json 2> gpurun_out/abb_$i.err
  python - "$i" "$flags" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/abb_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[2] or "(default)", "FAILED", e, open("gpurun_out/abb_%s.err" % sys.argv[1]).read()[-400:]); sys.exit(0)
kg = d["roofline"]["kernel_groups"]
print("%-60s value %8.1f  all_kernels_ms %.4f" % (sys.argv[2] or "(default)", d["value"], d["roofline"]["all_kernels_ms"]))
for k, v in kg.items():
    print("      %-95s %.4f ms" % (k[:95], v["ms"]))
PY
done
unset BSR_EXTRA_FLAGS
python -c "from blindshadowremoval_amd.build import build_library; build_library(force=True)"
