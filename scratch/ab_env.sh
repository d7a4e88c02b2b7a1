#!/bin/bash
# A/B of ENVIRONMENT switches for a dtype on the GPU box: bash scratch/ab_env.sh f16 "" "BSR_GEMM_ONE_WG=1" ...
dtype=$1; shift
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg python bench.py --dtype $dtype --no-cpu-baseline --no-secondary --streams 1 --steps 20 > gpurun_out/abe_$i.json 2> gpurun_out/abe_$i.err
  python - "$i" "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/abe_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[2] or "(default)", "FAILED", e, open("gpurun_out/abe_%s.err" % sys.argv[1]).read()[-400:]); sys.exit(0)
kg = d["roofline"]["kernel_groups"]
print("%-40s value %8.1f  all_kernels_ms %.4f  " % (sys.argv[2] or "(default)", d["value"], d["roofline"]["all_kernels_ms"]) + "  ".join("%s %.4f" % (k.split("(")[-1][:14], v["ms"]) for k, v in kg.items() if "res*" in k))
PY
done
