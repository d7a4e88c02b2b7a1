#!/bin/bash
# A/B of environment toggles on the bench (value = two lanes, single stream, trunk kernel groups): bash scratch/ab.sh "NAME=VAL ..." ...
run() { env $2 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/ab_$1.json 2> gpurun_out/ab_$1.err; python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_%s.json" % sys.argv[1]))
kg = d["roofline"]["kernel_groups"]
print(sys.argv[2] or "(default)", "| value", d["value"], "single", d["single_stream"]["value"], {k.split("(")[1][:18]: v["ms"] for k, v in kg.items() if "res*" in k})
PY
}
i=0
for cfg in "$@"; do i=$((i+1)); run $i "$cfg"; done
