#!/bin/bash
# A/B of the round-4 experiment toggles on the single-stream / two-lane bench (each line: value, single_stream, trunk kernel groups)
run() { python bench.py --no-cpu-baseline --no-secondary > gpurun_out/exp_$1.json 2> gpurun_out/exp_$1.err; python - "$1" <<'PY'
import json, sys
d = json.load(open("gpurun_out/exp_%s.json" % sys.argv[1]))
kg = d["roofline"]["kernel_groups"]
print(sys.argv[1], d["value"], d["single_stream"]["value"], {k.split(" ")[0] + ("/" + k.split("(")[1][:10] if "(" in k else ""): v["ms"] for k, v in kg.items() if "res*" in k})
PY
}
run base
BSR_EXP_HALF_TILE=1 run halftile
BSR_EXP_C3Q_NI4=1 run c3qni4
run base2
