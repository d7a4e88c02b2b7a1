#!/bin/bash
# first GPU pass of round 6: the new attention kernel alone, then the 16-bit parity tests, then the 16-bit bench lines
mkdir -p gpurun_out
python scratch/att4_check.py > gpurun_out/att4_check.txt 2>&1; tail -8 gpurun_out/att4_check.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f32x3 or f16 or attention or x3 or split or dtype" > gpurun_out/r6_pytest16.txt 2>&1; tail -15 gpurun_out/r6_pytest16.txt
for dt in f16 f32x3; do
  python bench.py --dtype $dt --no-cpu-baseline --no-secondary --streams 1 --steps 20 > gpurun_out/r6_first_$dt.json 2> gpurun_out/r6_first_$dt.err
  python - $dt <<'PY'
import json, sys
dt = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r6_first_%s.json" % dt).read().strip().splitlines()[-1])
    print(dt, "value", d["value"], "ms", d["ms_per_step"])
    for k, v in d["roofline"]["kernel_groups"].items(): print("   %-90s %.4f ms" % (k[:90], v["ms"]))
except Exception as e:
    print(dt, "FAILED", e); print(open("gpurun_out/r6_first_%s.err" % dt).read()[-1500:])
PY
done
