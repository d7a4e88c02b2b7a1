set -x
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "range or tiny or forward_matches or full_batch" 2>&1 | tail -8 > gpurun_out/r3_t2.log
cat gpurun_out/r3_t2.log
for d in 0 50 100 150; do
  BSR_DEPHASE=$d python bench.py --no-cpu-baseline --no-secondary --steps 30 > gpurun_out/r3_dephase_$d.json 2>/dev/null
  python - <<PY
import json
j=json.load(open("gpurun_out/r3_dephase_$d.json"))
print("DEPHASE $d value", j["value"], "ms", j["ms_per_step"], "reps", j["repeats"]["ms_per_step_all"])
for k,v in j["roofline"]["kernel_groups"].items(): print("   %-100s %7.4f ms frac %.3f" % (k[:100], v["ms"], v["frac"]))
PY
done
python bench.py --dtype f32x3 --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; j=json.load(sys.stdin); print('f32x3 value', j['value'], j['ms_per_step'])"
