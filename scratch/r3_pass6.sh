python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_heads or range_guard or tiny or edge_inputs or forward_matches or full_batch" 2>&1 | tail -8
python bench.py --no-cpu-baseline --no-secondary --steps 30 2>/dev/null > gpurun_out/r3_fuse.json; python - <<PY
import json
j=json.load(open("gpurun_out/r3_fuse.json"))
print("FUSED value", j["value"], "ms", j["ms_per_step"], "reps", j["repeats"]["ms_per_step_all"], "glue", j["roofline"]["glue_ms"])
for k,v in j["roofline"]["kernel_groups"].items(): print("   %-100s %7.4f ms frac %.3f" % (k[:100], v["ms"], v["frac"]))
PY
BSR_FUSE_HEADS=0 python bench.py --no-cpu-baseline --no-secondary --steps 30 2>/dev/null | python -c "import json,sys; j=json.load(sys.stdin); print('UNFUSED value', j['value'], j['ms_per_step'], 'glue', j['roofline']['glue_ms'], [ (k[:30],v['ms']) for k,v in j['roofline']['kernel_groups'].items() if 'heads' in k])"
python bench.py --dtype f32x3 --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; j=json.load(sys.stdin); print('f32x3 value', j['value'], j['ms_per_step'])"
python bench.py --dtype f16 --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; j=json.load(sys.stdin); print('f16 value', j['value'], j['ms_per_step'])"
