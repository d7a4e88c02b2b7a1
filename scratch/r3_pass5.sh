echo "== one tile per WG"; BSR_ITERS=200 BSR_ONE_TILE=1 ./scratch/bench_igemm 0 u 2>&1 | head -2
for d in 0 50 100 150; do echo "== persistent dephase $d"; BSR_ITERS=200 BSR_DEPHASE=$d ./scratch/bench_igemm 0 u 2>&1 | head -2; done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward_matches or full_batch or rows_are or wider or heights or edge" 2>&1 | tail -4
for d in 0 100; do BSR_DEPHASE=$d python bench.py --no-cpu-baseline --no-secondary --steps 30 2>/dev/null > gpurun_out/r3_pers_$d.json; python - <<PY
import json
j=json.load(open("gpurun_out/r3_pers_$d.json"))
print("PERSISTENT DEPHASE $d value", j["value"], "ms", j["ms_per_step"], "reps", j["repeats"]["ms_per_step_all"])
for k,v in j["roofline"]["kernel_groups"].items(): print("   %-100s %7.4f ms frac %.3f" % (k[:100], v["ms"], v["frac"]))
PY
done
BSR_PERSISTENT=0 python bench.py --no-cpu-baseline --no-secondary --steps 30 2>/dev/null | python -c "import json,sys; j=json.load(sys.stdin); print('NON-PERSISTENT value', j['value'], j['ms_per_step'])"
