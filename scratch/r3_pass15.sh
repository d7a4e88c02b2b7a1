BSR_ITERS=100 BSR_TIMELINE=1 ./scratch/bench_igemm 0 u 2>&1 | head -3
BSR_ITERS=300 ./scratch/bench_igemm 0 u | grep -v lifetime
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward_matches or full_batch or rows_are or heights or wider or tsm_variant or edge" 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-secondary --steps 30 2>/dev/null | python -c "
import json,sys; j=json.load(sys.stdin); print('value', j['value'], j['ms_per_step'], j['repeats']['ms_per_step_all']); [print('  ', k[:70], v['ms'], v['frac']) for k,v in j['roofline']['kernel_groups'].items()]"
