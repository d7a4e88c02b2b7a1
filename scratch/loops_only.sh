# The two loops (twice each) + the stage table, nothing else: after a change on the HOST side of the loops (loaders, file writers).
set -x
T=${1:-r5x}
python -m pytest tests/test_fsrnet.py tests/test_dataset.py tests/test_pngio.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ffhq > gpurun_out/${T}_loop_ffhq_$i.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ucb > gpurun_out/${T}_loop_ucb_$i.json 2>/dev/null
done
python tools/loop_stage_table.py --out gpurun_out/${T}_loop_stage_table.json > /dev/null 2>&1
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/%s_loop_*_[12].json' % "${T}")):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, {k:(v.get('items_per_sec') if isinstance(v,dict) else v) for k,v in l.get('loop',l).items() if isinstance(v,dict)})
    except Exception as e: print(f, e)
P
