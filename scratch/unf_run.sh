set -x
python -m pytest tests/test_unfilter_gpu.py -x -q -m gpu 2>&1 | tail -8
python -m pytest tests/test_dataset.py tests/test_prep_gpu.py tests/test_fsrnet.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ucb > gpurun_out/r6v_loop_ucb_$i.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ffhq > gpurun_out/r6v_loop_ffhq_$i.json 2>/dev/null
BSR_DEVICE_UNFILTER=0 python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ucb > gpurun_out/r6v0_loop_ucb_$i.json 2>/dev/null
BSR_DEVICE_UNFILTER=0 python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ffhq > gpurun_out/r6v0_loop_ffhq_$i.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6v*_loop_*.json')):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        lp=l.get('loop',l)
        print(f, {k:(v.get('images_per_sec'), v.get('split_s',{}).get('prep_wait_s')) for k,v in lp.items() if isinstance(v,dict) and 'images_per_sec' in v and k.startswith('device')})
    except Exception as e: print(f, e)
P
