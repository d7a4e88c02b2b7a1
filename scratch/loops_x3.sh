for i in 1 2; do
python bench.py --dtype f32x3 --steps 5 --no-cpu-baseline --no-secondary --no-sustained --loop ucb > gpurun_out/r6x3_loop_ucb_$i.json 2>/dev/null
python bench.py --dtype f32x3 --steps 5 --no-cpu-baseline --no-secondary --no-sustained --loop ffhq > gpurun_out/r6x3_loop_ffhq_$i.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6x3_loop_*.json')):
    l=json.loads(open(f).read().strip().splitlines()[-1]); lp=l.get('loop',l)
    print(f, lp.get('dtype'), {k:(v.get('images_per_sec'), v.get('split_s',{}).get('prep_wait_s')) for k,v in lp.items() if isinstance(v,dict) and 'images_per_sec' in v and k.startswith('device_p')})
P
