for i in 1 2 3; do
BSR_DEVICE_UNFILTER=1 python bench.py --steps 5 --no-cpu-baseline --no-secondary --no-sustained --loop ffhq > gpurun_out/ffab1_$i.json 2>/dev/null
BSR_DEVICE_UNFILTER=0 python bench.py --steps 5 --no-cpu-baseline --no-secondary --no-sustained --loop ffhq > gpurun_out/ffab0_$i.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/ffab*_*.json')):
    l=json.loads(open(f).read().strip().splitlines()[-1]); lp=l.get('loop',l)
    print(f, {k:(v.get('images_per_sec'), v.get('split_s',{}).get('prep_wait_s')) for k,v in lp.items() if isinstance(v,dict) and 'images_per_sec' in v and k.startswith('device_png')})
P
