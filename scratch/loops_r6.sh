for i in 1 2; do
python bench.py --steps 5 --no-cpu-baseline --no-secondary --no-sustained --loop ucb > gpurun_out/r6w_loop_ucb_$i.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --no-secondary --no-sustained --loop ffhq > gpurun_out/r6w_loop_ffhq_$i.json 2>/dev/null
done
python tools/loop_stage_table.py --out gpurun_out/r6_loop_stage_table.json > /dev/null 2>&1
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6w_loop_*.json')):
    l=json.loads(open(f).read().strip().splitlines()[-1]); lp=l.get('loop',l)
    print(f, {k:(v.get('images_per_sec'), v.get('split_s',{}).get('prep_wait_s')) for k,v in lp.items() if isinstance(v,dict) and 'images_per_sec' in v and k.startswith('device_p')})
st=json.load(open('gpurun_out/r6_loop_stage_table.json'))['stages']
for k,v in st.items():
    if k.startswith('loader_host'): print(k, v['items_per_sec'], v['job_cpu_ms_alone'])
P
