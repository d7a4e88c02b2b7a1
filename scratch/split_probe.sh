for d in f32 f32x3 f16; do
  python bench.py --batch 16 --dtype $d --no-cpu-baseline --no-secondary --no-sustained > gpurun_out/split_$d.json 2>/dev/null
  python - $d <<'P'
import json,sys
d=json.loads(open('gpurun_out/split_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
t=d.get('two_in_flight')
print(sys.argv[1], 'B=16 one at a time', d['value'], ' two lanes of 16 in flight', t.get('value') if isinstance(t,dict) else t)
P
done
