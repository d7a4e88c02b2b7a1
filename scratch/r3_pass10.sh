python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ffhq 2>/dev/null > gpurun_out/r3_loop_ffhq.json; python -c "
import json; j=json.load(open('gpurun_out/r3_loop_ffhq.json'))['loop']
for k,v in j.items(): print(k, v)"
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ucb 2>/dev/null > gpurun_out/r3_loop_ucb.json; python -c "
import json; j=json.load(open('gpurun_out/r3_loop_ucb.json'))['loop']
for k,v in j.items(): print(k, v)"
ls /dev/shm | head -3
