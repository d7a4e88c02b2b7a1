python -m pytest tests/test_prep_gpu.py tests/test_fsrnet.py tests/test_dataset.py -x -q -m gpu 2>&1 | tail -12
python bench.py --no-cpu-baseline --no-secondary --steps 5 --loop ffhq 2>gpurun_out/r3_loop_ffhq.err > gpurun_out/r3_loop_ffhq.json; python -c "
import json; j=json.load(open('gpurun_out/r3_loop_ffhq.json'))['loop']
for k,v in j.items(): print(k, v)"
python bench.py --no-cpu-baseline --no-secondary --steps 5 --loop ucb 2>gpurun_out/r3_loop_ucb.err > gpurun_out/r3_loop_ucb.json; python -c "
import json; j=json.load(open('gpurun_out/r3_loop_ucb.json'))['loop']
for k,v in j.items(): print(k, v)"
tail -3 gpurun_out/r3_loop_ucb.err
