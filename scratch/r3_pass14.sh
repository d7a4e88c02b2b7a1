python -m pytest tests/test_dataset.py tests/test_fsrnet.py tests/test_ucb_post.py tests/test_metrics.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ucb 2>/dev/null > gpurun_out/r3_loop_ucb.json; python -c "
import json; j=json.load(open('gpurun_out/r3_loop_ucb.json'))['loop']
print({m:(v['images_per_sec'], v.get('steady_images_per_sec')) for m,v in j.items() if isinstance(v,dict)})"; done
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ffhq 2>/dev/null > gpurun_out/r3_loop_ffhq.json; python -c "
import json; j=json.load(open('gpurun_out/r3_loop_ffhq.json'))['loop']
print({m:(v['images_per_sec'], v.get('steady_images_per_sec')) for m,v in j.items() if isinstance(v,dict)})"
