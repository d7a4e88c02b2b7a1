python -m pytest tests/test_prep_gpu.py tests/test_fsrnet.py tests/test_dataset.py -x -q -m gpu 2>&1 | tail -3
for cfgs in "ffhq 16 12 0 2 0 3000" "ffhq 24 16 0 2 0 3000" "ucb 12 4 32 4 0 2000" "ucb 16 4 48 4 0 2000" "ucb 24 4 64 6 0 2000"; do
  python scratch/loop_tune.py $cfgs 2>/dev/null | tail -1
done
