python -m pytest tests/test_fsrnet.py -x -q -m gpu 2>&1 | tail -3
for cfgs in "ffhq 12 8 0 2 0 2000" "ffhq 16 12 0 2 0 2000" "ffhq 24 16 0 2 0 2000" "ucb 12 4 32 4 0 1200" "ucb 16 4 48 4 0 1200"; do
  python scratch/loop_tune.py $cfgs 2>/dev/null | tail -1
done
