timeout 120 python -m pytest tests/test_unfilter_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 60 python scratch/unf_time.py 2>&1 | tail -2
