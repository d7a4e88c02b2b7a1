python -m pytest tests/test_gpu_png.py -x -q -m gpu 2>&1 | tail -4
bash scratch/png_prof.sh 2>&1 | tail -8
