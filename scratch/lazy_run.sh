python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f16 or f32x3 or 16_bit or conv1 or split" 2>&1 | tail -4
python scratch/layer_times.py f32x3 | grep -E "launches|conv1"
python scratch/layer_times.py f16 | grep -E "launches|conv1"
