set -x
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3_t1.log
cat gpurun_out/r3_t1.log
python bench.py > gpurun_out/r3_bench0.json 2> gpurun_out/r3_bench0.err
tail -c 600 gpurun_out/r3_bench0.json
bash tools/pmc_mfma_pass.sh r3 f32 f32x3
ls gpurun_out | grep r3_
