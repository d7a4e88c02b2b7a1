# The bench lines of final_pass_r6.sh alone — re-run after tools/pmc_traffic.py / pmc_mfma.py refreshed profiles/r6_pmc_*.json (so that the
# lines carry `traffic` / `mfma_busy` for the current kernel sources) or after bench.py changed.  BSR_SKIP_TESTS=1 skips the pytest leg.
set -x
T=r6
[ -n "$BSR_SKIP_TESTS" ] || { python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/${T}_final_tests.log; cat gpurun_out/${T}_final_tests.log; }
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --dtype f32x3 > gpurun_out/${T}_bench_f32x3.json 2>/dev/null
python bench.py --dtype f16 > gpurun_out/${T}_bench_f16.json 2>/dev/null
BSR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_dist1.json 2>/dev/null
python bench.py --workload tsm512 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_tsm512.json 2>/dev/null
# two ranks sharing ONE GPU (gloo control plane, peer-copy gather): the N = 2 data path exercised for real on the 1-GPU box
python bench.py --gpus 2 --gather peer --device 0 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_n2_one_gpu.json 2> gpurun_out/${T}_bench_n2_one_gpu.err
python bench.py --batch 16 --streams 1 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_b16.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ffhq > gpurun_out/${T}_loop_ffhq.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ucb > gpurun_out/${T}_loop_ucb.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ffhq > gpurun_out/${T}_loop_ffhq_2.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ucb > gpurun_out/${T}_loop_ucb_2.json 2>/dev/null
python tools/loop_stage_table.py --out gpurun_out/${T}_loop_stage_table.json > /dev/null 2>&1
