# The bench lines of final_pass.sh alone: re-run after tools/pmc_traffic.py / pmc_mfma.py refreshed profiles/r3_pmc_*.json, so that the lines carry `traffic` / `mfma_busy` for the current kernel sources.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
python bench.py --dtype f32x3 > gpurun_out/r3_bench_f32x3.json 2>/dev/null
python bench.py --dtype f16 > gpurun_out/r3_bench_f16.json 2>/dev/null
BSR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r3_bench_dist1.json 2>/dev/null
python bench.py --workload tsm512 --no-cpu-baseline --no-secondary > gpurun_out/r3_bench_tsm512.json 2>/dev/null
python bench.py --workload tsm512 --dtype f32x3 --no-cpu-baseline --no-secondary > gpurun_out/r3_bench_tsm512_f32x3.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ffhq > gpurun_out/r3_loop_ffhq.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ucb > gpurun_out/r3_loop_ucb.json 2>/dev/null
