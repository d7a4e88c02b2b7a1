# Round-3 evidence pass (run on the GPU box through gpurun, from the repo root): tests, benches, rocprofv3 traces and counter passes.
set -x
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3_final_tests.log; cat gpurun_out/r3_final_tests.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
python bench.py --dtype f32x3 > gpurun_out/r3_bench_f32x3.json 2>/dev/null
python bench.py --dtype f16 > gpurun_out/r3_bench_f16.json 2>/dev/null
BSR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r3_bench_dist1.json 2>/dev/null
python bench.py --workload tsm512 --no-cpu-baseline --no-secondary > gpurun_out/r3_bench_tsm512.json 2>/dev/null
python bench.py --workload tsm512 --dtype f32x3 --no-cpu-baseline --no-secondary > gpurun_out/r3_bench_tsm512_f32x3.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ffhq > gpurun_out/r3_loop_ffhq.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ucb > gpurun_out/r3_loop_ucb.json 2>/dev/null
# the kernel-trace summaries are taken with --streams 1: with two forwards in flight a kernel's traced duration includes the time it shares the chip
for d in f32 f32x3 f16; do
  sfx=""; [ $d != f32 ] && sfx="_$d"
  [ -n "$BSR_SKIP_PMC" ] || for c in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
    n=${c%%:*}; ctr=${c#*:}
    rm -rf gpurun_out/r3_pmc_${n}${sfx}
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/r3_pmc_${n}${sfx} -- python3 scratch/run_fwd.py 32 2 $d > gpurun_out/r3_pmc_${n}${sfx}.log 2>&1
  done
  rm -rf gpurun_out/r3_prof${sfx}
  if [ $d = f32 ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof -- python3 bench.py --streams 1 --no-cpu-baseline --no-secondary --repeats 1 > gpurun_out/r3_prof.log 2>&1
  else
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof${sfx} -- python3 bench.py --streams 1 --dtype $d --no-cpu-baseline --no-secondary --repeats 1 > gpurun_out/r3_prof${sfx}.log 2>&1
  fi
done
[ -n "$BSR_SKIP_PMC" ] || bash tools/pmc_mfma_pass.sh r3 f32 f32x3 f16
[ -n "$BSR_SKIP_PMC" ] && exit 0
BSR_ITERS=1500 ./scratch/bench_igemm 0 u > gpurun_out/r3_clock_stamps.txt 2>&1
BSR_ITERS=300 BSR_SOLO=1 ./scratch/bench_igemm 0 u > gpurun_out/r3_clock_stamps_solo.txt 2>&1
BSR_ITERS=100 BSR_TIMELINE=1 ./scratch/bench_igemm 0 u 2>&1 | head -4 > gpurun_out/r3_timeline.txt
BSR_ITERS=100 BSR_TIMELINE=1 BSR_SOLO=1 ./scratch/bench_igemm 0 u 2>&1 | head -4 >> gpurun_out/r3_timeline.txt
./scratch/coexec_probe > gpurun_out/r3_coexec.txt 2>&1
ls gpurun_out | grep r3_ | wc -l
