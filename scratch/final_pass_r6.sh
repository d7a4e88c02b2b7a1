# Round-6 evidence pass (run on the GPU box through gpurun, from the repo root): tests, benches, rocprofv3 traces and counter passes.
# BSR_SKIP_TESTS=1 skips the pytest leg, BSR_SKIP_PMC=1 the counter passes.
set -x
T=r6
[ -n "$BSR_SKIP_TESTS" ] || { python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/${T}_final_tests.log; cat gpurun_out/${T}_final_tests.log; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --dtype f32x3 > gpurun_out/${T}_bench_f32x3.json 2>/dev/null
python bench.py --dtype f16 > gpurun_out/${T}_bench_f16.json 2>/dev/null
BSR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_dist1.json 2>/dev/null
python bench.py --workload tsm512 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_tsm512.json 2>/dev/null
python bench.py --workload tsm512 --dtype f32x3 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_tsm512_f32x3.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ffhq > gpurun_out/${T}_loop_ffhq.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ucb > gpurun_out/${T}_loop_ucb.json 2>/dev/null
python tools/batch_sweep.py --out gpurun_out/${T}_batch_sweep.json > /dev/null 2>&1
python tools/loop_stage_table.py --out gpurun_out/${T}_loop_stage_table.json > /dev/null 2>&1
# the kernel-trace summaries are taken with --streams 1: with two forwards in flight a kernel's traced duration includes the time it shares the chip
for d in f32 f32x3 f16; do
  sfx=""; [ $d != f32 ] && sfx="_$d"
  [ -n "$BSR_SKIP_PMC" ] || for c in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
    n=${c%%:*}; ctr=${c#*:}
    rm -rf gpurun_out/${T}_pmc_${n}${sfx}
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/${T}_pmc_${n}${sfx} -- python3 scratch/run_fwd.py 32 2 $d > gpurun_out/${T}_pmc_${n}${sfx}.log 2>&1
  done
  rm -rf gpurun_out/${T}_prof${sfx}
  if [ $d = f32 ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -- python3 bench.py --streams 1 --no-cpu-baseline --no-secondary --no-sustained --repeats 1 > gpurun_out/${T}_prof.log 2>&1
  else
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof${sfx} -- python3 bench.py --streams 1 --dtype $d --no-cpu-baseline --no-secondary --no-sustained --repeats 1 > gpurun_out/${T}_prof${sfx}.log 2>&1
  fi
done
# configs[4] (TSM, 512x512 frames) and configs[2]'s batch (B = 16): kernel-trace summaries of their own (round 5)
rm -rf gpurun_out/${T}_prof_tsm512 gpurun_out/${T}_prof_b16
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_tsm512 -- python3 bench.py --workload tsm512 --streams 1 --no-cpu-baseline --no-secondary --no-sustained --repeats 1 > gpurun_out/${T}_prof_tsm512.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_b16 -- python3 bench.py --batch 16 --streams 1 --no-cpu-baseline --no-secondary --no-sustained --repeats 1 > gpurun_out/${T}_prof_b16.log 2>&1
python bench.py --batch 16 --streams 1 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_b16.json 2>/dev/null
# two forwards in flight, the profiler's view (tools/lane_overlap.py)
rm -rf gpurun_out/${T}_prof_lanes
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_prof_lanes -- python3 bench.py --no-cpu-baseline --no-secondary --no-sustained --repeats 1 > gpurun_out/${T}_prof_lanes.log 2>&1
python tools/lane_overlap.py gpurun_out/${T}_prof_lanes > gpurun_out/${T}_lane_overlap.txt 2>&1
[ -n "$BSR_SKIP_PMC" ] || bash tools/pmc_mfma_pass.sh ${T} f32 f32x3 f16
ls gpurun_out | grep ${T}_ | wc -l
