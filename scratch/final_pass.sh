set -x
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in f32 f32x3 f16; do
  sfx=""; [ $d != f32 ] && sfx="_$d"
  for c in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
    n=${c%%:*}; ctr=${c#*:}
    rm -rf gpurun_out/r2_pmc_${n}${sfx}
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/r2_pmc_${n}${sfx} -- python3 scratch/run_fwd.py 32 2 $d > gpurun_out/r2_pmc_${n}${sfx}.log 2>&1
  done
  rm -rf gpurun_out/r2_prof${sfx}
  if [ $d = f32 ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_prof -- python3 bench.py --no-cpu-baseline --no-secondary --repeats 1 > gpurun_out/r2_prof.log 2>&1
  else
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_prof${sfx} -- python3 bench.py --dtype $d --no-cpu-baseline --no-secondary --repeats 1 > gpurun_out/r2_prof${sfx}.log 2>&1
  fi
done
python bench.py --workload tsm512 --no-cpu-baseline --no-secondary > gpurun_out/r2_bench_tsm512.json 2>/dev/null
python bench.py --workload tsm512 --dtype f32x3 --no-cpu-baseline --no-secondary > gpurun_out/r2_bench_tsm512_f32x3.json 2>/dev/null
ls gpurun_out | grep r2_pmc | head -30
