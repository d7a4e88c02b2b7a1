set -x
python -m pytest tests/test_ucb_post_gpu.py -x -q -m gpu 2>&1 | tail -5
python scratch/ucb_time.py 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ucb_prof -o ucb -- python3 $GRAFT_REPO_ROOT/scratch/ucb_time.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv,glob
for f in glob.glob('gpurun_out/ucb_prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:24]: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
P

python -m pytest tests/test_fsrnet.py tests/test_dataset.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ucb > gpurun_out/r6u_loop_ucb_$i.json 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline --no-secondary --loop ffhq > gpurun_out/r6u_loop_ffhq_$i.json 2>/dev/null
done
tail -c 1500 gpurun_out/r6u_loop_ucb_1.json; tail -c 600 gpurun_out/r6u_loop_ucb_2.json; tail -c 600 gpurun_out/r6u_loop_ffhq_1.json
