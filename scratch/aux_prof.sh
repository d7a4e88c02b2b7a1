cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/ucb_prof $GRAFT_REPO_ROOT/gpurun_out/png_prof $GRAFT_REPO_ROOT/gpurun_out/unf_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ucb_prof -o ucb -- python3 $GRAFT_REPO_ROOT/scratch/ucb_time.py > $GRAFT_REPO_ROOT/gpurun_out/ucb_time.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/png_prof -o png -- python3 $GRAFT_REPO_ROOT/scratch/png_time.py > $GRAFT_REPO_ROOT/gpurun_out/png_time.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/unf_prof -o unf -- python3 $GRAFT_REPO_ROOT/scratch/unf_time.py > $GRAFT_REPO_ROOT/gpurun_out/unf_time.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/ucb_time.txt; tail -2 gpurun_out/png_time.txt; tail -2 gpurun_out/unf_time.txt
