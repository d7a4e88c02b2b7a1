# kernel time of one loop under rocprofv3 (which kernels the GPU spends the loop's wall time in): bash scratch/loop_kernels.sh ffhq 4000 12
cd /tmp && export TMPDIR=/tmp
K=${1:-ffhq}; N=${2:-4000}; W=${3:-12}
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/loopk_$K -o loopk -- python3 $GRAFT_REPO_ROOT/scratch/loop_workers_sweep.py $K $N $W > $GRAFT_REPO_ROOT/gpurun_out/loopk_$K.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/loopk_$K/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls gpurun_out/loopk_$K/*kernel_stats.csv | head -1)
head -25 $f | cut -c1-220 > gpurun_out/loopk_${K}_stats.txt
find gpurun_out/loopk_$K -name "*.csv" -size +1M -delete; find gpurun_out/loopk_$K -name "*.db" -delete
