# Round 5, ONE gpurun call: the evidence pass, the counter summaries (computed on the box so that the bench lines that follow can quote
# them for the kernel sources they were taken on), then the bench lines / loops.  Outputs: gpurun_out/r5_*, gpurun_out/profiles_r5/.
bash scratch/final_pass_r5.sh > gpurun_out/r5_final_pass.log 2>&1
for d in f32 f32x3 f16; do python tools/pmc_traffic.py r5 $d > /dev/null 2>&1; python tools/pmc_mfma.py r5 $d > /dev/null 2>&1; done
BSR_SKIP_TESTS=1 bash scratch/bench_only_r5.sh > gpurun_out/r5_bench_only.log 2>&1
mkdir -p gpurun_out/profiles_r5 && cp profiles/r5_pmc_* gpurun_out/profiles_r5/
cat gpurun_out/r5_final_tests.log
