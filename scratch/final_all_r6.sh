# Round 6, ONE gpurun call: the evidence pass, the counter summaries (computed on the box so that the bench lines that follow can quote
# them for the kernel sources they were taken on), then the bench lines / loops.  Outputs: gpurun_out/r6_*, gpurun_out/profiles_r6/.
bash scratch/final_pass_r6.sh > gpurun_out/r6_final_pass.log 2>&1
for d in f32 f32x3 f16; do python tools/pmc_traffic.py r6 $d > /dev/null 2>&1; python tools/pmc_mfma.py r6 $d > /dev/null 2>&1; done
BSR_SKIP_TESTS=1 bash scratch/bench_only_r6.sh > gpurun_out/r6_bench_only.log 2>&1
mkdir -p gpurun_out/profiles_r6 && cp profiles/r6_pmc_* gpurun_out/profiles_r6/
cat gpurun_out/r6_final_tests.log
