# After `gpurun -- bash scratch/final_all_r6.sh`: copy what is judged from gpurun_out/ into profiles/ and regenerate the documents.
set -e
cd "$(dirname "$0")/.."
cp gpurun_out/profiles_r6/r6_pmc_* profiles/
python tools/update_profiles.py r6 gpurun_out/r6_bench.json gpurun_out/r6_prof > /dev/null
python tools/update_profiles.py r6 gpurun_out/r6_bench_f32x3.json gpurun_out/r6_prof_f32x3 f32x3 > /dev/null
python tools/update_profiles.py r6 gpurun_out/r6_bench_f16.json gpurun_out/r6_prof_f16 f16 > /dev/null
for f in dist1 tsm512 tsm512_f32x3 b16 n2_one_gpu; do cp gpurun_out/r6_bench_$f.json profiles/r6_bench_$f.json; done
cp gpurun_out/r6_loop_ffhq.json profiles/r6_loop_ffhq_1.json; cp gpurun_out/r6_loop_ffhq_2.json profiles/r6_loop_ffhq_2.json
cp gpurun_out/r6_loop_ucb.json profiles/r6_loop_ucb_1.json; cp gpurun_out/r6_loop_ucb_2.json profiles/r6_loop_ucb_2.json
cp gpurun_out/r6_loop_stage_table.json gpurun_out/r6_batch_sweep.json gpurun_out/r6_lane_overlap.txt profiles/
cp gpurun_out/f16_margins.txt profiles/r6_f16_margins.txt
for w in tsm512 b16; do cp "$(ls -t gpurun_out/r6_prof_$w/*/*kernel_stats.csv | head -1)" profiles/r6_kernel_stats_$w.csv; done
python tools/round_readme.py r6 > /dev/null
git status --short | head -40
