for n in bench_igemm_v bench_igemm_vDBSR_NO_STAGE bench_igemm_vDBSR_NO_STAGE_W bench_igemm_vDBSR_NO_STAGEDBSR_NO_STAGE_W; do
  echo "== $n solo"; BSR_ITERS=100 BSR_SOLO=1 ./scratch/$n 0 u 2>&1 | head -2
  echo "== $n pair"; BSR_ITERS=100 ./scratch/$n 0 u 2>&1 | head -2
done
