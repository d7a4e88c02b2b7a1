cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_heads or forward_matches or edge_inputs" 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-secondary --steps 30 2>/dev/null | python -c "
import json,sys; j=json.load(sys.stdin); print('value', j['value'], j['ms_per_step']); [print('  ', k[:60], v['ms'], v['frac']) for k,v in j['roofline']['kernel_groups'].items() if 'heads' in k or 'clr_conv1' in k]"
rm -rf gpurun_out/r3_pmc_fetch_t; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r3_pmc_fetch_t -- python3 scratch/run_fwd.py 32 2 f32 > /dev/null 2>&1
python - <<'PY'
import glob, pandas as pd
c = pd.read_csv(max(glob.glob("gpurun_out/r3_pmc_fetch_t/*/*counter_collection.csv")))
c = c[c["Kernel_Name"].str.contains("n16")]
print((c.groupby("Kernel_Name")["Counter_Value"].mean() * 2 * 1024 / 1e6).to_string())
PY
