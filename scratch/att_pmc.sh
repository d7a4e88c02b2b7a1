#!/bin/bash
# Counter passes over ONE attention variant on the GPU box: bash scratch/att_pmc.sh scratch/libatt_a_base.so 2
lib=$1; dt=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L > gpurun_out/att_counters_avail.txt 2>&1
pick() { local out=""; for c in "$@"; do grep -qw "$c" gpurun_out/att_counters_avail.txt && out="$out $c"; done; echo $out; }
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  ctr=$(pick $grp)
  [ -z "$ctr" ] && continue
  rm -rf gpurun_out/att_pmc_$i
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/att_pmc_$i -- python3 scratch/att_only.py $lib $dt 6 > gpurun_out/att_pmc_$i.log 2>&1
  echo "pass $i ($ctr) rc=$?"
done
python3 - <<'PY'
import glob, pandas as pd
tot = {}
for d in sorted(glob.glob("gpurun_out/att_pmc_*/")):
    for f in glob.glob(d + "*/*counter_collection.csv"):
        t = pd.read_csv(f)
        t = t[t["Kernel_Name"].str.contains("attention")]
        last = t[t["Dispatch_Id"] >= t["Dispatch_Id"].max() - 2]          # the last three launches
        for c, v in last.groupby("Counter_Name")["Counter_Value"].sum().items():
            tot[c] = v / 3.0
for k in sorted(tot): print("%-32s %16.0f" % (k, tot[k]))
PY
