#!/bin/bash
# Counter passes over WHOLE forwards, summarised per kernel (what each kernel's waves spend their cycles on): bash scratch/fwd_pmc.sh f16
dt=${1:-f16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L > gpurun_out/fwd_counters_avail.txt 2>&1
pick() { local out=""; for c in "$@"; do grep -qw "$c" gpurun_out/fwd_counters_avail.txt && out="$out $c"; done; echo $out; }
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  ctr=$(pick $grp)
  [ -z "$ctr" ] && continue
  rm -rf gpurun_out/fwd_pmc_${dt}_$i
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/fwd_pmc_${dt}_$i -- python3 scratch/run_fwd.py 32 2 $dt > gpurun_out/fwd_pmc_${dt}_$i.log 2>&1
  echo "pass $i ($ctr) rc=$?"
done
python3 - "$dt" <<'PY'
import glob, sys, pandas as pd
dt = sys.argv[1]
frames = []
for d in sorted(glob.glob("gpurun_out/fwd_pmc_%s_*/" % dt)):
    for f in glob.glob(d + "*/*counter_collection.csv"):
        t = pd.read_csv(f)
        half = t["Dispatch_Id"].max() // 2
        t = t[t["Dispatch_Id"] > half]                       # the second forward
        frames.append(t.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].sum().reset_index())
a = pd.concat(frames).groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].mean().unstack()
a.index = [k.replace("void bsr::", "")[:70] for k in a.index]
pd.set_option("display.width", 400, "display.max_columns", 50, "display.max_colwidth", 72)
cols = [c for c in a.columns]
w = a["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in a else None
out = pd.DataFrame(index=a.index)
if "GRBM_GUI_ACTIVE" in a: out["gpu_kcyc"] = (a["GRBM_GUI_ACTIVE"] / 8 / 1e3).round(0)
for c in ("SQ_BUSY_CYCLES",):
    if c in a: out["busy_kcyc"] = (a[c] / 1e3 / 32).round(0)
def frac(n, d="SQ_WAVE_CYCLES"):
    return (a[n] / a[d]).round(3) if n in a and d in a else None
for name, c in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst", "SQ_WAIT_INST_ANY"), ("act_any", "SQ_ACTIVE_INST_ANY"), ("act_valu", "SQ_ACTIVE_INST_VALU"), ("act_lds", "SQ_ACTIVE_INST_LDS"),
                ("wait_lds", "SQ_WAIT_INST_LDS"), ("act_vmem", "SQ_ACTIVE_INST_VMEM"), ("wait_vmem", "SQ_WAIT_INST_VMEM"), ("act_sca", "SQ_ACTIVE_INST_SCA"), ("act_misc", "SQ_ACTIVE_INST_MISC")):
    f = frac(c)
    if f is not None: out[name] = f
if "SQ_LDS_BANK_CONFLICT" in a and "SQ_LDS_IDX_ACTIVE" in a: out["lds_conflict"] = (a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"]).round(3)
if "SQ_LDS_IDX_ACTIVE" in a and "GRBM_GUI_ACTIVE" in a: out["lds_idx_per_gpu_cyc"] = (a["SQ_LDS_IDX_ACTIVE"] / (a["GRBM_GUI_ACTIVE"] / 8) ).round(2)
if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "GRBM_GUI_ACTIVE" in a: out["mfma_busy"] = (a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8) / 1024).round(3)
for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"):
    if c in a: out[c.replace("SQ_INSTS_", "n_").lower()] = (a[c] / 1e6).round(2)
print(out.sort_values(out.columns[0], ascending=False).to_string())
out.to_csv("gpurun_out/fwd_pmc_%s_summary.csv" % dt)
a.to_csv("gpurun_out/fwd_pmc_%s_raw.csv" % dt)
PY
find gpurun_out/fwd_pmc_${dt}_* -name "*.csv" -size +2M -delete 2>/dev/null
