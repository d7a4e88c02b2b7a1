set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BSR_ITERS=1500 ./scratch/bench_igemm 0 u > gpurun_out/r3_clock_stamps.txt 2>&1
cat gpurun_out/r3_clock_stamps.txt
rm -rf gpurun_out/r3_clock_pmc
BSR_ITERS=12 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/r3_clock_pmc -- ./scratch/bench_igemm 0 u > gpurun_out/r3_clock_pmc.txt 2>&1
tail -5 gpurun_out/r3_clock_pmc.txt
python - <<'PY'
import glob, pandas as pd
f = max(glob.glob("gpurun_out/r3_clock_pmc/*/*counter_collection.csv"))
c = pd.read_csv(f)
c = c[c["Kernel_Name"].str.contains("igemm_conv_kernel")]
c["dur"] = c["End_Timestamp"] - c["Start_Timestamp"]
w = c.pivot_table(index=["Dispatch_Id", "Grid_Size", "dur"], columns="Counter_Name", values="Counter_Value").reset_index()
w["clock_gui"] = w["GRBM_GUI_ACTIVE"] / 8 / w["dur"]
w["clock_sqbusy32"] = w["SQ_BUSY_CYCLES"] / 32 / w["dur"]
w["mfma_busy_nominal"] = w["SQ_VALU_MFMA_BUSY_CYCLES"] / (w["dur"] * 2.4 * 1024)
w["wave_cyc_per_slot_GHz"] = w["SQ_WAVE_CYCLES"] * 4 / 2048 / w["dur"]
print(w.groupby("Grid_Size")[["dur", "clock_gui", "clock_sqbusy32", "mfma_busy_nominal", "wave_cyc_per_slot_GHz"]].median().to_string())
PY
