# kernel timeline of the FFHQ loop WITHOUT loader workers (elements served from HBM: scratch/loop_ablate.py) — rocprofv3 cannot wrap a loop with
# worker processes.  Prints, per kernel family, time per batch; and the gaps on the compute stream.
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/loop_trace
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/loop_trace -- python3 $GRAFT_REPO_ROOT/scratch/loop_ablate.py ${1:-ffhq} ${2:-2000} > $GRAFT_REPO_ROOT/gpurun_out/loop_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import glob, pandas as pd
f = glob.glob("gpurun_out/loop_trace/*/*kernel_trace.csv")[0]
t = pd.read_csv(f).sort_values("Start_Timestamp")
t["dur"] = t["End_Timestamp"] - t["Start_Timestamp"]
# the last third of the run = steady state of the last repetition
lo = t["Start_Timestamp"].quantile(0.70); hi = t["Start_Timestamp"].quantile(0.98)
s = t[(t["Start_Timestamp"] >= lo) & (t["Start_Timestamp"] <= hi)].copy()
span = (s["End_Timestamp"].max() - s["Start_Timestamp"].min()) / 1e6
name = s["Kernel_Name"].str.replace("void ", "").str.slice(0, 60)
g = s.groupby(name)["dur"].agg(["sum", "count"]).sort_values("sum", ascending=False)
nfw = int(s["Kernel_Name"].str.contains("stem7").sum())
print("window %.1f ms, %d forwards -> %.3f ms per batch; kernel time %.1f ms (%.1f %% of the window)" % (span, nfw, span / max(nfw, 1), g["sum"].sum() / 1e6, 100 * g["sum"].sum() / 1e6 / span))
g["ms_per_batch"] = g["sum"] / 1e6 / max(nfw, 1)
g["per_batch"] = g["count"] / max(nfw, 1)
print(g[["ms_per_batch", "per_batch"]].head(28).to_string())
# union of busy intervals (kernels may overlap across streams)
iv = s[["Start_Timestamp", "End_Timestamp"]].values
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for a, b in iv[1:]:
    if a > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = a, b
    else:
        cur_e = max(cur_e, b)
busy += cur_e - cur_s
print("GPU busy (union of kernel intervals) %.1f %% of the window" % (100 * busy / 1e6 / span))
mc = glob.glob("gpurun_out/loop_trace/*/*memory_copy_trace.csv")
if mc:
    m = pd.read_csv(mc[0]); m = m[(m["Start_Timestamp"] >= lo) & (m["Start_Timestamp"] <= hi)]
    m["dur"] = m["End_Timestamp"] - m["Start_Timestamp"]
    print(m.groupby("Direction")["dur"].agg(["sum", "count"]).assign(ms_per_batch=lambda d: d["sum"] / 1e6 / max(nfw, 1)).to_string())
PY
find gpurun_out/loop_trace -name "*.csv" -size +3M -delete
