"""Refresh profiles/<tag>_kernel_stats[_<dtype>].csv, profiles/<tag>_bench_n1[_<dtype>].json and the per-kernel table in
profiles/README.md from a gpurun_out/ run of:
    python bench.py [--dtype D] > gpurun_out/<bench>.json
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/<prof> -- python3 bench.py [--dtype D] --no-cpu-baseline --no-secondary --repeats 1
(the profiled command times 1 region of K steps: forwards = K + warm-up + the 3 event-timed ones of the roofline leg).
Usage: python tools/update_profiles.py r2 gpurun_out/r2_bench.json gpurun_out/r2_prof [dtype]"""
import glob
import json
import os
import re
import shutil
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, bench_json, prof_dir = sys.argv[1], sys.argv[2], sys.argv[3]
dtype = sys.argv[4] if len(sys.argv) > 4 else "f32"
sfx = "" if dtype == "f32" else "_" + dtype
stats = max(glob.glob(os.path.join(ROOT, prof_dir, "*", "*kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(stats, os.path.join(ROOT, "profiles", "%s_kernel_stats%s.csv" % (tag, sfx)))
shutil.copy(os.path.join(ROOT, bench_json), os.path.join(ROOT, "profiles", "%s_bench_n1%s.json" % (tag, sfx)))
b = json.load(open(os.path.join(ROOT, bench_json)))
nfwd = b["steps"] + b["warmup"] + 3
d = pd.read_csv(stats)
d = d[d["Name"].str.contains("bsr::")].copy()
d["kernel"] = (d["Name"].str.replace("void bsr::", "").str.replace("bsr::", "").str.replace(r"\((Conv|ConvN16|Stem)Args.*\)$", "", regex=True)
               .str.replace(r"\(float const\*.*", "", regex=True))
d["launches/fwd"] = (d["Calls"] / nfwd).round(2)
d["avg us"] = (d["AverageNs"] / 1000).round(1)
d["us/fwd"] = (d["TotalDurationNs"] / nfwd / 1000).round(1)
rows = ["| kernel | launches / fwd | avg µs | µs / fwd |", "|---|---|---|---|"]
for _, r in d.sort_values("us/fwd", ascending=False).iterrows():
    rows.append("| `%s` | %g | %.1f | %.1f |" % (r["kernel"], r["launches/fwd"], r["avg us"], r["us/fwd"]))
rows.append("| **sum** | %g | | **%.0f** |" % (d["launches/fwd"].sum(), d["us/fwd"].sum()))
rf = b["roofline"]
cb = b.get("cpu_baseline") or {}
head = ("* `%s_bench_n1%s.json` — `python bench.py%s` (N = 1, %d steps, %d warm-up): **%.0f images/s**%s, %.3f ms per %d-image step "
        "(repeats: min %.3f / median %.3f ms); dominant kernel `%s`: **%.1f TFLOP/s = %.1f %% of its %.0f TFLOP/s matrix peak** "
        "(avg launch %.4f ms — compare the rocprofv3 average below; algorithmic HBM rate %.0f GB/s = %.1f %% of 8 TB/s; nearer roof: %s), whole 3x3-conv path %.1f TFLOP/s (%.1f %%), "
        "all kernels %.1f TFLOP/s; CPU oracle %s images/s on %s host threads.\n"
        "* `%s_kernel_stats%s.csv` — `rocprofv3 --kernel-trace --stats -- python3 bench.py --streams 1%s --no-cpu-baseline --no-secondary --repeats 1` (%d forwards, one at a time: with two in flight a traced duration would include the time a kernel shares the chip). Per forward:\n\n"
        % (tag, sfx, "" if dtype == "f32" else " --dtype " + dtype, b["steps"], b["warmup"], b["value"],
           ((" with two forwards in flight (`single_stream`, one at a time: %.0f)" % b["single_stream"]["value"]) if b.get("single_stream") else
            (" one forward at a time (`two_in_flight`: %.0f)" % b["two_in_flight"]["value"]) if b.get("two_in_flight") else ""),
           b["ms_per_step"], b["config"]["images_per_gpu_per_step"],
           b["repeats"]["ms_per_step_min"], b["repeats"]["ms_per_step_median"], rf["kernel"].split(" — ")[0], rf["mfma_view"]["achieved_TFLOPs"],
           100 * rf["mfma_view"]["frac"], rf["mfma_view"]["peak_TFLOPs"],
           rf["avg_launch_ms"], rf["hbm_view"]["alg_GBps"], 100 * rf["hbm_view"]["frac"], rf["bound"], rf["path_3x3"]["achieved"], 100 * rf["path_3x3"]["frac"], rf["all_kernels_tflops"],
           cb.get("value"), cb.get("cores"), tag, sfx, "" if dtype == "f32" else " --dtype " + dtype, nfwd))
block = "<!-- BEGIN %s%s TABLE -->\n%s%s\n<!-- END %s%s TABLE -->" % (tag, sfx, head, "\n".join(rows), tag, sfx)
readme = os.path.join(ROOT, "profiles", "README.md")
txt = open(readme).read()
pat = re.compile(r"<!-- BEGIN %s%s TABLE -->.*?<!-- END %s%s TABLE -->" % (tag, sfx, tag, sfx), re.S)
txt = pat.sub(lambda m: block, txt) if pat.search(txt) else txt + "\n" + block + "\n"
open(readme, "w").write(txt)
print(block)
