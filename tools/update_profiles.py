"""Refresh profiles/<tag>_kernel_stats.csv, <tag>_bench_n1.json and the per-kernel table in profiles/README.md from a
gpurun_out/ run of:
    python bench.py > gpurun_out/bench_<x>.json
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_<x> -- python3 bench.py --no-cpu-baseline
Usage: python tools/update_profiles.py r1 gpurun_out/bench_r1c.json gpurun_out/prof_r1c"""
import glob
import json
import os
import re
import shutil
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, bench_json, prof_dir = sys.argv[1], sys.argv[2], sys.argv[3]
stats = max(glob.glob(os.path.join(ROOT, prof_dir, "*", "*kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(stats, os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(ROOT, bench_json), os.path.join(ROOT, "profiles", tag + "_bench_n1.json"))
b = json.load(open(os.path.join(ROOT, bench_json)))
nfwd = b["steps"] + b["warmup"] + 3                     # + the 3 event-timed forwards of the roofline leg
d = pd.read_csv(stats)
d = d[d["Name"].str.contains("bsr::")]
d["kernel"] = (d["Name"].str.replace("void bsr::", "").str.replace("bsr::", "").str.replace("(bsr::ConvArgs)", "").str.replace("(ConvArgs)", "")
               .str.replace("(bsr::ConvN16Args)", "").str.replace("(ConvN16Args)", "").str.replace("(bsr::StemArgs)", "").str.replace("(StemArgs)", "")
               .str.replace(r"\(float const\*.*", "", regex=True))
d["launches/fwd"] = (d["Calls"] / nfwd).round(2)
d["avg us"] = (d["AverageNs"] / 1000).round(1)
d["us/fwd"] = (d["TotalDurationNs"] / nfwd / 1000).round(1)
rows = ["| kernel | launches / fwd | avg µs | µs / fwd |", "|---|---|---|---|"]
for _, r in d.sort_values("us/fwd", ascending=False).iterrows():
    rows.append("| `%s` | %g | %.1f | %.1f |" % (r["kernel"], r["launches/fwd"], r["avg us"], r["us/fwd"]))
rows.append("| **sum** | %g | | **%.0f** |" % (d["launches/fwd"].sum(), d["us/fwd"].sum()))
rf = b["roofline"]
head = ("* `%s_bench_n1.json` — `python bench.py` (N = 1, %d steps, %d warm-up): **%.0f images/s**, %.2f ms per 32-image forward; "
        "dominant kernel (`igemm_conv_kernel<3,3,1,true,…,2,32,1>`: up2, up3, clr_up3) **%.1f TFLOP/s = %.1f %% of the 157.3 TFLOP/s fp32 MFMA peak** "
        "(avg launch %.4f ms — compare the rocprofv3 average below), whole 3x3-conv path %.1f TFLOP/s (%.1f %%), all kernels %.1f TFLOP/s; CPU oracle %s images/s on %s host threads.\n"
        "* `%s_kernel_stats.csv` — `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline` (%d forwards). Per forward:\n\n"
        % (tag, b["steps"], b["warmup"], b["value"], b["ms_per_step"], rf["achieved"], 100 * rf["frac"], rf["avg_launch_ms"], rf["path_3x3"]["achieved"], 100 * rf["path_3x3"]["frac"], rf["all_kernels_tflops"],
           (b.get("cpu_baseline") or {}).get("value"), (b.get("cpu_baseline") or {}).get("cores"), tag, nfwd))
block = "<!-- BEGIN %s TABLE -->\n%s%s\n<!-- END %s TABLE -->" % (tag, head, "\n".join(rows), tag)
readme = os.path.join(ROOT, "profiles", "README.md")
txt = open(readme).read()
pat = re.compile(r"<!-- BEGIN %s TABLE -->.*?<!-- END %s TABLE -->" % (tag, tag), re.S)
txt = pat.sub(lambda m: block, txt) if pat.search(txt) else txt + "\n" + block + "\n"
open(readme, "w").write(txt)
print(block)
