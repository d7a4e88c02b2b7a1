"""Summarise the HBM-traffic PMC passes into profiles/<round>_pmc_traffic.{csv,json}.

Inputs: two rocprofv3 runs of `scratch/run_fwd.py 32 2` (B = 32, two forwards), one per counter because FETCH_SIZE
(3 TCC slots) and WRITE_SIZE (2) do not fit one pass (MI355X_MICROARCH.md, rocprofv3 PMC slots):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 scratch/run_fwd.py 32 2
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 scratch/run_fwd.py 32 2
Counter values are KB per dispatch.  gfx950 correction (same guide, HBM section): FETCH_SIZE counts 128-B read requests
as 64 B, so wide coalesced reads report exactly half their bytes -> read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact.
Only the second forward of each run is used (first one carries one-time allocation / clears)."""
import glob
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"


def load(d, name):
    c = pd.read_csv(max(glob.glob(os.path.join(ROOT, "gpurun_out", d, "*", "*counter_collection.csv")), key=os.path.getmtime))
    c = c[(c["Counter_Name"] == name) & ~c["Kernel_Name"].str.contains("fillBuffer")]
    ids = sorted(c["Dispatch_Id"].unique())
    assert len(ids) % 2 == 0, "expected exactly two forwards"
    c = c[c["Dispatch_Id"].isin(ids[len(ids) // 2:])]            # the second forward
    c["kernel"] = (c["Kernel_Name"].str.replace("void bsr::", "").str.replace("bsr::", "").str.replace("(bsr::ConvArgs)", "")
                   .str.replace("(bsr::ConvN16Args)", "").str.replace(r"\(float const\*.*", "", regex=True))
    return c.groupby("kernel")["Counter_Value"].agg(["sum", "count"])


f, w = load("pmc_fetch", "FETCH_SIZE"), load("pmc_write", "WRITE_SIZE")
m = f.join(w, lsuffix="_fetch", rsuffix="_write")
m["launches_per_forward"] = m["count_fetch"]
m["read_MB_per_forward"] = 2 * m["sum_fetch"] * 1024 / 1e6          # gfx950: x2
m["write_MB_per_forward"] = m["sum_write"] * 1024 / 1e6
m["hbm_MB_per_forward"] = m["read_MB_per_forward"] + m["write_MB_per_forward"]
out = m[["launches_per_forward", "read_MB_per_forward", "write_MB_per_forward", "hbm_MB_per_forward"]].round(1)
out.to_csv(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.csv"))
is33 = out.index.str.contains("<3, 3")          # the 3x3 / stride-2 3x3 / transposed 3x3 launches (roofline kernel class)
dom = out[out.index.str.contains("<3, 3, 1, true, 4, 32, 4, 1, 1, 2, 32")]
summary = {
    "batch": 32,
    "dominant_kernel_hbm_bytes_per_launch": float(dom["hbm_MB_per_forward"].sum() * 1e6 / max(dom["launches_per_forward"].sum(), 1)),
    "path_3x3_hbm_bytes_per_forward": float(out.loc[is33, "hbm_MB_per_forward"].sum() * 1e6),
    "path_3x3_launches": int(out.loc[is33, "launches_per_forward"].sum()),
    "all_kernels_hbm_bytes_per_forward": float(out["hbm_MB_per_forward"].sum() * 1e6),
    "note": "read = 2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B), write = WRITE_SIZE; KB per dispatch, second forward of a B=32 run",
}
with open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json"), "w") as fjson:
    json.dump(summary, fjson, indent=1)
print(out.to_string())
print(summary)
