"""Summarise the HBM-traffic PMC passes into profiles/<tag>_pmc_traffic[_<dtype>].{csv,json}.

Inputs: separate rocprofv3 runs of `scratch/run_fwd.py 32 2 <dtype>` (B = 32, two forwards), one per counter set because FETCH_SIZE
(3 TCC slots) and WRITE_SIZE (2) do not fit one pass (MI355X_MICROARCH.md, rocprofv3 PMC slots), plus an optional L2 hit/miss pass:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/<tag>_pmc_fetch[_<dtype>] -- python3 scratch/run_fwd.py 32 2 <dtype>
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/<tag>_pmc_write[_<dtype>] -- python3 scratch/run_fwd.py 32 2 <dtype>
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/<tag>_pmc_l2[_<dtype>] -- python3 scratch/run_fwd.py 32 2 <dtype>
Counter values are KB per dispatch.  gfx950 correction (same guide, HBM section): FETCH_SIZE counts 128-B read requests as 64 B, so
wide coalesced reads report exactly half their bytes -> read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact.  Only the second forward
of each run is used (the first carries one-time allocation / clears).  The JSON records the hash of the kernel sources the passes
ran on (the snapshot gpurun shipped = this work tree), so bench.py only quotes the figure for the same kernels.

Usage: python tools/pmc_traffic.py <tag> [dtype]"""
import glob
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blindshadowremoval_amd.build import source_sha16      # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
sfx = "" if dtype == "f32" else "_" + dtype

# algorithmic activation bytes per image of the layers whose launches are summarised below: input read once + output written once (fp32)
LAYER_IO_MB = {"up2": (64 * 64 * 160 + 128 * 128 * 64) * 4e-6, "up3": (128 * 128 * 128 + 256 * 256 * 64) * 4e-6,
               "clr_up3": (128 * 128 * 96 + 256 * 256 * 64) * 4e-6}
if dtype != "f32":          # 32-channel K chunks everywhere: clr_up1 runs on the same instantiation (bench.py's 16-bit kernel groups)
    LAYER_IO_MB["clr_up1"] = (32 * 32 * 261 + 64 * 64 * 128) * 4e-6


def load(d, name):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", d, "*", "*counter_collection.csv"))
    if not files:
        return None
    c = pd.read_csv(max(files, key=os.path.getmtime))
    c = c[(c["Counter_Name"] == name) & ~c["Kernel_Name"].str.contains("fillBuffer")]
    ids = sorted(c["Dispatch_Id"].unique())
    assert len(ids) % 2 == 0, "expected exactly two forwards"
    c = c[c["Dispatch_Id"].isin(ids[len(ids) // 2:])].copy()            # the second forward
    c["kernel"] = (c["Kernel_Name"].str.replace("void bsr::", "").str.replace("bsr::", "").str.replace(r"\((bsr::)?(Conv|ConvN16|Stem)Args.*\)$", "", regex=True)
                   .str.replace(r"\(float const\*.*", "", regex=True))
    return c.groupby("kernel")["Counter_Value"].agg(["sum", "count"])


f, w = load("%s_pmc_fetch%s" % (tag, sfx), "FETCH_SIZE"), load("%s_pmc_write%s" % (tag, sfx), "WRITE_SIZE")
m = f.join(w, lsuffix="_fetch", rsuffix="_write")
m["launches_per_forward"] = m["count_fetch"]
m["read_MB_per_forward"] = 2 * m["sum_fetch"] * 1024 / 1e6          # gfx950: x2
m["write_MB_per_forward"] = m["sum_write"] * 1024 / 1e6
m["hbm_MB_per_forward"] = m["read_MB_per_forward"] + m["write_MB_per_forward"]
cols = ["launches_per_forward", "read_MB_per_forward", "write_MB_per_forward", "hbm_MB_per_forward"]
hit, miss = load("%s_pmc_l2%s" % (tag, sfx), "TCC_HIT_sum"), load("%s_pmc_l2%s" % (tag, sfx), "TCC_MISS_sum")
if hit is not None and miss is not None:
    m["l2_hit_rate"] = hit["sum"] / (hit["sum"] + miss["sum"])
    cols.append("l2_hit_rate")
out = m[cols].round(3)
out.to_csv(os.path.join(ROOT, "profiles", "%s_pmc_traffic%s.csv" % (tag, sfx)))
is33 = out.index.str.contains("<3, 3") | out.index.str.contains("conv3_f16_kernel")          # the 3x3 / stride-2 3x3 / transposed 3x3 launches (roofline kernel class)
dom = out[out.index.str.contains("<3, 3, 1, true, 4, 32, 4, 1, 1, 2, 32") | out.index.str.contains("conv3_f16_kernel<true")]      # igemm_{conv,h16}_kernel<3,3,1,TR,...,NI=2,CC=32>: up2, up3, clr_up3; f16 (round 6): conv3_f16_kernel<TR>
alg = sum(LAYER_IO_MB.values()) * 32 * 1e6 / len(LAYER_IO_MB)
per_kernel = {k: {"launches_per_forward": int(r["launches_per_forward"]), "hbm_bytes_per_forward": float(r["hbm_MB_per_forward"] * 1e6)}
              for k, r in out.iterrows()}
summary = {
    "batch": 32, "dtype": dtype, "kernel_src_sha16": source_sha16(), "per_kernel": per_kernel,
    "dominant_kernel_rows": list(dom.index),
    "dominant_kernel_hbm_bytes_per_launch": float(dom["hbm_MB_per_forward"].sum() * 1e6 / max(dom["launches_per_forward"].sum(), 1)),
    "dominant_kernel_algorithmic_bytes_per_launch": alg,
    "path_3x3_hbm_bytes_per_forward": float(out.loc[is33, "hbm_MB_per_forward"].sum() * 1e6),
    "path_3x3_launches": int(out.loc[is33, "launches_per_forward"].sum()),
    "all_kernels_hbm_bytes_per_forward": float(out["hbm_MB_per_forward"].sum() * 1e6),
    "note": "read = 2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B), write = WRITE_SIZE; KB per dispatch, second forward of a B=32 run; "
            "the memory-side counters include Infinity-Cache hits (MI355X_MICROARCH.md, HBM)",
}
with open(os.path.join(ROOT, "profiles", "%s_pmc_traffic%s.json" % (tag, sfx)), "w") as fjson:
    json.dump(summary, fjson, indent=1)
print(out.to_string())
print(summary)
