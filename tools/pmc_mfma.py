"""Summarise the matrix-pipe / clock PMC passes into profiles/<tag>_pmc_mfma[_<dtype>].{csv,json}.

north_star asks for "MFMA utilisation against chip peak" from rocprof; DESIGN.md's claim that the fp32 kernels sit at the
CLOCK-LIMITED ceiling (0.79 of the nominal 2.4 GHz peak = 0.9x of what the sustained clock allows) must be recomputable from
tracked files.  Inputs: separate rocprofv3 counter passes of `scratch/run_fwd.py 32 4 <dtype>` (B = 32, four forwards; the last two
are used), counters only with --kernel-trace, the program directly after `--` (tools/r3_pmc_pass.sh):
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE ... -d gpurun_out/<tag>_pmc_mfma_a[_<dtype>] -- python3 scratch/run_fwd.py 32 4 <dtype>
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES ... -d gpurun_out/<tag>_pmc_mfma_b[_<dtype>] -- ...
Any subset of these counters may be present (a name rocprofv3 -L does not list on the box is simply skipped by the pass script).

Per kernel (summed over the launches of the used forwards):
  duration            dispatch End - Start of the SAME pass (profiled dispatches are serialised, so no overlap)
  clock_ghz           GRBM_GUI_ACTIVE / 8 / duration — rocprofv3 reports the SUM over the 8 XCDs (MI355X_MICROARCH.md, "DVFS
                      give-back"); reads high on dispatches shorter than ~0.3 ms (ramp-in idle is not GUI-active)
  mfma_busy           SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): share of the GPU-active cycles in which a
                      SIMD's matrix pipe is busy = utilisation against what the clock the chip actually held allows
  mfma_busy_nominal   SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs): the same against the NOMINAL clock —
                      the quantity a "fraction of the 157.3 TFLOP/s peak" measures (before the K / N padding of a layer)
so  frac_of_nominal_peak  ~=  mfma_busy x clock_ghz / 2.4 x (algorithmic / issued FLOP).

Usage: python tools/pmc_mfma.py <tag> [dtype]"""
import glob
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blindshadowremoval_amd.build import source_sha16      # noqa: E402

N_SIMD = 256 * 4            # MI355X: 256 CUs x 4 SIMDs
N_XCD = 8
NOMINAL_GHZ = 2.4

tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
sfx = "" if dtype == "f32" else "_" + dtype


def short(name: str) -> str:
    import re
    for a in ("void bsr::", "bsr::", "(bsr::ConvArgs)", "(ConvArgs)", "(bsr::ConvN16Args)", "(ConvN16Args)", "(bsr::StemArgs)", "(StemArgs)"):
        name = name.replace(a, "")
    return re.sub(r"\((ConvArgs|ConvN16Args|StemArgs|float|bsr::|unsigned|int|void).*", "", name)


def load_pass(d):
    """-> DataFrame [kernel, dispatch, dur_ns, <counter columns>] of the last half of the forwards of one pass."""
    files = glob.glob(os.path.join(ROOT, "gpurun_out", d, "*", "*counter_collection.csv"))
    if not files:
        return None
    c = pd.read_csv(max(files, key=os.path.getmtime))
    c = c[c["Kernel_Name"].str.contains("bsr::")]
    ids = sorted(c["Dispatch_Id"].unique())
    first_of_fwd = [i for i in ids if "stem7_kernel" in c[c["Dispatch_Id"] == i]["Kernel_Name"].iloc[0]]
    assert len(first_of_fwd) >= 2, "expected at least two forwards"
    start = first_of_fwd[len(first_of_fwd) // 2]                      # first dispatch of the second half of the forwards
    c = c[c["Dispatch_Id"] >= start].copy()
    c["kernel"] = c["Kernel_Name"].map(short)
    c["dur_ns"] = c["End_Timestamp"] - c["Start_Timestamp"]
    wide = c.pivot_table(index=["Dispatch_Id", "kernel", "dur_ns"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
    wide.attrs["forwards"] = len(first_of_fwd) - len(first_of_fwd) // 2
    return wide


passes = {}
for p in ("a", "b", "c"):
    w = load_pass("%s_pmc_mfma_%s%s" % (tag, p, sfx))
    if w is not None:
        passes[p] = w
if "a" not in passes:
    raise SystemExit("no gpurun_out/%s_pmc_mfma_a%s pass found" % (tag, sfx))

rows = {}
for p, w in passes.items():
    nf = w.attrs["forwards"]
    counters = [c for c in w.columns if c not in ("Dispatch_Id", "kernel", "dur_ns")]
    g = w.groupby("kernel")
    agg = g[counters + ["dur_ns"]].sum()
    agg["launches"] = g.size()
    for k, r in agg.iterrows():
        d = rows.setdefault(k, {})
        d.setdefault("launches_per_forward", int(round(r["launches"] / nf)))
        for cname in counters:
            d[cname] = float(r[cname]) / nf                           # per forward
        d["dur_us_pass_" + p] = float(r["dur_ns"]) / nf / 1e3
        if "GRBM_GUI_ACTIVE" in counters:
            d["dur_us"] = float(r["dur_ns"]) / nf / 1e3               # the duration the clock is derived against (same pass as GRBM_GUI_ACTIVE)

out = []
for k, d in rows.items():
    rec = {"kernel": k, "launches_per_forward": d["launches_per_forward"]}
    dur_us = d.get("dur_us", d.get("dur_us_pass_a"))
    rec["us_per_forward"] = round(dur_us, 1)
    gui = d.get("GRBM_GUI_ACTIVE")
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if gui:
        rec["clock_ghz"] = round(gui / N_XCD / (dur_us * 1e3), 3)
    if busy is not None:
        # SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE may come from different passes: scale by that pass's own duration
        dur_busy = next((d["dur_us_pass_" + p] for p, w in passes.items() if "SQ_VALU_MFMA_BUSY_CYCLES" in w.columns), dur_us)
        rec["mfma_busy_nominal"] = round(busy / (dur_busy * 1e3 * NOMINAL_GHZ * N_SIMD), 4)
        if gui:
            rec["mfma_busy"] = round(busy / (gui / N_XCD * N_SIMD) * (dur_us / dur_busy), 4)
    for cname in ("SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_VALU_MFMA_MOPS_BF16",
                  "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_F32", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY",
                  "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
        if cname in d:
            rec[cname] = round(d[cname], 1)
    out.append(rec)
out.sort(key=lambda r: -r["us_per_forward"])
df = pd.DataFrame(out)
df.to_csv(os.path.join(ROOT, "profiles", "%s_pmc_mfma%s.csv" % (tag, sfx)), index=False)
per_kernel = {r["kernel"]: {k: v for k, v in r.items() if k != "kernel"} for r in out}
tot_us = sum(r["us_per_forward"] for r in out)
summary = {
    "batch": 32, "dtype": dtype, "kernel_src_sha16": source_sha16(), "forwards_used": passes["a"].attrs["forwards"],
    "counters": sorted({c for w in passes.values() for c in w.columns if c not in ("Dispatch_Id", "kernel", "dur_ns")}),
    "n_simd": N_SIMD, "nominal_ghz": NOMINAL_GHZ, "per_kernel": per_kernel,
    "forward": {
        "us": round(tot_us, 1),
        "clock_ghz_time_weighted": round(sum(r.get("clock_ghz", 0) * r["us_per_forward"] for r in out) / tot_us, 3) if all("clock_ghz" in r for r in out) else None,
        "mfma_busy_time_weighted": round(sum(r.get("mfma_busy", 0) * r["us_per_forward"] for r in out) / tot_us, 4) if all("mfma_busy" in r for r in out) else None,
    },
    "note": "counters per dispatch, serialised profiled dispatches, last half of the forwards of a B=32 run; clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; "
            "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); mfma_busy_nominal = the same against duration x 2.4 GHz. "
            "Profiled passes clock 2-5 % lower than un-profiled runs (MI355X_MICROARCH.md, DVFS give-back (2)): compare ratios, not absolute times.",
}
with open(os.path.join(ROOT, "profiles", "%s_pmc_mfma%s.json" % (tag, sfx)), "w") as fjson:
    json.dump(summary, fjson, indent=1)
print(df.to_string())
print(json.dumps(summary["forward"]))
