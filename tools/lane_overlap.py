"""How much of a `rocprofv3 --kernel-trace` run had TWO of the library's kernels on the chip at once — the profiler's view of
`bench.py --streams 2` (two forwards in flight on two HIP streams).  Usage:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3_prof_lanes -- python3 bench.py --no-cpu-baseline --no-secondary --repeats 1
    python tools/lane_overlap.py gpurun_out/r3_prof_lanes > profiles/r3_lane_overlap.txt"""
import glob
import os
import sys

import pandas as pd

d = sys.argv[1]
trace = max(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)
t = pd.read_csv(trace)
t = t[t["Kernel_Name"].str.contains("bsr::")].copy()
print("trace: %s — %d dispatches of the library's kernels on hardware queues %s (streams %s)" % (
    os.path.basename(trace), len(t), sorted(t["Queue_Id"].unique().tolist()), sorted(t["Stream_Id"].unique().tolist())))
# sweep line over start / end events
ev = sorted([(s, 1) for s in t["Start_Timestamp"]] + [(e, -1) for e in t["End_Timestamp"]])
active, last, busy = 0, ev[0][0], {}
for ts, dlt in ev:
    busy[active] = busy.get(active, 0) + (ts - last)
    active += dlt
    last = ts
span = ev[-1][0] - ev[0][0]
ge1 = sum(v for k, v in busy.items() if k >= 1)
ge2 = sum(v for k, v in busy.items() if k >= 2)
print("span %.1f ms: >= 1 kernel resident %.1f %% of it, >= 2 kernels resident %.1f %% (%.1f %% of the busy time)" % (
    span / 1e6, 100.0 * ge1 / span, 100.0 * ge2 / span, 100.0 * ge2 / max(ge1, 1)))
# the same per phase of bench.py: the two-lane regions come first (warm-up + timed region), then the single-stream region and the event-timed forwards
per_q = t.groupby("Queue_Id").agg(n=("Kernel_Name", "size"), first=("Start_Timestamp", "min"), last=("End_Timestamp", "max"))
for q, r in per_q.iterrows():
    print("queue %s: %d dispatches over %.1f ms" % (q, r["n"], (r["last"] - r["first"]) / 1e6))
if len(per_q) > 1:                                   # inside the window in which the second queue is in use (the warm-up + timed region of --streams 2)
    q2 = per_q.sort_values("n").index[0]
    lo, hi = per_q.loc[q2, "first"], per_q.loc[q2, "last"]
    # the second queue is used again late in the run (round 4: the event-timed two-lane forwards of `roofline_in_flight`): the window is
    # its FIRST busy period — warm-up + timed region — which ends where the queue then stays idle for more than 5 ms (the single-stream region)
    tq = t[t["Queue_Id"] == q2].sort_values("Start_Timestamp")
    starts, ends = tq["Start_Timestamp"].tolist(), tq["End_Timestamp"].tolist()
    for i in range(1, len(starts)):
        if starts[i] - ends[i - 1] > 5e6:
            hi = ends[i - 1]
            break
    active, last, b2 = 0, ev[0][0], {}
    for ts, dlt in ev:
        a, c = max(last, lo), min(ts, hi)
        if c > a:
            b2[active] = b2.get(active, 0) + (c - a)
        active += dlt
        last = ts
    w = hi - lo
    print("while queue %s is in use (%.1f ms): >= 2 kernels resident %.1f %% of the time, exactly 1 %.1f %%, none %.1f %%" % (
        q2, w / 1e6, 100.0 * sum(v for k, v in b2.items() if k >= 2) / w, 100.0 * b2.get(1, 0) / w, 100.0 * b2.get(0, 0) / w))
dom = t[t["Kernel_Name"].str.contains("igemm_conv_kernel<3, 3, 1, true, 4, 32, 4, 1, 1, 2, 32, 1")]
if len(per_q) > 1:
    dom = dom[(dom["Start_Timestamp"] >= lo) & (dom["End_Timestamp"] <= hi)]          # the launches of the two-lane window
dur = (dom["End_Timestamp"] - dom["Start_Timestamp"]) / 1e3
print("dominant kernel: %d launches, duration min %.0f / median %.0f / max %.0f us (alone: ~410 us; longer = it shared the chip with the other forward's kernels)" % (
    len(dom), dur.min(), dur.median(), dur.max()))

# per-kernel mean duration inside the two-lane window against the whole run's single-lane launches (outside the window): the profiler's
# cross-check of bench.py's `roofline_in_flight.dominant_kernel.avg_launch_ms` (event-bracketed) — a kernel that shares the chip lasts longer
if len(per_q) > 1:
    import re
    t["short"] = t["Kernel_Name"].map(lambda n: re.sub(r"\((bsr::|float|unsigned|int|void).*", "", n.replace("void bsr::", "").replace("bsr::", "")))
    t["dur_us"] = (t["End_Timestamp"] - t["Start_Timestamp"]) / 1e3
    inside = t[(t["Start_Timestamp"] >= lo) & (t["End_Timestamp"] <= hi)]
    outside = t[t["Start_Timestamp"] > hi]
    a = inside.groupby("short")["dur_us"].agg(["count", "mean"])
    b = outside.groupby("short")["dur_us"].agg(["count", "mean"])
    j = a.join(b, lsuffix="_two_lanes", rsuffix="_alone", how="inner").sort_values("mean_two_lanes", ascending=False)
    print("per kernel, mean launch duration in us — two forwards in flight | one at a time (later phases of the same run):")
    for k, r in j.iterrows():
        if r["mean_two_lanes"] >= 20:
            print("  %-75s %7.1f (%4d launches) | %7.1f (%4d)  x%.2f" % (k[:75], r["mean_two_lanes"], r["count_two_lanes"], r["mean_alone"], r["count_alone"], r["mean_two_lanes"] / r["mean_alone"]))
