"""ONE generator for a round's section of profiles/README.md (rounds 3-5 each carried their own copy of this file: VERDICT round 5, code health).

    python tools/round_readme.py r6

(re)writes, between <!-- BEGIN r6 NOTES --> / <!-- END r6 NOTES -->, what can be COMPUTED from the tracked profiles/r6_* files — the summary
table of the three dtypes (value, sustained value and clock, dominant kernel, whole-forward fractions, parity), the kernel-group tables with
their matrix-roof / HBM fractions (algorithmic and, where the counter passes match the kernel sources, from the counters), the side lines
(TSM 512x512, B = 16, one-rank RCCL, N = 2 on one GPU, loops) — and puts the round's header in front of the per-kernel tables
tools/update_profiles.py wrote (<!-- BEGIN r6 TABLE --> ...).  Prose (what was tried, why) lives in profiles/HISTORY.md and DESIGN.md section 7
and is written by hand: nothing here invents a sentence a number does not support.  Missing files are skipped, not guessed."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
prev = "r%d" % (int(tag[1:]) - 1)


def J(name):
    p = os.path.join(ROOT, "profiles", name)
    if not os.path.isfile(p):
        return None
    with open(p) as f:
        return json.load(f)


def g(d, *keys, default=None):
    for k in keys:
        if not isinstance(d, dict) or k not in d or d[k] is None:
            return default
        d = d[k]
    return d


def fmt(v, f="%.0f"):
    return "-" if v is None else f % v


rows, groups = [], []
for dtype, sfx in (("f32", ""), ("f32x3", "_f32x3"), ("f16", "_f16")):
    b, p = J("%s_bench_n1%s.json" % (tag, sfx)), J("%s_bench_n1%s.json" % (prev, sfx))
    if b is None:
        continue
    rf, su = b["roofline"], b.get("sustained") or {}
    par = g(b, "parity") or g(b, "f32x3", "parity") if dtype != "f32" else g(b, "cpu_baseline", "parity")
    rows.append("| %s | **%s** (%s) | %s | %s at %s GHz (x%s) | %s | `%s`: %s of its %s roof | %s / %s | %s |" % (
        dtype, fmt(b["value"]), fmt(g(p, "value")), fmt(b["ms_per_step"], "%.3f"), fmt(su.get("value")), fmt(su.get("clock_ghz"), "%.2f"), fmt(su.get("vs_value"), "%.3f"),
        fmt(g(b, "two_in_flight_value")), rf["kernel"].split(" (")[0][:60], fmt(rf["frac"], "%.3f"), rf["bound"],
        fmt(rf.get("all_kernels_tflops"), "%.1f"), fmt(rf.get("all_kernels_tflops_executed"), "%.1f"), fmt(g(par, "max_abs_err"), "%.1e") if isinstance(par, dict) else "-"))
    kg = ["| `%s` | %.3f | %d | %.3f | %.3f | %s | %s |" % (k[:100], v["ms"], v["launches"], v["frac"], v["hbm_frac"], v.get("hbm_frac_counters", "-"), v.get("traffic_ratio", "-"))
          for k, v in rf["kernel_groups"].items()]
    groups.append("Kernel groups, %s (`%s_bench_n1%s.json`; ms per forward, launches, fraction of the group's matrix roof, HBM fraction by algorithmic bytes, by the counters, "
                  "counter / algorithmic bytes):\n\n| group | ms | launches | matrix frac | hbm frac | hbm frac (counters) | traffic ratio |\n|---|---|---|---|---|---|---|\n%s\n" % (dtype, tag, sfx, "\n".join(kg)))

side = []
t5 = J("%s_bench_tsm512.json" % tag)
if t5: side.append("* `%s_bench_tsm512.json` — BASELINE configs[4]'s per-rank shape (TSM generator, 512x512 frames): **%.0f frames/s** at f32, dominant kernel %.3f of its roof." % (tag, t5["value"], t5["roofline"]["frac"]))
t5x = J("%s_bench_tsm512_f32x3.json" % tag)
if t5x: side.append("* `%s_bench_tsm512_f32x3.json` — the same at f32x3: %.0f frames/s." % (tag, t5x["value"]))
b16 = J("%s_bench_b16.json" % tag)
if b16: side.append("* `%s_bench_b16.json` — %s: **%.0f images/s**." % (tag, b16["config"]["workload"].split(" synthetic")[0], b16["value"]))
d1 = J("%s_bench_dist1.json" % tag)
if d1: side.append("* `%s_bench_dist1.json` — the RCCL path on one rank: %.0f images/s, gather %.3f ms exposed per step, verified: %s." % (
    tag, d1["value"], g(d1, "config", "allgather", "ms_exposed_per_step", default=0.0), g(d1, "config", "allgather", "verified")))
d2 = J("%s_bench_n2_one_gpu.json" % tag)
if d2: side.append("* `%s_bench_n2_one_gpu.json` — `bench.py --gpus 2 --device 0`: the N = 2 step (packed forward, double-buffered peer-copy gather, verification) with the REAL "
                   "generator, both ranks on one GPU: %.0f images/s for the pair (two forwards sharing a chip — not a scaling figure), verified: %s." % (
                       tag, d2["value"], g(d2, "config", "allgather", "verified")))
for kind in ("ffhq", "ucb"):
    ls = [J("%s_loop_%s_%d.json" % (tag, kind, i)) for i in (1, 2)]
    ls = [x["loop"] for x in ls if x and "loop" in x]
    if ls:
        keys = [k for k in ls[0] if isinstance(ls[0][k], dict) and "images_per_sec" in ls[0][k]]
        side.append("* `%s_loop_%s_{1,2}.json` — `FSRNet.%s` end to end: %s." % (tag, kind, "testFFHQ" if kind == "ffhq" else "test", "; ".join(
            "%s %s images/s" % (k, " / ".join("%.0f" % x[k]["images_per_sec"] for x in ls if k in x)) for k in keys)))
stg = J("%s_loop_stage_table.json" % tag)
if stg and "stages" in stg:
    st = stg["stages"]
    side.append("* `%s_loop_stage_table.json` — the loops' host stages, each ALONE through %s worker processes (`tools/loop_stage_table.py`): %s." % (
        tag, stg.get("worker_processes", "?"), "; ".join("%s %.0f items/s (%.2f ms of CPU per item alone)" % (k, v["items_per_sec"], v["job_cpu_ms_alone"])
                                                       for k, v in st.items() if k.startswith("loader_host_half"))))
for extra, what in (("%s_nsplit_f32.txt" % tag, "the fp32 c3q GEMM under finer N splits (`-DBSR_NL_NSPLIT`; HISTORY: bounded attempt 2)"),
                    ("%s_batch_sweep.json" % tag, "`tools/batch_sweep.py`: images/s against the batch per forward"),
                    ("%s_lane_overlap.txt" % tag, "`tools/lane_overlap.py`: the two-forwards-in-flight mode in the profiler's view"),
                    ("%s_pmc_mfma.json" % tag, "(+ `_f32x3`, `_f16`) matrix-pipe utilisation and clock per kernel from the counter passes (`tools/pmc_mfma.py`)"),
                    ("%s_pmc_traffic.json" % tag, "(+ `_f32x3`, `_f16`) HBM bytes per kernel from the counter passes (`tools/pmc_traffic.py`), the source of `traffic` and `hbm frac (counters)`"),
                    ("%s_kernel_stats_tsm512.csv" % tag, "(+ `_b16`) rocprofv3 kernel-trace summaries of configs[4]'s per-rank shape and of B = 16"),
                    ("%s_kernel_stats_ucb_post.csv" % tag, "(+ `_png`, `_unfilter`) rocprofv3 kernel-trace summaries of the loops' device stages alone: `bsr_ucb_post` on 16 UCB items "
                     "(`scratch/ucb_time.py`: the stage chain), `bsr_png_encode` / `_figs` on 16 strips (`scratch/png_time.py`), `bsr_png_unfilter` on 16 / 32 photographs (`scratch/unf_time.py`)")):
    if os.path.isfile(os.path.join(ROOT, "profiles", extra)):
        side.append("* `%s` — %s." % (extra, what))
mg = os.path.join(ROOT, "profiles", "%s_f16_margins.txt" % tag)
if os.path.isfile(mg):
    side.append("* `%s_f16_margins.txt` — what the f16-mode tests measured against F16_TOL = 2e-3: %s." % (
        tag, "; ".join("%s %s (%s)" % (ln.split()[0].replace("test_", ""), ln.split()[2], ln.split()[-1]) for ln in open(mg).read().strip().splitlines())))

notes = ("<!-- BEGIN %s NOTES -->\n"
         "Computed from the tracked `profiles/%s_*` files by `tools/round_readme.py %s` (prose: `profiles/HISTORY.md`, `DESIGN.md` section 7).  `value` = one forward at a time, "
         "%s in brackets; `sustained` = the >= 3-s region after >= 2 s of warm-up with the in-kernel clock; whole-forward TFLOP/s by the reference's op count / by the work executed.\n\n"
         "| dtype | images/s (%s) | ms / step | sustained | two in flight | dominant kernel | all kernels TFLOP/s | max abs err |\n|---|---|---|---|---|---|---|---|\n%s\n\n%s\n%s\n"
         "<!-- END %s NOTES -->" % (tag, tag, tag, prev, prev, "\n".join(rows), "\n".join(groups), "\n".join(side), tag))

readme = os.path.join(ROOT, "profiles", "README.md")
s = open(readme).read()
pat = re.compile(r"<!-- BEGIN %s NOTES -->.*?<!-- END %s NOTES -->" % (tag, tag), re.S)
if pat.search(s):
    s = pat.sub(lambda m: notes, s)
else:
    # a new round: its header, the per-kernel tables update_profiles.py appended at the end of the file, and the notes, in front of the previous round
    tables = []
    for sfx in ("", "_f32x3", "_f16"):
        m = re.search(r"\n?<!-- BEGIN %s%s TABLE -->.*?<!-- END %s%s TABLE -->\n?" % (tag, sfx, tag, sfx), s, re.S)
        if m:
            tables.append(m.group(0).strip("\n"))
            s = s[:m.start()] + "\n" + s[m.end():]
    head = "## Round %s (MI355X, ROCm 7.2, B = 32 per forward)\n\n" % tag[1:]
    block = head + "\n\n".join(tables + [notes]) + "\n\n"
    m = re.search(r"^## Round %s " % prev[1:], s, re.M)
    s = s[:m.start()] + block + s[m.start():] if m else s.rstrip("\n") + "\n\n" + block
open(readme, "w").write(s)
print(notes[:1500])
