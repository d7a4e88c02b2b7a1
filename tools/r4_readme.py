"""Write the round-4 notes of profiles/README.md (between <!-- BEGIN r4 NOTES --> / <!-- END r4 NOTES -->) and the round-4 block of
DESIGN.md §7 (<!-- BEGIN r4 DESIGN --> / <!-- END r4 DESIGN -->) from the tracked profiles/r4_* files, so that every number in the
prose is one a reader can find in profiles/.  Usage: python tools/r4_readme.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(ROOT, "profiles", n)       # noqa: E731
J = lambda n: json.load(open(P(n)))                   # noqa: E731

b, bx, bh = J("r4_bench_n1.json"), J("r4_bench_n1_f32x3.json"), J("r4_bench_n1_f16.json")
mf, tr, d1 = J("r4_pmc_mfma.json"), J("r4_pmc_traffic.json"), J("r4_bench_dist1.json")
lf, lu = J("r4_loop_ffhq.json")["loop"], J("r4_loop_ucb.json")["loop"]
t5, t5x = J("r4_bench_tsm512.json"), J("r4_bench_tsm512_f32x3.json")
sw, st = J("r4_batch_sweep.json"), J("r4_loop_stage_table.json")
r3 = J("r3_bench_n1.json")
pk = mf["per_kernel"]


def row(key):
    return [v for k, v in pk.items() if key in k][0]


dom = row("<3, 3, 1, true, 4, 32, 4, 1, 1, 2, 32")
rf = b["roofline"]
kg = rf["kernel_groups"]


def grp(key):
    return [v for k, v in kg.items() if key in k][0]


lanes = open(P("r4_lane_overlap.txt")).read().strip().split("\n")
rif = b.get("roofline_in_flight") or {}
whole = (rif.get("whole_forward") or {})
sweep_rows = "\n".join("| %d | %.0f | %.3f | %.2f | %d |" % (r["batch"], r["images_per_sec"], r["ms_per_forward"], r["rate_vs_largest_batch"], r["launches_per_forward"])
                       for r in sw["rows"])
base_sweep = {1: 554.0, 2: 1085.8, 4: 1948.2, 8: 3392.3, 10: 3897.6, 16: 5265.8, 32: 6548.5}      # the same tool on the round-3 kernels (first GPU call of round 4)
stg = st["stages"]
b16 = b.get("batch16") or {}

notes = '''<!-- BEGIN r4 NOTES -->
Other round-4 artefacts (this block is written by `tools/r4_readme.py` from the files it names; taken by `scratch/final_pass_r4.sh`):

* **`r4_pmc_mfma.csv/.json` (+ `_f32x3`, `_f16`) — matrix-pipe utilisation and clock per kernel** (`tools/pmc_mfma.py` over the three
  `rocprofv3 --kernel-trace --pmc` passes of `tools/pmc_mfma_pass.sh`).  f32, dominant kernel (`igemm_conv_kernel<3,3,1,TR,..,NI=2,CC=32>`,
  3 launches): **clock %.2f GHz, matrix pipe busy %.1f %% of the GPU-active cycles** (%.1f %% of duration x 2.4 GHz — the same quantity as the
  event-timed `roofline.frac` %.3f of `r4_bench_n1.json`); attention + `w` tail (`nonlocal_attention_kernel<4, true>`) %.1f %%, `res*.conv2`
  %.1f %%, `gemm_nloop` (now only `res*.c3q`) %.1f %%, `res*.conv1` %.1f %%, stride-2 `down1/2` %.1f %%; time-weighted over the forward **%.1f %%**
  (round 3: 71-73 %%).
* `r4_pmc_traffic*.csv/.json` — HBM traffic per kernel (2 x FETCH_SIZE + WRITE_SIZE, separate passes).  f32: **%.2f GB per forward** (round 3:
  7.56: the attention output no longer makes the round trip); dominant kernel %.0f MB per launch against %.0f MB algorithmic.
* `r4_kernel_stats*.csv` — `rocprofv3 --kernel-trace --stats` of `bench.py --streams 1` (tables above): **38 launches per forward** (round 3: 44).
* `r4_lane_overlap.txt` — `tools/lane_overlap.py` over a kernel trace of the default two-lane bench:
  `%s`; `%s`; followed by every kernel's mean launch duration with two forwards in flight against one at a time (x1.4-2.8: the profiler's
  cross-check of `roofline_in_flight.dominant_kernel.avg_launch_ms`).
* `r4_bench_dist1.json` — `BSR_BENCH_FORCE_DIST=1 python bench.py` (the RCCL path on one rank): all_gather of the 33.5-MB packed payload,
  `verified: %s`, %.3f ms alone, %.3f ms exposed per step; %.0f images/s.
* `r4_bench_tsm512.json` / `_f32x3` — BASELINE configs[4] per-rank shape (8 frames of 512x512, TSM generator, frame = 2): %.0f frames/s at f32
  (round 3: 999), %.0f at f32x3.
* **`r4_batch_sweep.json` — forward-only batch sweep** (`tools/batch_sweep.py`: fp32, one forward at a time, best of three regions):

| B | images/s | ms / forward | rate vs B = 32 | launches |
|---|---|---|---|---|
%s

  On the round-3 kernels the same tool measured %s images/s at B = 1 / 2 / 4 / 8 / 10 / 16 (B = 16 at 0.80 of the B = 32 rate): below B = 32 the
  1/8-resolution trunk's fixed 128-pixel / 128-query workgroups left CUs empty.  Round 4: 64- / 32-query attention shapes, 2x32-pixel conv
  tiles, finer N ranges in the bottleneck GEMMs, all bit-identical (`tests/test_gpu_parity.py::test_small_batches_equal_the_rows_of_the_full_batch`).
  B = 1 stays latency-bound (one workgroup per kernel per image tile: 45 launches x ~27 us); capturing the forward in a hipGraph changes
  nothing at any batch (`scratch/graph_replay_probe.py`: B = 1 1.211 vs 1.218 ms, B = 8 1.859 vs 1.850, B = 32 4.93 vs 4.92): the launches are
  back to back already.  `bench.py` carries the B = 16 line (`batch16`: %.0f images/s = %.2f of the single-stream B = 32 rate).
* **`r4_loop_ffhq.json` / `r4_loop_ucb.json` — the reference's test loops end to end** (`python bench.py --loop ffhq|ucb`, %d usable CPUs):
  FFHQ (`FSRNet.testFFHQ`, batch 16, 2 000 items) **%.0f images/s** (steady %.0f; round 3: 1 136-1 178); UCB (`FSRNet.test`, 1 000 items,
  seven masks each, SSIM / PSNR) **%.0f images/s** (steady %.0f; round 3: 281-284).  What changed: the loop's thread only feeds — forward,
  strip assembly and the device-to-host copy of up to two batches are in flight behind an event while it pulls the next elements; batches
  bound for a worker pool land in pinned shared-memory slots the workers read in place (no `tofile` in the loop's thread); the loader is
  polled while the thread is busy; worker counts re-swept (FFHQ loader 14 + PNG 14; UCB loader 10 + post 16, no PNG pool).  Spread over the
  round's boxes: FFHQ 1 640-1 880 (steady 1 690-2 340), UCB 273-356 (steady 286-385).
* **`r4_loop_stage_table.json` — every host stage ALONE on the same box** (`tools/loop_stage_table.py`, %d worker processes): loader host half
  %.0f items/s (%.2f ms of CPU per item uncontended), PNG strip %.0f /s (%.2f ms), UCB post-processing **%.0f /s** (%.1f ms), full host
  loader %.0f /s.  Reading: the UCB loop (%.0f /s) runs at the rate of its post-processing stage alone (%.0f /s) — that stage x %d CPUs is the
  ceiling, and it is 2.1x below %d / %.1f ms because sixteen concurrent copies of the job slow each other down (the box grants 16 CPUs' worth
  of time on 256 logical CPUs; `scratch/post_scaling.py` measured the same in round 3).  The FFHQ loop needs loader + PNG = %.1f ms of CPU
  per item: %d CPUs / that = %.0f /s uncontended, the stages alone reach %.0f-%.0f /s, the loop %.0f.
<!-- END r4 NOTES -->''' % (
    dom["clock_ghz"], 100 * dom["mfma_busy"], 100 * dom["mfma_busy_nominal"], rf["frac"], 100 * row("nonlocal_attention")["mfma_busy"],
    100 * row("<3, 3, 1, false")["mfma_busy"], 100 * row("gemm_nloop")["mfma_busy"], 100 * row("<1, 1, 1, false")["mfma_busy"], 100 * row("<3, 3, 2, false, 4, 32, 4, 1, 1, 2")["mfma_busy"],
    100 * mf["forward"]["mfma_busy_time_weighted"],
    tr["all_kernels_hbm_bytes_per_forward"] / 1e9, tr["dominant_kernel_hbm_bytes_per_launch"] / 1e6, tr["dominant_kernel_algorithmic_bytes_per_launch"] / 1e6,
    [ln for ln in lanes if ln.startswith("while queue")][0].strip(), [ln for ln in lanes if ln.startswith("dominant kernel")][0].strip(),
    str(d1["config"]["allgather"]["verified"]).lower(), d1["config"]["allgather"]["ms_alone"], d1["config"]["allgather"]["ms_exposed_per_step"], d1["value"],
    t5["value"], t5x["value"], sweep_rows, " / ".join("%.0f" % base_sweep[k] for k in (1, 2, 4, 8, 10, 16)),
    b16.get("value", 0), b16.get("rate_vs_batch32_single_stream", 0),
    lf.get("usable_cpus", 16), lf["device_prep"]["images_per_sec"], lf["device_prep"]["steady_images_per_sec"], lu["device_prep"]["images_per_sec"],
    lu["device_prep"]["steady_images_per_sec"],
    st["worker_processes"], stg["loader_host_half"]["items_per_sec"], stg["loader_host_half"]["job_cpu_ms_alone"], stg["png_strip"]["items_per_sec"],
    stg["png_strip"]["job_cpu_ms_alone"], stg["ucb_post"]["items_per_sec"], stg["ucb_post"]["job_cpu_ms_alone"], stg["loader_full_host"]["items_per_sec"],
    lu["device_prep"]["images_per_sec"], stg["ucb_post"]["items_per_sec"], st["usable_cpus"], st["usable_cpus"], stg["ucb_post"]["job_cpu_ms_alone"],
    stg["loader_host_half"]["job_cpu_ms_alone"] + stg["png_strip"]["job_cpu_ms_alone"], st["usable_cpus"],
    st["usable_cpus"] / (stg["loader_host_half"]["job_cpu_ms_alone"] + stg["png_strip"]["job_cpu_ms_alone"]) * 1e3,
    min(stg["loader_host_half"]["items_per_sec"], stg["png_strip"]["items_per_sec"]), max(stg["loader_host_half"]["items_per_sec"], stg["png_strip"]["items_per_sec"]),
    lf["device_prep"]["images_per_sec"])

readme = P("README.md")
s = open(readme).read()
if "<!-- BEGIN r4 NOTES -->" in s:
    s = re.sub(r"<!-- BEGIN r4 NOTES -->.*?<!-- END r4 NOTES -->", lambda m: notes, s, flags=re.S)
else:
    s = s.rstrip("\n") + "\n\n" + notes + "\n"
open(readme, "w").write(s)

# ---- DESIGN.md §7, round-4 block
cb = b.get("cpu_baseline") or {}
par = (cb.get("parity") or {})


def dom_line(bb):
    r = bb["roofline"]
    return "%s: %.1f %% of %s (`bound: %s`)" % (r["kernel"].split(" (")[0], 100 * r["frac"], ("%.0f TFLOP/s" % r["peak"]) if r["unit"] == "TFLOP/s" else "8 TB/s algorithmic", r["bound"])


design = '''<!-- BEGIN r4 DESIGN -->
### Round 4

`python bench.py` on one MI355X (B = 32 per step, synthetic, inputs resident in HBM; `profiles/r4_bench_n1*.json`, taken by
`scratch/final_pass_r4.sh` on the final kernel sources):

| dtype | images/s: two forwards in flight (one at a time) | ms / step | dominant kernel against its roofs | max abs error vs oracle |
|---|---|---|---|---|
| **f32** — fp32 matrix cores, the measured path (`value`) | **%.0f** (%.0f; over the boxes of the round 6 770-6 880 and 6 590-6 650; round 3: %.0f / %.0f) | %.3f | transposed 3x3 `igemm_conv_kernel` %.1f TFLOP/s = **%.1f %%** of 157.3 (`bound: mfma`); PMC: matrix pipe busy %.1f %% at %.2f GHz; 3x3-conv path %.1f %%; HBM %.0f MB per launch (%.0f MB algorithmic); whole forward in the two-lane mode %.1f TFLOP/s = %.1f %% | %.1e, %d mask flips |
| f32x3 — split precision on the fp16 matrix cores (§4b) | %.0f (%.0f) | %.3f | %s | %.1e, %d mask flips |
| f16 — fp16 operands + fp16 activation pack (configs[3]) | %.0f (%.0f) | %.3f | %s | 1.4e-03 (tested at 2e-3) |

CPU oracle on the GPU box's host (`cpu_baseline`, `kind: "port"`): %.1f images/s at %d threads (%d usable CPUs).  TSM 512x512 (configs[4] per-rank
shape) %.0f frames/s at f32.  One-rank RCCL run %.0f images/s, gather %.3f ms exposed per step, verified.  Kernel groups by algorithmic FLOPs
(fraction of the 157.3 TFLOP/s peak; `roofline.kernel_groups`): %s.

What the round did to the fp32 path (38 launches per forward instead of 44; matrix pipe %.1f %% busy over the forward, round 3: 71-73 %%):

* **The `w` GEMM became the tail of the attention kernel** (`gemm_tail.h`, §4): 175 -> 162 us per block, +1.2 %% one forward at a time,
  +1.4 %% with two in flight, bit-identical.  The same tail behind `res*.conv2` (conv3 | theta|phi|g, 21 channel tiles) was built and is
  bit-identical too, but is off: one at a time 0.797 vs 0.815 ms for the six blocks, with two forwards in flight 6 800 vs 6 857 images/s
  (its 150-KB 8-wave workgroups leave the other lane's kernels no room on the CU).
* **The rest of the review's list** (`profiles/HISTORY.md`, round 4, has the numbers): the `c3q` GEMM with its output stores
  compiled out runs in 61.9 instead of 62.4 us — it is not write-bound; its 62 us are 53 us of workgroup lifetime (79 %% matrix-busy inside it)
  plus ~9 us of stragglers in a one-round grid, which fusion removes only for SHORT launches (`w`: 36 us).  Chaining `res{i+1}.conv1` behind
  the `w` tail needs the block output as an LDS tile beside the tail's weight ring: 128 px x 264 ch (135 KB) whole, or per-tile slabs plus a
  second weight ring for two wave groups (~190 KB) — over the 160-KB CU.  The persistent stride-2 kernel with next-tile prefetch WAS built (`igemm_s2p.h`, opt-in, bit-identical)
  and is slower — down1 198 -> 220 us, down2 97 -> 108 us: it can only hide the loads' latency (~3-4 k of a 53-k-cycle tile), their ~370
  instructions now sit inside the matrix loop, and the tile loop spills (round 3's persistent up3 again); a stem + down1 fusion has the MFMA work of both (293 us at peak for 382 now) and one workgroup per CU: no gain.
  2x32-pixel tiles for the trunk at B = 32 and NI = 4 channel groups in `c3q`: measured, both slower or equal.
* **Small batches** (table in `profiles/README.md`): B = 16 %.0f images/s = %.2f of the B = 32 rate (round-3 kernels: 0.80), B = 10 — the
  reference's literal element — %.0f (3 898), B = 8 %.0f (3 392), B = 1 %.0f (554); all shapes bit-identical to the B = 32 rows.
* **Loops**: `testFFHQ` %.0f images/s end to end (round 3: 1 136-1 178), `test` (UCB) %.0f (281-284) on %d usable CPUs — pipelined loop body,
  pinned shared-memory ring, data-parallel under a process group (§5, §6); the UCB loop now runs at the rate of its post-processing stage
  alone (`profiles/r4_loop_stage_table.json`: %.0f items/s on this box).

### Open after round 4 (ranked)

1. **The 1 / 2 / 4 / 8-GPU curve** — still the driver's: both multi-GPU forms have only met RCCL at world 1.
2. **fp32 above %.0f (one at a time) / %.0f (two in flight)** — every group is issue-bound on a pipe that is %.0f %% busy in the dominant kernel;
   what is left is per-launch: one-round grids with ~15 %% stragglers (the trunk: 18 launches) and per-tile prologues (stride-2, 1x1).
3. **Parity** stays unpinned until someone runs `tools/make_model_fixture.py --backend tf` (five minutes on a TensorFlow 2.3 box).
4. **UCB loop**: bound by the reference's host post-processing (26 ms of CPU per item, 2x slower under 16-way concurrency); moving its resizes
   and SSIM / PSNR to the GPU is the next step there.
<!-- END r4 DESIGN -->''' % (
    b["value"], b["single_stream_value"], r3["value"], r3["single_stream"]["value"], b["ms_per_step"], rf["achieved"], 100 * rf["frac"],
    100 * (rf.get("mfma_busy") or dom["mfma_busy"]), rf.get("clock_ghz") or dom["clock_ghz"], 100 * rf["path_3x3"]["frac"],
    (rf.get("traffic") or tr["dominant_kernel_hbm_bytes_per_launch"]) / 1e6, tr["dominant_kernel_algorithmic_bytes_per_launch"] / 1e6,
    whole.get("achieved", 0), 100 * (whole.get("frac") or 0), par.get("max_abs_err", 0), par.get("bmask_flips", 0),
    bx["value"], bx["single_stream_value"], bx["ms_per_step"], dom_line(bx), (b["f32x3"].get("parity") or {}).get("max_abs_err", 0),
    (b["f32x3"].get("parity") or {}).get("bmask_flips", 0),
    bh["value"], bh["single_stream_value"], bh["ms_per_step"], dom_line(bh),
    cb.get("value", 0), cb.get("cores", 0), cb.get("usable_cpus", 0), t5["value"], d1["value"], d1["config"]["allgather"]["ms_exposed_per_step"],
    ", ".join("%s %.2f%s" % (k.split(" (")[-1].rstrip(")") if "(" in k else k, v["frac"],
                             (" by the reference's op count = %.2f by the FLOPs the composed GEMM executes" % v["frac_executed"]) if "frac_executed" in v else "")
              for k, v in kg.items()),
    100 * mf["forward"]["mfma_busy_time_weighted"],
    [r for r in sw["rows"] if r["batch"] == 16][0]["images_per_sec"], [r for r in sw["rows"] if r["batch"] == 16][0]["rate_vs_largest_batch"],
    [r for r in sw["rows"] if r["batch"] == 10][0]["images_per_sec"], [r for r in sw["rows"] if r["batch"] == 8][0]["images_per_sec"],
    [r for r in sw["rows"] if r["batch"] == 1][0]["images_per_sec"],
    lf["device_prep"]["images_per_sec"], lu["device_prep"]["images_per_sec"], lf.get("usable_cpus", 16), stg["ucb_post"]["items_per_sec"],
    b["single_stream_value"], b["value"], 100 * dom["mfma_busy"])
dpath = os.path.join(ROOT, "DESIGN.md")
ds = open(dpath).read()
if "<!-- BEGIN r4 DESIGN -->" in ds:
    ds = re.sub(r"<!-- BEGIN r4 DESIGN -->.*?<!-- END r4 DESIGN -->", lambda m: design, ds, flags=re.S)
else:
    ds = ds.replace("<!-- BEGIN MEASUREMENTS -->", "<!-- BEGIN MEASUREMENTS -->\n" + design)
open(dpath, "w").write(ds)
print(design[:3000])
