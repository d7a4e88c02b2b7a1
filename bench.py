#!/usr/bin/env python3
"""bench.py — images/sec of the GSC generator batched forward at 256x256 on N MI355X (BASELINE.json).

A step = one forward of the hot path over one synthetic batch of 32 images per GPU (BASELINE configs[1]:
"Batch=32 synthetic 256x256x3, full GSC generator fp32"), inputs already resident in HBM.  For N > 1 the
batch shards one-process-per-GPU (weak scaling: 32 images per rank) and each step ends with the RCCL
all-gather that re-assembles the consumed outputs (con_rgb + dif) on every rank, overlapped with the next
step's compute.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--dtype f32|f16] [--workload gsc256|tsm512]
                    [--no-cpu-baseline] [--loop ffhq|ucb]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes this process a LAUNCHER: it starts N rank
processes of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set) before anything touches
the GPU, relays rank 0's JSON line and exits non-zero if any rank fails.  Under `torch.distributed.run` the
ranks already exist and the launcher is skipped.  `--backend gloo --stub` runs the same rank logic on CPU
with a stand-in generator (tests/test_bench_cpu.py): it checks the launcher, sharding, all-gather overlap
bookkeeping and the JSON contract, and is labelled `"stub": true` — never a measurement.
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMAGE = 18.104        # 9 052.06 MMAC per 256x256 image
GFLOP_3X3_PER_IMAGE = 11.017    # 3x3 conv + transposed 3x3 ("3x3-conv path")
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, = fp32 vector peak
PEAK_F16_MFMA_TFLOPS = 2500.0   # same guide: BF16/F16 MFMA dense (only used for the opt-in --dtype f16 line)

# Algorithmic MMAC per 256x256 image of every launch of the forward, by the layer name bsr_timing_entry reports
# (fused launches carry the sum of the reference layers they compute).
_RES_CONV1 = (12.98, 33.69, 33.69, 34.21, 34.21, 34.21)
LAYER_MMAC = {"conv1": 308.28, "down1": 301.99, "down2": 150.99, "down3": 56.62, "up1": 227.38, "up2": 377.49, "up3": 1207.96,
              "heads": 2 * 205.52, "clr_up1": 307.89, "clr_up2": 452.98, "clr_up3": 905.97, "clr_conv1": 613.42 + 16.78 + 3.15}
for _i in range(6):
    LAYER_MMAC["res%d.conv1" % _i] = _RES_CONV1[_i]
    LAYER_MMAC["res%d.conv2" % _i] = 150.99
    LAYER_MMAC["res%d.c3q" % _i] = 4 * 33.69            # conv3 + theta | phi | g (composed offline into one K = 128 GEMM)
    LAYER_MMAC["res%d.attention" % _i] = 2 * 134.22
    LAYER_MMAC["res%d.w" % _i] = 33.69
# Algorithmic activation ELEMENTS per 256x256 image of each launch, (read, written): every input element read once + every output
# element written once (weights, 12 MB in all, stay in L2).  x bytes per element x images / time = the algorithmic HBM rate.
def _io(hw_in, c_in, hw_out, c_out):
    return (hw_in * hw_in * c_in, hw_out * hw_out * c_out)


LAYER_IO_ELEMS = {"conv1": _io(256, 3, 256, 32), "down1": _io(256, 32, 128, 64), "down2": _io(128, 64, 64, 64), "down3": _io(64, 64, 32, 96),
                  "up1": _io(32, 257, 64, 96), "up2": _io(64, 160, 128, 64), "up3": _io(128, 128, 256, 64), "heads": _io(256, 64, 256, 16),
                  "clr_up1": _io(32, 261, 64, 128), "clr_up2": _io(64, 128, 128, 96), "clr_up3": _io(128, 96, 256, 64),
                  "clr_conv1": _io(256, 64 + 1 + 3, 256, 4)}
for _i in range(6):
    _cin = (99, 257, 257, 261, 261, 261)[_i]
    LAYER_IO_ELEMS["res%d.conv1" % _i] = _io(32, _cin, 32, 128)
    LAYER_IO_ELEMS["res%d.conv2" % _i] = _io(32, 128, 32, 128)
    LAYER_IO_ELEMS["res%d.c3q" % _i] = _io(32, 128 + _cin, 32, 288 + 384)          # + the block input (skip folded into y3x)
    LAYER_IO_ELEMS["res%d.attention" % _i] = _io(32, 384, 32, 128)
    LAYER_IO_ELEMS["res%d.w" % _i] = _io(32, 128 + 288, 32, 264)
# Launches that compute SEVERAL reference layers (round 4: at full batches the `w` GEMM is the tail of the attention kernel): priced with
# the sum of their parts; LAYER_MMAC stays the per-layer table of SURVEY Appendix C.
FUSED_LAUNCHES = {"res%d.attw" % _i: ("res%d.attention" % _i, "res%d.w" % _i) for _i in range(6)}
for _i in range(6):
    LAYER_IO_ELEMS["res%d.attw" % _i] = _io(32, 384 + 288, 32, 264)          # qkv + y3x in, block output out (att never reaches HBM)


def launch_mmac(name):
    parts = FUSED_LAUNCHES.get(name)
    return sum(LAYER_MMAC[p] for p in parts) if parts else LAYER_MMAC[name]


# MMAC the kernel really EXECUTES per image where that differs from the reference's op count: conv3 and theta|phi|g are composed offline
# into ONE K = 128 GEMM with N = 288 + 384 (pack.py), so the launch does 1024 x 128 x 672 MACs for the 4 x 33.69 MMAC of the four
# reference convs — its fraction "by algorithmic FLOPs" can exceed 1; `frac_executed` prices the instructions it issues
EXECUTED_MMAC = {"res%d.c3q" % _i: 1024 * 128 * 672 / 1e6 for _i in range(6)}


def launch_mmac_executed(name):
    parts = FUSED_LAUNCHES.get(name)
    if parts:
        return sum(EXECUTED_MMAC.get(p, LAYER_MMAC[p]) for p in parts)
    return EXECUTED_MMAC.get(name, LAYER_MMAC[name])


# f16 mode: bytes per element of each launch's (input, output) tensor — the fp16 activation pack; everything else 4 / 4
F16_IO_BYTES = {"conv1": (4, 2), "down1": (2, 2), "down2": (2, 2), "down3": (2, 4), "up1": (4, 2), "up2": (2, 2), "up3": (2, 2), "heads": (2, 4),
                "clr_up1": (4, 2), "clr_up2": (2, 2), "clr_up3": (2, 2), "clr_conv1": (2, 4)}


def layer_bytes(name, dtype):
    bi, bo = F16_IO_BYTES.get(name, (4, 4)) if dtype == "f16" else (4, 4)
    ei, eo = LAYER_IO_ELEMS[name]
    return ei * bi + eo * bo


PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def workload_tables(tsm, hw):
    """(MMAC per image by launch name, activation elements (read, written) per image, executed MMAC of the composed GEMMs) of one
    hw x hw image, from the layer table itself: GSC (/root/reference/model.py:198-226) or the TSM variant
    (/root/reference/model_with_TSM.py:231-259, :261-325: res0 sees cat[x 96 | x_share 192 | uv 3] = 291 channels, blocks 3-5
    cat[x_hole 291 | bmask | x_share 582 | uv 3] = 877; up1 291 -> 96, clr_up1 877 -> 128).  Attention is quadratic in the
    (hw/8)^2 tokens: two tokens x tokens x 128 contractions per block (model.py:51-53)."""
    h1, h2, h4, h8 = hw, hw // 2, hw // 4, hw // 8
    t = h8 * h8
    c_res = [291, 291, 291, 877, 877, 877] if tsm else [99, 257, 257, 261, 261, 261]      # input channels of each block
    c_up1, c_clr1 = (291, 877) if tsm else (257, 261)
    mm = lambda pix, k, cin, cout: pix * k * cin * cout / 1e6
    m = {"conv1": mm(h1 * h1, 49, 3, 32), "down1": mm(h2 * h2, 9, 32, 64), "down2": mm(h4 * h4, 9, 64, 64), "down3": mm(h8 * h8, 9, 64, 96),
         "up1": mm(t, 9, c_up1, 96), "up2": mm(h4 * h4, 9, 160, 64), "up3": mm(h2 * h2, 9, 128, 64), "heads": 2 * mm(h1 * h1, 49, 64, 1),
         "clr_up1": mm(t, 9, c_clr1, 128), "clr_up2": mm(h4 * h4, 9, 128, 96), "clr_up3": mm(h2 * h2, 9, 96, 64),
         "clr_conv1": mm(h1 * h1, 9, 65, 16) + mm(h1 * h1, 1, 16, 16) + mm(h1 * h1, 1, 16, 3)}
    io = {"conv1": (h1 * h1 * 3, h1 * h1 * 32), "down1": (h1 * h1 * 32, h2 * h2 * 64), "down2": (h2 * h2 * 64, h4 * h4 * 64), "down3": (h4 * h4 * 64, t * 96),
          "up1": (t * c_up1, h4 * h4 * 96), "up2": (h4 * h4 * 160, h2 * h2 * 64), "up3": (h2 * h2 * 128, h1 * h1 * 64), "heads": (h1 * h1 * 64, h1 * h1 * 16),
          "clr_up1": (t * c_clr1, h4 * h4 * 128), "clr_up2": (h4 * h4 * 128, h2 * h2 * 96), "clr_up3": (h2 * h2 * 96, h1 * h1 * 64),
          "clr_conv1": (h1 * h1 * (64 + 1 + 3), h1 * h1 * 4)}
    ex = {}
    for i in range(6):
        cin = c_res[i]
        m["res%d.conv1" % i] = mm(t, 1, cin, 128)
        m["res%d.conv2" % i] = mm(t, 9, 128, 128)
        m["res%d.c3q" % i] = 4 * mm(t, 1, 128, 257)  # conv3 + theta | phi | g (composed offline into one K = 128 GEMM)
        m["res%d.attention" % i] = 2 * mm(t, 1, t, 128)
        m["res%d.w" % i] = mm(t, 1, 128, 257)
        ex["res%d.c3q" % i] = mm(t, 1, 128, 672)
        io["res%d.conv1" % i] = (t * cin, t * 128)
        io["res%d.conv2" % i] = (t * 128, t * 128)
        io["res%d.c3q" % i] = (t * (128 + min(cin, 288)), t * (288 + 384))          # + the block input (skip folded into y3x)
        io["res%d.attention" % i] = (t * 384, t * 128)
        n_out = 288 if tsm else 264                   # channels the `w` GEMM stores: the block output's stride (264), at most its 9 tiles
        io["res%d.w" % i] = (t * (128 + 288), t * n_out)
        io["res%d.attw" % i] = (t * (384 + 288), t * n_out)                          # qkv + y3x in, block output out (att never reaches HBM)
    return m, io, ex


def set_workload(tsm, hw):
    """Re-price every launch for another workload (bench.py --workload tsm512): the tables are rebuilt IN PLACE, so every user of
    LAYER_MMAC / LAYER_IO_ELEMS / EXECUTED_MMAC / GFLOP_PER_IMAGE sees the frames actually run."""
    global GFLOP_PER_IMAGE, GFLOP_3X3_PER_IMAGE
    m, io, ex = workload_tables(tsm, hw)
    LAYER_MMAC.clear(); LAYER_MMAC.update(m)
    LAYER_IO_ELEMS.clear(); LAYER_IO_ELEMS.update(io)
    EXECUTED_MMAC.clear(); EXECUTED_MMAC.update(ex)
    GFLOP_PER_IMAGE = 2e-3 * sum(m.values())
    px = hw * hw
    GFLOP_3X3_PER_IMAGE = 2e-3 * (sum(m[n] for n in LAYERS_3X3) - px * (16 * 16 + 16 * 3) / 1e6)

# the "3x3-conv path" of north_star / SURVEY §8d: 3x3, stride-2 3x3 and transposed 3x3 layers (clr_conv1's launch also carries the fused 1x1 tail)
LAYERS_3X3 = (["down1", "down2", "down3", "up1", "up2", "up3", "clr_up1", "clr_up2", "clr_up3", "clr_conv1"] + ["res%d.conv2" % i for i in range(6)])
# kernel instantiation -> the layers it runs (csrc/bsr_api.hip launch table)
KERNEL_GROUPS = {
    "igemm_conv_kernel<3,3,1,TR,4,32,4,1,1,NI=2,CC=32> (transposed 3x3: up2, up3, clr_up3)": ["up2", "up3", "clr_up3"],
    "igemm_conv_kernel<3,3,1,TR,...> other instantiations (up1, clr_up1, clr_up2)": ["up1", "clr_up1", "clr_up2"],
    "igemm_conv_kernel<3,3,1> (res*.conv2)": ["res%d.conv2" % i for i in range(6)],
    "igemm_conv_kernel<3,3,2> (down1-3)": ["down1", "down2", "down3"],
    "nonlocal_attention_kernel": ["res%d.attention" % i for i in range(6)],
    "nonlocal_attention_kernel<4, FUSEW> (res*.attention + res*.w tail)": ["res%d.attw" % i for i in range(6)],
    "gemm_nloop_kernel (res*.c3q, res*.w)": ["res%d.%s" % (i, n) for i in range(6) for n in ("c3q", "w")],
    "igemm_conv_kernel<1,1,1> (res*.conv1)": ["res%d.conv1" % i for i in range(6)],
    "conv_n16_kernel<3,3,GS,TAIL> (clr_conv1 + clr_conv2 + clr_conv3 + dif)": ["clr_conv1"],
    "conv_n16_kernel<7,1> (heads conv2|conv3)": ["heads"],
    "stem7_kernel (conv1)": ["conv1"],
}


# ----------------------------------------------------------------------------------------------- launcher
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, argv, check_devices: bool, one_device=None) -> int:
    """Start n rank processes of this script and relay rank 0's stdout.  Runs BEFORE this process touches the GPU
    (torch.cuda.device_count() does not initialise it); never exec()s."""
    if check_devices:
        import torch
        have = torch.cuda.device_count()
        if one_device is not None:
            if have <= one_device:
                sys.stderr.write("bench.py: --device %d but only %d GPU(s) are visible\n" % (one_device, have))
                return 2
        elif have < n:
            sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) are visible: refusing to fall back to fewer ranks\n" % (n, have))
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    rc = 0
    out0 = b""
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, code))
                    for q in pending:
                        procs[q].terminate()        # exact PIDs of our own children
            if pending:
                if 0 in pending and procs[0].stdout is not None:
                    # rank 0 prints one short line at the very end: read it without blocking the poll loop for long
                    import select
                    rd, _, _ = select.select([procs[0].stdout], [], [], 0.2)
                    if rd:
                        chunk = os.read(procs[0].stdout.fileno(), 1 << 16)
                        out0 += chunk
                else:
                    time.sleep(0.2)
        rest = procs[0].stdout.read() if procs[0].stdout is not None else b""
        out0 += rest or b""
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    if rc == 0 and not out0.strip():
        sys.stderr.write("bench.py: rank 0 printed nothing\n")
        rc = 1
    return rc


# ----------------------------------------------------------------------------------------------- cpu baseline
def physical_cores() -> int:
    """Physical cores of the host (lscpu: sockets x cores per socket); falls back to os.cpu_count()."""
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        vals = {}
        for line in txt.splitlines():
            k, _, v = line.partition(":")
            vals[k.strip()] = v.strip()
        n = int(vals["Socket(s)"]) * int(vals["Core(s) per socket"])
        if n > 0:
            return n
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_baseline(weights, gen=None, device=None, budget_s=75.0):
    """Oracle (torch-CPU restatement of the reference's TF graph; TF itself is not installable here) timed on the host
    cores of this box on BOUNDED samples of the same synthetic workload: 1 thread, all physical cores
    (lscpu) and a short sweep in between (oneDNN does not scale monotonically on a 2-socket host).  `value` is the best
    point, `cores` the thread count that produced it; every point is listed."""
    import torch
    from oracle.gsc_oracle import GeneratorOracle
    from blindshadowremoval_amd.dataset import usable_cpus
    logical, phys, usable = os.cpu_count() or 1, physical_cores(), usable_cpus()
    oracle = GeneratorOracle(weights)
    torch.manual_seed(0)
    inp32, uv32 = torch.rand(32, 256, 256, 3), torch.rand(32, 256, 256, 3)
    t_all = time.perf_counter()
    points = []

    def measure(threads, b, max_runs, share):
        """median of up to max_runs forwards of the first b images after one warm-up, inside `share` of the budget"""
        torch.set_num_threads(threads)
        t_start = time.perf_counter()
        ts = []
        for i in range(max_runs + 1):
            t0 = time.perf_counter()
            oracle(inp32[:b], uv32[:b])
            dt = time.perf_counter() - t0
            if i > 0 or max_runs == 0:
                ts.append(dt)
            if time.perf_counter() - t_start > share * budget_s and ts:
                break
        ts.sort()
        med = ts[len(ts) // 2]
        points.append({"threads": threads, "batch": b, "forwards": len(ts), "images_per_sec": round(b / med, 3)})

    # The host cores this job may USE are the cgroup CPU quota / affinity (round 3: the GPU box shows 256 logical CPUs but throttles
    # the container to 16 — which is why "all 128 physical cores" measured SLOWER than one thread in round 2): the sweep is 1 thread,
    # half the usable CPUs, all of them, and — for the record — twice that many.
    measure(1, 2, 2, 0.15)                                  # 1 thread: ~1 image/s, so a 2-image sample
    for th in sorted({max(2, usable // 2), 2 * usable} - {1, usable}):
        if time.perf_counter() - t_all > 0.5 * budget_s:
            break
        measure(th, 8, 2, 0.12)
    measure(usable, 32, 3, 0.3)                             # every usable CPU on the full configs[1] batch
    best = max(points, key=lambda p: p["images_per_sec"])
    out = {"value": best["images_per_sec"], "unit": "images/sec", "cores": best["threads"], "kind": "port",
           "usable_cpus": usable, "physical_cores": phys, "logical_cpus": logical, "points": points,
           "sample": "oracle-CPU (torch/oneDNN fp32 restatement of model.py, proxy for the TF2-CPU path) on synthetic 256x256 images: "
                     "1 thread x 2 images, half / twice the usable CPUs x 8 images, all %d usable CPUs (cgroup quota / affinity; the box shows %d "
                     "logical CPUs on %d physical cores) x the 32-image configs[1] batch; median forward per point, best point reported" % (usable, logical, phys)}
    if gen is not None:
        # the checker role of the oracle (BASELINE metric: "... PSNR vs TF2 ref"): the HIP outputs of an 8-image sample against it,
        # compared given the same 32x32 threshold mask (SURVEY F7 protocol, tests/parity_util.py)
        import math
        torch.set_num_threads(best["threads"])
        inp, uv = inp32[:8], uv32[:8]
        hip = [t.cpu() for t in gen(inp.to(device), uv.to(device))]
        bmask = gen.probe("bmask").cpu()
        pr = {}
        oracle(inp, uv, probes=pr)
        flips = int((bmask != pr["bmask"]).sum())
        ref = oracle(inp, uv, bmask_override=bmask)
        err = max(float((a - r).abs().max()) for a, r in zip(hip, ref))
        mse = float(((hip[1].double().clamp(0, 1) - ref[1].double().clamp(0, 1)) ** 2).mean())
        out["parity"] = {"max_abs_err": err, "psnr_db_con_rgb": (round(-10.0 * math.log10(mse), 2) if mse > 0 else None),
                         "bmask_flips": flips, "sample": "8 images of the CPU sample, all four outputs"}
    return out


# ----------------------------------------------------------------------------------------------- stub generator (CPU tests only)
class _StubGenerator:
    """Stand-in for the HIP generator so the rank logic (launcher, sharding, double-buffered all-gather, JSON contract)
    runs on a CPU box over gloo.  NOT a fallback of the product path: bench lines produced with it carry "stub": true."""

    def __call__(self, inputs, uv, out=None, packed_out=None):
        import torch
        g0 = inputs.mean(dim=3, keepdim=True)
        res = (g0, inputs * 0.5 + uv * 0.5, torch.cat([g0, g0 * 0, -g0], 3), g0 - uv[..., :1])
        if packed_out is not None:
            packed_out[..., :3].copy_(res[1])
            packed_out[..., 3:].copy_(res[3])
            return res[0], packed_out[..., :3], res[2], packed_out[..., 3:]
        if out is not None:
            for o, r in zip(out, res):
                o.copy_(r)
            return out
        return res


# ----------------------------------------------------------------------------------------------- roofline from per-launch events
H16_LAYERS = (["down1", "down2", "down3", "up1", "up2", "up3", "clr_up1", "clr_up2", "clr_up3"] + ["res%d.conv2" % i for i in range(6)])   # = pack.H16_LAYERS


# layers that run split-precision (hi/lo fp16 planes, three fp16 matrix instructions per K group) in BOTH 16-bit modes = pack.X3_LAYERS + attention
X3_LAYERS = (["res%d.%s" % (i, n) for i in range(6) for n in ("conv1", "c3q", "w", "attention", "attw")] + ["heads", "clr_conv1", "conv1"])


def group_peak(layers, dtype):
    """Matrix-core peak that bounds a kernel group: fp32 MFMA for fp32 kernels (every kernel of dtype f32; the stem in all modes); in
    the 16-bit modes the dense fp16 peak — divided by 3 where an algorithmic MAC costs three fp16 MACs (hi.hi + hi.lo + lo.hi):
    every 16-bit kernel of f32x3, and the 1x1 / attention / 16-channel kernels of f16 too."""
    if dtype != "f32" and all(n in H16_LAYERS for n in layers):
        return PEAK_F16_MFMA_TFLOPS / (3.0 if dtype == "f32x3" else 1.0)
    if dtype != "f32" and all(n in X3_LAYERS for n in layers):
        return PEAK_F16_MFMA_TFLOPS / 3.0
    return PEAK_F32_MFMA_TFLOPS


def roofline_from_events(gen, run_once, B, dtype, n_rep=3):
    """HIP events around every launch of a few extra forwards on the forward's stream (bsr_set_timing): per-layer device time ->
    the dominant kernel instantiation's achieved TFLOP/s, the 3x3-conv path's, and every kernel group's fraction of the peak."""
    import torch
    gen.set_timing(True)
    layer_ms = {}
    for _ in range(n_rep):
        run_once()
        torch.cuda.synchronize()
        for name, ms, _cls in gen.get_launch_timing():
            layer_ms[name] = layer_ms.get(name, 0.0) + ms / n_rep
    gen.set_timing(False)
    groups = {}
    kgroups = dict(KERNEL_GROUPS)
    if dtype != "f32":
        # the 16-bit kernels use 32-channel K chunks everywhere, so clr_up1 (261 -> 288 channels, NI = 2) shares the dominant instantiation
        dom_key = [k for k in kgroups if "up2, up3, clr_up3" in k][0]
        oth_key = [k for k in kgroups if "other instantiations" in k][0]
        kgroups.pop(dom_key)
        kgroups.pop(oth_key)
        if dtype == "f16":       # round 6: the f16 mode's stride-1 and transposed 3x3 layers run on conv3_f16_kernel (csrc/conv3_f16.h), all six transposed layers on ONE instantiation
            kgroups.pop([k for k in kgroups if "(res*.conv2)" in k][0])
            kgroups["conv3_f16_kernel<TR> (transposed 3x3: up1, up2, up3, clr_up1, clr_up2, clr_up3)"] = ["up1", "up2", "up3", "clr_up1", "clr_up2", "clr_up3"]
            kgroups["conv3_f16_kernel<S1> (res*.conv2)"] = ["res%d.conv2" % i for i in range(6)]
        else:
            kgroups["igemm_conv_kernel<3,3,1,TR,4,32,4,1,1,NI=2,CC=32> (transposed 3x3: up2, up3, clr_up3, clr_up1)"] = ["up2", "up3", "clr_up3", "clr_up1"]
            kgroups["igemm_conv_kernel<3,3,1,TR,4,32,4,1,1,NI=1,CC=32> (transposed 3x3: up1, clr_up2)"] = ["up1", "clr_up2"]
    for gname, layers in kgroups.items():
        layers = [n for n in layers if n in layer_ms]        # a group's layers that ran as their own launch in this forward (others: a fused launch's group)
        ms = sum(layer_ms[n] for n in layers)
        if ms <= 0:
            continue
        gflop = 2e-3 * sum(launch_mmac(n) for n in layers) * B
        gpeak = group_peak(layers, dtype)
        label = gname
        if gpeak != PEAK_F32_MFMA_TFLOPS and "(res*.conv1)" in gname and B * 8 * 2 >= 256:
            label = "gemm_nloop_kernel<4,NCH,H=2,MINW=1> (res*.conv1)"      # round 5: the resident-activation GEMM at full batches (csrc/gemm_nloop.h)
        elif gpeak != PEAK_F32_MFMA_TFLOPS:        # the 16-bit instantiations (csrc/igemm_h16.h, attention_h16.h, gemm_nloop / conv_n16 with H = 2)
            label = (gname.replace("igemm_conv_kernel", "igemm_h16_kernel").replace("nonlocal_attention_kernel<4, FUSEW>", "nonlocal_attention_h16_kernel<FUSEW>").replace("nonlocal_attention_kernel", "nonlocal_attention_h16_kernel")
                     .replace("gemm_nloop_kernel", "gemm_nloop_kernel<..,H=2>").replace("conv_n16_kernel<", "conv_n16_kernel<H=2,")
                     .replace("stem7_kernel", "stem7_kernel<4,H=2>"))
        gbytes = 1e-9 * sum(layer_bytes(n, dtype) for n in layers) * B
        groups[label] = {"ms": round(ms, 4), "launches": len(layers), "tflops": round(gflop / ms, 2), "peak": round(gpeak, 1),
                         "peak_kind": ("hardware roof: fp32 matrix pipe" if gpeak == PEAK_F32_MFMA_TFLOPS else
                                       "hardware roof: dense fp16 matrix pipe" if gpeak == PEAK_F16_MFMA_TFLOPS else
                                       "emulation-adjusted ceiling, NOT a hardware roof: the fp16 matrix peak / 3 (one algorithmic MAC = three fp16 MACs: hi.hi + hi.lo + lo.hi)"),
                         "frac": round(gflop / ms / gpeak, 4), "alg_GBps": round(gbytes / ms * 1e3, 1), "hbm_frac": round(gbytes / ms * 1e3 / PEAK_HBM_GBPS, 4),
                         "gflop": gflop}
        gexec = 2e-3 * sum(launch_mmac_executed(n) for n in layers) * B
        if abs(gexec - gflop) > 1e-6 * gflop:        # composed weights: fewer MACs issued than the reference's op count
            groups[label]["frac_executed"] = round(gexec / ms / gpeak, 4)
            groups[label]["note"] = ("`frac` is by the reference's op count (%.1f GFLOP); the launch issues %.1f GFLOP (conv3 and theta|phi|g composed offline into "
                                     "one K = 128 GEMM): `frac_executed`" % (gflop, gexec))
    dom_name = max(groups, key=lambda k: groups[k]["ms"])
    dom = groups[dom_name]
    peak = dom["peak"]
    # the "3x3-conv path": every launch that computes a 3x3 layer.  A fused launch that contains one (none since the conv2 + GEMM-tail form
    # was retired in round 6) cannot be split, so it would enter with ALL its time and ALL its algorithmic work
    t33 = sum(layer_ms.get(n, 0.0) for n in LAYERS_3X3)
    gflop33 = GFLOP_3X3_PER_IMAGE * B
    fused33 = [n for n, parts in FUSED_LAUNCHES.items() if n in layer_ms and any(p_ in LAYERS_3X3 for p_ in parts)]
    for n in fused33:
        t33 += layer_ms[n]
        gflop33 += 2e-3 * B * sum(LAYER_MMAC[p_] for p_ in FUSED_LAUNCHES[n] if p_ not in LAYERS_3X3)
    path = gflop33 / t33
    peak33 = group_peak(["up3"], dtype)
    t_all = sum(layer_ms.values())
    glue_ms = sum(ms for n, ms in layer_ms.items() if n not in LAYER_MMAC and n not in FUSED_LAUNCHES)
    # the dominant kernel is priced against both roofs; `bound` names the nearer one (fp32 kernels: the fp32 matrix pipe; the 16-bit
    # kernels of f32x3 / f16: HBM once the matrix work has shrunk by 16/3 or 16)
    mfma_view = {"achieved_TFLOPs": dom["tflops"], "peak_TFLOPs": peak, "frac": dom["frac"], "peak_kind": dom["peak_kind"]}
    hbm_view = {"alg_GBps": dom["alg_GBps"], "peak_GBps": PEAK_HBM_GBPS, "frac": dom["hbm_frac"],
                "note": "algorithmic activation bytes (input read once + output written once) / device time of the same launches"}
    hbm_bound = dom["hbm_frac"] > dom["frac"]
    rf = {"mode": "one forward at a time (one handle, one stream): every per-kernel figure of this object",
          "bound": "hbm" if hbm_bound else "mfma", "achieved": dom["alg_GBps"] if hbm_bound else dom["tflops"],
          "peak": PEAK_HBM_GBPS if hbm_bound else peak, "unit": "GB/s" if hbm_bound else "TFLOP/s",
          "frac": dom["hbm_frac"] if hbm_bound else dom["frac"], "traffic": None, "mfma_view": mfma_view, "hbm_view": hbm_view,
          "kernel": dom_name + " — the largest kernel instantiation, %.0f %% of the forward's device time" % (100 * dom["ms"] / t_all),
          "launches_per_forward": dom["launches"], "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
          "duration_source": ("HIP events recorded around every launch on the forward's stream (bsr_set_timing), mean of %d extra forwards after the timed "
                              "region; the rocprofv3 --kernel-trace average of the same kernel is in profiles/ (3-4 %% longer: profiler overhead); event-timed "
                              "launches carry ~1-2 us of event overhead each, so all_kernels_ms slightly exceeds ms_per_step" % n_rep),
          "algorithmic_gflop_per_launch": round(dom["gflop"] / dom["launches"], 2),
          "path_3x3": {"achieved": round(path, 2), "peak": round(peak33, 1), "frac": round(path / peak33, 4), "unit": "TFLOP/s",
                       "launches": sum(1 for n in LAYERS_3X3 if n in layer_ms) + len(fused33), "ms": round(t33, 4),
                       "algorithmic_gflop": round(gflop33, 2),
                       "note": ("%d of the launches are res*.conv2 fused with the conv3 | theta|phi|g GEMM: counted with their whole time and the GEMM's work "
                                "(%.1f GFLOP beyond the path's %.1f)" % (len(fused33), gflop33 - GFLOP_3X3_PER_IMAGE * B, GFLOP_3X3_PER_IMAGE * B)) if fused33 else None},
          "all_kernels_tflops": round(GFLOP_PER_IMAGE * B / t_all, 2), "all_kernels_ms": round(t_all, 4),
          "all_kernels_tflops_executed": round(2e-3 * sum(launch_mmac_executed(n) for n in layer_ms if n in LAYER_MMAC or n in FUSED_LAUNCHES) * B / t_all, 2),
          "all_kernels_frac": (round(GFLOP_PER_IMAGE * B / t_all / PEAK_F32_MFMA_TFLOPS, 4) if dtype == "f32" else None),
          "all_kernels_frac_executed": (round(2e-3 * sum(launch_mmac_executed(n) for n in layer_ms if n in LAYER_MMAC or n in FUSED_LAUNCHES) * B / t_all / PEAK_F32_MFMA_TFLOPS, 4)
                                        if dtype == "f32" else None),
          "all_kernels_note": ("whole forward, every launch incl. glue: `all_kernels_tflops` by the reference's op count (%.3f GFLOP per image), `_executed` by the "
                               "matrix work the launches issue (conv3 and theta|phi|g are composed offline into one K = 128 GEMM: fewer MACs)" % GFLOP_PER_IMAGE),
          "kernel_groups": {k: {kk: vv for kk, vv in v.items() if kk != "gflop"} for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])},
          "glue_ms": round(glue_ms, 4)}
    return rf, dom_name


def roofline_in_flight(gens, lanes, run_on, B, dtype, dom_name, ms_per_step, n_rep=3):
    """The dominant kernel and the whole forward in the mode `value` is taken in — two forwards in flight on two handles / streams:
    HIP events around every launch of BOTH handles while steps alternate between them (an event pair then brackets the kernel's
    residency including what the other lane's kernels take from it), and the whole-forward rate from the timed region itself."""
    import torch
    layers = None
    for key, names in KERNEL_GROUPS.items():
        if key == dom_name or dom_name.startswith(key):
            layers = names
    if layers is None:
        return None
    for g in gens:
        g.set_timing(True)
    tot, cnt = 0.0, 0
    for _ in range(n_rep):
        for k in range(len(gens)):
            run_on(k)
        torch.cuda.synchronize()
        for g in gens:
            for name, ms, _cls in g.get_launch_timing():
                if name in layers:
                    tot += ms
                    cnt += 1
    for g in gens:
        g.set_timing(False)
    if cnt == 0:
        return None
    avg = tot / cnt
    gflop = 2e-3 * sum(launch_mmac(n) for n in layers) * B / len(layers)
    whole = GFLOP_PER_IMAGE * B / ms_per_step
    return {"mode": "two forwards in flight (two handles, two HIP streams) = the mode of `two_in_flight`, not of `value`",
            "whole_forward": {"achieved": round(whole, 2), "peak": PEAK_F32_MFMA_TFLOPS if dtype == "f32" else None, "unit": "TFLOP/s",
                              "frac": round(whole / PEAK_F32_MFMA_TFLOPS, 4) if dtype == "f32" else None,
                              "note": "18.104 GFLOP x images / ms_per_step of the timed region: the chip-level figure of this mode"},
            "dominant_kernel": {"kernel": dom_name, "avg_launch_ms": round(avg, 4), "launches_timed": cnt, "algorithmic_gflop_per_launch": round(gflop, 2),
                                "note": "event-bracketed residency of a launch while the OTHER lane's kernels share the chip with it (the rocprofv3 trace of the "
                                        "same mode: profiles/r4_lane_overlap.txt, median 704 us against 410 alone) — a duration, not a rate: the kernel's own "
                                        "roofline fraction is `roofline.frac`, taken one forward at a time"}}


# kernel-group label (KERNEL_GROUPS / the 16-bit relabelling) -> substring of the rocprofv3 kernel names of that group
GROUP_KERNEL_KEY = (("conv3_f16_kernel<TR>", "conv3_f16_kernel<true"), ("conv3_f16_kernel<S1>", "conv3_f16_kernel<false"),
                    ("up2, up3, clr_up3", "<3, 3, 1, true, 4, 32, 4, 1, 1, 2, 32"), ("(res*.c3q", "gemm_nloop_kernel<3,"), ("attention", "attention"),
                    ("(res*.conv2)", "<3, 3, 1, false"), ("(down1-3)", "<3, 3, 2, false"), ("clr_conv1", "conv_n16_kernel<3, 3"),
                    ("heads", "conv_n16_kernel<7, 1"), ("stem7", "stem7_kernel"), ("(res*.conv1)", "<1, 1, 1, false"), ("(res*.conv1)", "gemm_nloop_kernel<4,"),
                    ("up1, clr_up2", "<3, 3, 1, true, 4, 32, 4, 1, 1, 1, 32"), ("other instantiations", "<3, 3, 1, true, 4, 32, 4, 1, 1, 1,"),
                    ("other instantiations", "<3, 3, 1, true, 4, 32, 4, 1, 1, 2, 24"))


def attach_traffic(rf, dom_name, B, dtype):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/pmc_traffic.py; separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE runs as MI355X_MICROARCH.md prescribes).  Reported only while the kernel sources still hash to what
    the passes were taken on; otherwise null with the reason."""
    from blindshadowremoval_amd.build import source_sha16
    sha = source_sha16()
    sfx = "" if dtype == "f32" else "_" + dtype
    for tag in ("r6", "r5", "r4", "r3", "r2", "r1"):
        tpath = os.path.join(ROOT, "profiles", "%s_pmc_traffic%s.json" % (tag, sfx))
        if not os.path.isfile(tpath):
            continue
        with open(tpath) as ft:
            t = json.load(ft)
        key = [k for g, k in GROUP_KERNEL_KEY if g in dom_name]
        rows = [v for k, v in (t.get("per_kernel") or {}).items() if key and key[0] in k]
        if B != t.get("batch") or t.get("kernel_src_sha16") != sha:
            rf["traffic_note"] = ("profiles/%s_pmc_traffic%s.json was measured on kernel sources %s / batch %s, this build is %s / batch %d: not reported"
                                  % (tag, sfx, t.get("kernel_src_sha16"), t.get("batch"), sha, B))
        elif not rows:
            rf["traffic_note"] = "profiles/%s_pmc_traffic%s.json has no rows for the dominant kernel of this run" % (tag, sfx)
        else:
            n = sum(v["launches_per_forward"] for v in rows)
            rf["traffic"] = sum(v["hbm_bytes_per_forward"] for v in rows) / max(n, 1)
            rf["traffic_note"] = ("HBM bytes per launch of the same kernel (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes, profiles/%s_pmc_traffic%s.csv: %d "
                                  "launches per forward, taken on kernel sources %s = this build; the memory-side counters include Infinity-Cache hits); "
                                  "algorithmic bytes per launch (input read once, output written once): %.4g"
                                  % (tag, sfx, n, sha, rf["hbm_view"]["alg_GBps"] * 1e9 * rf["avg_launch_ms"] * 1e-3))
        return


def attach_group_traffic(rf, B, dtype):
    """Every kernel group's HBM rate from the COUNTERS (the same committed passes as `traffic`: 2 x FETCH_SIZE + WRITE_SIZE per launch,
    summed over the group's launches of one forward) beside the algorithmic one: `counter_GBps`, `hbm_frac_counters` (of the 8 TB/s
    spec) and `traffic_ratio` = counter bytes / algorithmic bytes (re-reads show up here) — only while the kernel sources hash to what
    the passes ran on."""
    from blindshadowremoval_amd.build import source_sha16
    sfx = "" if dtype == "f32" else "_" + dtype
    tpath = next((os.path.join(ROOT, "profiles", "%s_pmc_traffic%s.json" % (tag, sfx)) for tag in ("r6", "r5", "r4")
                  if os.path.isfile(os.path.join(ROOT, "profiles", "%s_pmc_traffic%s.json" % (tag, sfx)))), None)
    if tpath is None:
        return
    with open(tpath) as ft:
        t = json.load(ft)
    if B != t.get("batch") or t.get("kernel_src_sha16") != source_sha16():
        return
    per = t.get("per_kernel") or {}
    for gname, g in rf["kernel_groups"].items():
        keys = [k for sub, k in GROUP_KERNEL_KEY if sub in gname]
        rows = [v for k, v in per.items() if any(key in k for key in keys)]
        if not rows or g["ms"] <= 0:
            continue
        nbytes = sum(v["hbm_bytes_per_forward"] for v in rows)
        g["counter_GBps"] = round(nbytes / g["ms"] * 1e-6, 1)
        g["hbm_frac_counters"] = round(nbytes / g["ms"] * 1e-6 / PEAK_HBM_GBPS, 4)
        if g.get("alg_GBps"):
            g["traffic_ratio"] = round(g["counter_GBps"] / g["alg_GBps"], 3)
    rf["kernel_groups_traffic_note"] = ("counter_GBps / hbm_frac_counters / traffic_ratio: HBM bytes of the group's launches from %s (separate rocprofv3 --pmc passes on "
                                        "these kernel sources) over the event-timed ms of this run.  read = 2 x FETCH_SIZE is calibrated for wide coalesced reads (16 B per lane over whole 128-B "
                                        "lines: MI355X_MICROARCH.md, HBM); kernels that fetch 32- / 64-byte pieces of a line per request (the stride-2 layers' 16-channel chunks, "
                                        "the 3x3 layers' halo columns) may be over-counted by up to 2x — read their ratios as upper bounds" % os.path.basename(tpath))


def attach_mfma(rf, dom_name, B, dtype):
    """Matrix-pipe utilisation and the clock the chip held, for the dominant kernel, from the committed counter passes
    (tools/pmc_mfma.py: separate rocprofv3 --pmc runs of SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE ...; profiles/r3_pmc_mfma*.json) —
    under the same rule as `traffic`: only while the kernel sources hash to what the passes ran on.  north_star asks for "MFMA
    utilisation against chip peak": `mfma_busy` = busy cycles / GPU-active cycles (utilisation at the clock actually held),
    `clock_ghz` = GRBM_GUI_ACTIVE / 8 XCDs / duration; frac (of the NOMINAL 2.4 GHz peak) ~= mfma_busy x clock_ghz / 2.4 x
    (algorithmic / issued FLOP)."""
    from blindshadowremoval_amd.build import source_sha16
    sha = source_sha16()
    sfx = "" if dtype == "f32" else "_" + dtype
    rf["mfma_busy"] = rf["clock_ghz"] = None
    mtag = next((t for t in ("r6", "r5", "r4", "r3") if os.path.isfile(os.path.join(ROOT, "profiles", "%s_pmc_mfma%s.json" % (t, sfx)))), None)
    if mtag is None:
        rf["mfma_note"] = "no profiles/r*_pmc_mfma%s.json" % sfx
        return
    mpath = os.path.join(ROOT, "profiles", "%s_pmc_mfma%s.json" % (mtag, sfx))
    with open(mpath) as fm:
        m = json.load(fm)
    key = [k for g, k in GROUP_KERNEL_KEY if g in dom_name]
    rows = [v for k, v in (m.get("per_kernel") or {}).items() if key and key[0] in k]
    if B != m.get("batch") or m.get("kernel_src_sha16") != sha:
        rf["mfma_note"] = ("profiles/%s_pmc_mfma%%s.json was measured on kernel sources %%s / batch %%s, this build is %%s / batch %%d: not reported" % mtag
                           % (sfx, m.get("kernel_src_sha16"), m.get("batch"), sha, B))
    elif not rows or "mfma_busy" not in rows[0]:
        rf["mfma_note"] = "profiles/%s_pmc_mfma%s.json has no matrix-pipe counters for the dominant kernel of this run" % (mtag, sfx)
    else:
        us = sum(r["us_per_forward"] for r in rows)
        rf["mfma_busy"] = round(sum(r["mfma_busy"] * r["us_per_forward"] for r in rows) / us, 4)
        rf["clock_ghz"] = round(sum(r["clock_ghz"] * r["us_per_forward"] for r in rows) / us, 3)
        rf["mfma_busy_of_nominal_clock"] = round(sum(r["mfma_busy_nominal"] * r["us_per_forward"] for r in rows) / us, 4)
        rf["mfma_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) and GRBM_GUI_ACTIVE / 8 / duration of the same kernel "
                           "(profiles/" + mtag + "_pmc_mfma%s.csv, separate rocprofv3 --pmc passes on kernel sources %s = this build; profiled dispatches are "
                           "serialised and clock 2-5 %% differently from the un-profiled run; the in-kernel s_memtime / s_memrealtime clock of the "
                           "same kernel is in profiles/r3_clock_stamps.txt)" % (sfx, sha))


def sustained_region(run_step, device_index, B, world, ms_per_step, warm_s=2.0, region_s=3.2):
    """A timed region that lasts SECONDS (`value` is K steps = ~0.1 s: the reference times whole runs, train_test_GSC.py:852,860): >= warm_s
    of back-to-back forwards untimed, then >= region_s timed, one forward at a time — with the clock the chip held during it, sampled from
    INSIDE the chip by a one-wave kernel on a side stream (bsr_clock_trace: shader cycles against the 100-MHz real-time counter)."""
    import torch
    from blindshadowremoval_amd import _lib
    lib = _lib.load()
    n_warm = max(1, int(warm_s * 1e3 / ms_per_step) + 1)
    n_reg = max(1, int(region_s * 1e3 / ms_per_step) + 1)
    samples = int((warm_s + region_s) * 2.5e4) + 4096              # one pair per ~100 us, with slack: the kernel is stopped by the flag
    dev = torch.device("cuda", device_index)
    buf = torch.zeros(2 * samples, dtype=torch.int64, device=dev)
    stop = torch.zeros(1, dtype=torch.int32, device=dev)
    taken = torch.zeros(1, dtype=torch.int32, device=dev)
    side, flag_stream = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    _lib.check(lib.bsr_clock_trace(device_index, buf.data_ptr(), samples, 30, stop.data_ptr(), taken.data_ptr(), side.cuda_stream), "bsr_clock_trace")
    for i in range(n_warm):
        run_step(i)
    # the region is bracketed on the host like `value` (the stream is drained on both sides)
    cur = torch.cuda.current_stream(dev)
    cur.synchronize()
    t0 = time.perf_counter()
    for i in range(n_reg):
        run_step(i)
    cur.synchronize()
    dt = time.perf_counter() - t0
    with torch.cuda.stream(flag_stream):
        stop.fill_(1)
    flag_stream.synchronize()
    side.synchronize()
    n = int(taken.item())
    tr = buf[:2 * n].cpu().view(n, 2).double()
    clock = {"clock_ghz": None}
    if n >= 64:
        # the pairs that fall inside the timed region: the last dt seconds before the stop (100 MHz ticks)
        ticks = tr[:, 1]
        t_end = float(ticks[-1])
        inside = (ticks >= t_end - dt * 1e8) & (ticks <= t_end)
        seg = tr[inside]
        if seg.shape[0] >= 32:
            step_ = max(1, seg.shape[0] // 200)                     # ~200 intervals of >= 100 us
            a, b_ = seg[:-step_:step_], seg[step_::step_]
            ghz = ((b_[:, 0] - a[:, 0]) / (b_[:, 1] - a[:, 1]) * 0.1)
            ghz = ghz[(b_[:, 1] - a[:, 1]) > 0]
            g = sorted(float(x) for x in ghz)
            clock = {"clock_ghz": round(g[len(g) // 2], 3), "clock_ghz_min": round(g[0], 3), "clock_ghz_max": round(g[-1], 3), "clock_intervals": len(g)}
    res = {"value": round(world * B * n_reg / dt, 2), "unit": "images/sec", "seconds": round(dt, 3), "steps": n_reg, "warm_steps": n_warm,
           "ms_per_step": round(dt / n_reg * 1e3, 4), "forwards_in_flight": 1,
           "clock_source": "s_memtime / s_memrealtime pairs taken every ~100 us by a one-wave kernel on a side stream during the region (bsr_clock_trace)"}
    res.update(clock)
    return res


def secondary_f32x3(weights, device, inp, uv, out, B, args, world, timed, with_parity, lanes=None):
    """The same workload on the split-precision path, reported BESIDE the f32 line (never as `value`): per-GPU images/s of this
    rank, its own roofline object, and (N = 1) its parity against the oracle under the fp32 tolerances."""
    import torch
    from blindshadowremoval_amd import Generator
    gen = Generator(device=device, dtype="f32x3").load_weights(weights)

    def region(gens_, lanes_, outs_):
        for i in range(args.warmup + args.steps):
            if i == args.warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            k = i % len(gens_)
            with (torch.cuda.stream(lanes_[k]) if lanes_[k] is not None else contextlib.nullcontext()):
                gens_[k](inp, uv, out=outs_[k])
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    dt_serial = region([gen], [None], [out])
    dt = dt_serial
    two = lanes is not None and len(lanes) > 1 and lanes[0] is not None
    if two:                                             # as the f32 line: two forwards in flight on two handles / streams
        gen_b = Generator(device=device, dtype="f32x3").load_weights(weights)
        out_b = tuple(torch.empty_like(t) for t in out)
        dt = region([gen, gen_b], list(lanes[:2]), [out, out_b])
        gen_b.close()
    rf, dom = roofline_from_events(gen, lambda: gen(inp, uv, out=out), B, "f32x3")
    attach_traffic(rf, dom, B, "f32x3")
    attach_group_traffic(rf, B, "f32x3")
    attach_mfma(rf, dom, B, "f32x3")
    res = {"dtype": "f32x3", "value": round(B * args.steps / dt_serial, 2), "unit": "images/sec (this GPU, no collective)", "ms_per_step": round(dt_serial / args.steps * 1e3, 4),
           "steps": args.steps, "forwards_in_flight": 1, "value_mode": "one forward at a time",
           "two_in_flight": ({"value": round(B * args.steps / dt, 2), "ms_per_step": round(dt / args.steps * 1e3, 4)} if two else None), "roofline": rf,
           "note": "EVERY matrix kernel of the forward runs on v_mfma_f32_32x32x16_f16 with both operands split into hi + lo fp16 planes (three "
                   "instructions per K group: hi.hi + hi.lo + lo.hi, fp32 accumulate) — stem, 3x3 / stride-2 / transposed 3x3, the 1x1 GEMMs, "
                   "attention, heads, colour tail; activations stay fp32 in HBM (theta|phi|g travel pre-split), glue kernels are the fp32 ones"}
    if with_parity:
        from oracle.gsc_oracle import GeneratorOracle
        torch.manual_seed(0)
        i8, u8 = torch.rand(8, 256, 256, 3), torch.rand(8, 256, 256, 3)
        hip = [t.cpu() for t in gen(i8.to(inp.device), u8.to(inp.device))]
        bmask = gen.probe("bmask").cpu()
        oracle, pr = GeneratorOracle(weights), {}
        oracle(i8, u8, probes=pr)
        ref = oracle(i8, u8, bmask_override=bmask)
        res["parity"] = {"max_abs_err": max(float((a - r).abs().max()) for a, r in zip(hip, ref)), "bmask_flips": int((bmask != pr["bmask"]).sum()),
                         "sample": "8 synthetic images, all four outputs vs the CPU oracle (tolerance 1e-3)"}
    gen.close()
    return res


def secondary_batch16(gen, dev, rate_b32):
    """BASELINE configs[2]'s batch (B = 16) forward only, one at a time, on the same handle — beside the B = 32 line, never `value`.
    Below B = 32 the 1/8-resolution trunk picks smaller workgroup shapes (csrc/attention.h QW, 2x32 conv tiles) to keep the chip
    covered; the full sweep B = 1 ... 32 is tools/batch_sweep.py -> profiles/r4_batch_sweep.json."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(4321)
    inp = torch.rand(16, 256, 256, 3, generator=g).to(dev)
    uv = torch.rand(16, 256, 256, 3, generator=g).to(dev)
    out = tuple(torch.empty((16, 256, 256, c), device=dev) for c in (1, 3, 3, 1))
    for _ in range(3):
        gen(inp, uv, out=out)
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            gen(inp, uv, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        best = dt if best is None else min(best, dt)
    res = {"workload": "BASELINE configs[2] batch: 16 synthetic 256x256x3 images, forward only, one at a time", "value": round(16 / best, 2),
           "unit": "images/sec", "ms_per_forward": round(best * 1e3, 4), "forwards_timed": 60}
    if rate_b32:
        res["rate_vs_batch32_single_stream"] = round(16 / best / rate_b32, 4)
    return res


# ----------------------------------------------------------------------------------------------- one rank
def run_rank(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    on_gpu = not args.stub
    distributed = world > 1 or os.environ.get("BSR_BENCH_FORCE_DIST") == "1"      # the latter exercises the collective path on one rank
    if args.device is not None:
        local_rank = args.device                        # every rank on the same GPU (gloo control plane, peer-copy gather)
    if on_gpu:
        if not torch.cuda.is_available() or local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d has no GPU %d (visible: %d)" % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    real_stdout = None
    if distributed:
        # RCCL prints a version banner on stdout (through C stdio, flushed at exit): keep stdout for the ONE JSON line by pointing
        # fd 1 at stderr for the rest of the process and writing the line to the saved descriptor
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")

    def init_group():
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    red_dev = dev if args.backend == "nccl" else torch.device("cpu")      # where the scalar reductions of the timing live (gloo: host tensors)

    B = args.batch
    tsm = args.workload == "tsm512"
    HW = 512 if tsm else 256
    weights = None
    if args.stub:
        gen = _StubGenerator()
    else:
        from blindshadowremoval_amd import Generator, GeneratorTSM, init_weights
        weights = init_weights(1, variant="tsm" if tsm else "gsc")
        gen = (GeneratorTSM if tsm else Generator)(device=local_rank, dtype=args.dtype).load_weights(weights)
    # Steps are independent forwards: with --streams 2 step i runs on handle / stream i & 1 (its own workspace, the same weights), two
    # in flight.  Each launch of the 1/8-resolution trunk is ONE round of workgroups; alone on the chip its tail and the next launch's
    # ramp idle most CUs, a second forward's kernels fill them (scratch/two_stream.py: +4 % at f32, +8 % at f32x3).
    nlanes = args.streams if on_gpu else 1
    gens = [gen]
    lanes, lanes_verified = [None], True
    if nlanes > 1:
        gens += [(GeneratorTSM if tsm else Generator)(device=local_rank, dtype=args.dtype).load_weights(weights) for _ in range(nlanes - 1)]
        from blindshadowremoval_amd.lanes import concurrent_streams
        lanes, lanes_verified = concurrent_streams(local_rank, nlanes)       # streams SEEN to overlap (distinct hardware queues), not merely distinct objects
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    inp = torch.rand(B, HW, HW, 3, generator=g).to(dev)       # synthetic, resident in HBM before timing
    uv = torch.rand(B, HW, HW, 3, generator=g).to(dev)
    reg = ((torch.rand(B, HW, HW, 6, generator=g) - 0.5) * 0.2).to(dev) if tsm else None
    outs = [tuple(torch.empty((B, HW, HW, c), device=dev) for c in (1, 3, 3, 1)) for _ in range(2)]
    packed = [torch.empty((B, HW, HW, 4), device=dev) for _ in range(2)]          # con_rgb | dif: what callers consume
    gathered = [torch.empty((world * B, HW, HW, 4), device=dev) for _ in range(2)] if distributed else None
    pending = [None, None]
    last = [None, None]                     # the output tuple of the last forward in each slot
    pre_rf = None
    if distributed:
        # The per-launch events of the roofline object are taken on rank 0 BEFORE the process group exists: with RCCL initialised the
        # same event-bracketed forwards run ~10 % longer (profiles/r3_bench_dist1.json of the earlier rounds: 5.64 vs 5.14 ms of kernel
        # time, frac 0.716 vs 0.793) although the timed region itself is unchanged — an artefact of event recording, not of the kernels.
        if rank == 0 and on_gpu and not tsm:
            for _ in range(12):                     # the chip is cold here (in the single-process run the events follow the timed regions)
                gen(inp, uv, out=outs[0])
            torch.cuda.synchronize()
            pre_rf = roofline_from_events(gen, lambda: gen(inp, uv, out=outs[0]), B, args.dtype)
        init_group()
    peer = None
    if distributed and args.gather == "peer" and not args.no_gather:
        from blindshadowremoval_amd.peer_gather import PeerGather
        ctl = dist.new_group(backend="gloo") if args.backend == "nccl" else None
        peer = PeerGather((B, HW, HW, 4), torch.float32, dev, control_group=ctl)
        gathered = [peer.gathered(0), peer.gathered(1)]

    def on_lane(k):
        return torch.cuda.stream(lanes[k]) if lanes[k] is not None else contextlib.nullcontext()

    def forward(slot, pack=False, lane=None):
        k = (slot % nlanes) if lane is None else lane
        g_ = gens[k]
        with on_lane(k):
            if tsm:
                return g_(inp, uv, reg, 2, True)        # frame = 2 (image + mirror pairs, train_with_TSM.py:676)
            if pack:                                    # con_rgb | dif written by the tail kernel straight into the all-gather payload (bsr_forward_packed)
                return g_(inp, uv, out=outs[slot], packed_out=packed[slot])
            return g_(inp, uv, out=outs[slot])

    def step(i, gather=True, lane=None):
        slot = i & 1
        k = (slot % nlanes) if lane is None else lane
        with on_lane(k):                            # the gather of step i is ordered after ITS forward's stream; step i+1 runs beside both
            if pending[slot] is not None:           # buffers of step i-2 are free once its gather completed
                pending[slot].wait()
                pending[slot] = None
            gathering = distributed and gather and not args.no_gather
            o = forward(slot, pack=gathering, lane=k)
            last[slot] = o
            if gathering:
                if tsm:
                    torch.cat((o[1], o[3]), dim=3, out=packed[slot])
                if peer is not None:
                    # finish(step i-1) BEFORE push(step i): its barrier also orders everyone's use of this slot's previous contents (step
                    # i-2) before anyone overwrites them; the copies of step i then run beside the forward of step i+1
                    if peer_open[0]:
                        peer.finish()
                    peer.push(slot, packed[slot])
                    peer_open[0] = True
                else:
                    pending[slot] = dist.all_gather_into_tensor(gathered[slot], packed[slot], async_op=True)

    peer_open = [False]

    def drain():
        for s in (0, 1):
            if pending[s] is not None:
                pending[s].wait()
                pending[s] = None
        if peer is not None and peer_open[0]:
            peer.finish()
            peer_open[0] = False

    def timed(nsteps, fn):
        """barrier + synchronize on both sides, MAX over ranks (the contract's timed region)"""
        if distributed:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for i in range(nsteps):
            fn(i)
        drain()
        sync()
        if distributed:
            dist.barrier()
        sync()
        own = time.perf_counter() - t0
        mx = own
        if distributed:
            t = torch.tensor([own], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            mx = float(t.item())
        return mx, own

    # `value` is the configuration BASELINE names, literally: ONE batch of B images per GPU resident at a time — every step is one whole
    # forward on ONE handle and ONE stream, the next step starts when the stream reaches it (round 5; rounds 3-4 took `value` with two
    # forwards in flight, which keeps 2 x B images resident: that figure is now the side measurement `two_in_flight`)
    serial = (lambda i: step(i, lane=0))
    for i in range(args.warmup):
        serial(i)
    drain()
    elapsed, own = timed(args.steps, serial)                # THE timed region: exactly K steps
    reps = [elapsed] + [timed(args.steps, serial)[0] for _ in range(max(0, args.repeats - 1))]
    extra = {}
    sustained = None
    if on_gpu and not args.no_sustained:
        sustained = sustained_region(serial if not distributed else (lambda i: step(i, gather=False, lane=0)), local_rank, B, 1, elapsed / args.steps * 1e3)
        drain()
        if distributed:                                 # every rank ran its own region: report the slowest rank's rate times the world
            t = torch.tensor([sustained["value"]], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            sustained["value"] = round(float(t.item()) * world, 2)
            sustained["note_dist"] = "no collective inside this region; value = world x the slowest rank's rate"
    if nlanes > 1:                                      # side measurement: consecutive steps alternate between two handles on two HIP streams
        for i in range(max(2, args.warmup)):
            step(i)
        drain()
        t_two = min(timed(args.steps, step)[0] for _ in range(2))
        extra["two_in_flight"] = {"value": round(world * B * args.steps / t_two, 2), "ms_per_step": round(t_two / args.steps * 1e3, 4),
                                  "streams_seen_to_overlap": lanes_verified,
                                  "note": "NOT `value`: the same K steps alternating between two handles on two HIP streams (two forwards in flight, 2 x B images "
                                          "resident; ms_per_step is an inverse throughput there); every output is bit-identical to the forward running alone"}
    if distributed and not args.no_gather:
        t_nog, _ = timed(args.steps, lambda i: step(i, gather=False))

        def gather_only(i):
            slot = i & 1
            with on_lane(slot % nlanes):
                if peer is not None:
                    if peer_open[0]:
                        peer.finish()
                    peer.push(slot, packed[slot])
                    peer_open[0] = True
                    return
                if pending[slot] is not None:
                    pending[slot].wait()
                pending[slot] = dist.all_gather_into_tensor(gathered[slot], packed[slot], async_op=True)
        t_g, _ = timed(args.steps, gather_only)
        owns = [None] * world
        dist.all_gather_object(owns, own)
        # the collective's RESULT, not only its cost: one more step, then every rank checks that its own shard of the gathered buffer is
        # bit-identical to what it packed and that every OTHER rank's shard has the checksum that rank computed locally
        step(0)
        drain()
        sync()
        mine = gathered[0][rank * B:(rank + 1) * B]
        ok = bool(torch.equal(mine, packed[0])) and bool(torch.equal(packed[0][..., :3], last[0][1])) and bool(torch.equal(packed[0][..., 3:], last[0][3]))
        sums = [None] * world
        dist.all_gather_object(sums, float(packed[0].double().sum().item()))
        for r_, s_ in enumerate(sums):
            ok = ok and float(gathered[0][r_ * B:(r_ + 1) * B].double().sum().item()) == s_
        oks = [None] * world
        dist.all_gather_object(oks, ok)
        extra.update({"allgather": {"bytes_per_rank": packed[0].numel() * 4, "backend": args.backend, "verified": all(oks),
                               "form": ("peer: full-mesh direct peer-to-peer copies on a copy stream + gloo control barrier (peer_gather.py)" if peer is not None
                                        else "rccl: all_gather_into_tensor, asynchronous, double-buffered"),
                               "ms_alone": round(t_g / args.steps * 1e3, 4),
                               "ms_per_step_without_gather": round(t_nog / args.steps * 1e3, 4),
                               "ms_exposed_per_step": round((elapsed - t_nog) / args.steps * 1e3, 4)},
                 "per_rank_images_per_sec": [round(B * args.steps / o, 2) for o in owns]})

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        rs = sorted(r / args.steps * 1e3 for r in reps)
        cfg = {"workload": None, "images_per_gpu_per_step": B, "global_batch": world * B, "height": HW, "width": HW,
               "parallelism": "dp%d" % world, "forwards_in_flight": 1,
               "collective": ("all_gather(con_rgb|dif) per step, async, double-buffered" if distributed and not args.no_gather else "none")}
        if tsm:
            cfg["workload"] = ("BASELINE configs[4] per-rank shape: TSM generator (model_with_TSM.py), %d frames of 512x512 per GPU per step, "
                               "frame=2 groups, seeded random-init weights in the ckpt-110 variable layout" % B)
        elif args.dtype == "f32":
            which = ("BASELINE configs[1]: batch=32" if B == 32 else "BASELINE configs[2]'s batch on synthetic images (forward only): batch=16" if B == 16
                     else "BASELINE configs[1] shape at another batch: batch=%d" % B)
            cfg["workload"] = ("%s synthetic 256x256x3 per GPU, full GSC generator fp32 (seeded random-init weights in the ckpt-94 variable layout)" % which)
        elif args.dtype == "f32x3":
            cfg["workload"] = ("BASELINE configs[1] shape (batch=%d synthetic 256x256x3 per GPU, full GSC generator) with every matrix kernel on the fp16 matrix "
                               "cores in split precision (hi.hi + hi.lo + lo.hi, fp32 accumulate, fp32 activations): fp32-class accuracy, NOT the headline dtype" % B)
        else:
            cfg["workload"] = ("BASELINE configs[3] per-rank shape: batch=%d synthetic 256x256x3 per GPU, fp16 MFMA conv path (single-fp16 operands and fp16 "
                               "activations on the 3x3-conv layers, split precision elsewhere, fp32 accumulate); NOT the headline configuration" % B)
        two = extra.pop("two_in_flight", None)
        cfg.update(extra)
        if args.device is not None:
            cfg["all_ranks_on_device"] = args.device
            cfg["parallelism"] = "dp%d on ONE GPU (--device %d): the N > 1 step logic with the real generator; the ranks share a chip, `value` is not a scaling figure" % (world, args.device)
        result = {
            "metric": "images/sec at 256x256 batch inference (GSC generator forward)" if not tsm else "images/sec at 512x512 (TSM generator forward)",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic", "config": cfg,
            "repeats": {"n": len(rs), "ms_per_step_min": round(rs[0], 4), "ms_per_step_median": round(rs[len(rs) // 2], 4),
                        "ms_per_step_all": [round(r, 4) for r in rs], "note": "`value` is the FIRST timed region of exactly K steps; the others repeat it"},
        }
        result["value_mode"] = "one forward at a time"
        if sustained is not None:
            sustained["vs_value"] = round(sustained["value"] / result["value"], 4)
            sustained["note"] = ("the same step repeated for seconds after seconds of warm-up: `value` is %d steps (%.0f ms) on a chip that may still be "
                                 "settling its clock; a ratio under 0.98 means the short region flattered the rate" % (args.steps, elapsed * 1e3))
            result["sustained"] = sustained
        result["single_stream_value"] = result["value"]            # the key rounds 3-4 carried the serial figure under: now `value` itself
        if two is not None:
            result["two_in_flight_value"] = two["value"]
            result["two_in_flight"] = two
        if not args.stub:
            from blindshadowremoval_amd import _lib
            result["library_source_sha16"] = _lib.source_sha()      # the hash compiled INTO libbsr_hip.so (== the tree's, or the load had refused it)
        if args.stub:
            result["stub"] = True
            result["roofline"] = None
            result["cpu_baseline"] = None
        else:
            if tsm:
                # configs[4]'s per-rank shape priced with ITS work: 512x512 frames, the TSM channel plan, attention over 4096 tokens
                set_workload(True, HW)
                rf, dom_name = roofline_from_events(gen, lambda: forward(0, lane=0), B, args.dtype)
                rf["work_per_frame"] = {"gflop": round(GFLOP_PER_IMAGE, 3), "gflop_attention": round(2e-3 * sum(LAYER_MMAC["res%d.attention" % i] for i in range(6)), 3),
                                        "note": "TSM generator at %dx%d: model_with_TSM.py channel plan (291 / 877-wide trunk), attention over %d tokens "
                                                "(quadratic: 2 x tokens^2 x 128 MACs per block); ShareLayer warp / reduce / unwarp kernels are glue (no matrix work)" % (HW, HW, (HW // 8) ** 2)}
                rf["traffic_note"] = "no counter pass for this workload: see profiles/r5_kernel_stats_tsm512.csv for the kernel durations"
                result["roofline"] = rf
            else:
                rf, dom_name = pre_rf if pre_rf is not None else roofline_from_events(gen, lambda: forward(0, lane=0), B, args.dtype)
                attach_traffic(rf, dom_name, B, args.dtype)
                attach_group_traffic(rf, B, args.dtype)
                attach_mfma(rf, dom_name, B, args.dtype)
                result["roofline"] = rf
                if two is not None and not distributed:
                    result["roofline_in_flight"] = roofline_in_flight(gens, lanes, lambda k: forward(k, lane=k), B, args.dtype, dom_name, two["ms_per_step"])
                if args.dtype == "f32" and world == 1 and not args.no_secondary:
                    result["batch16"] = secondary_batch16(gen, dev, value)
            if not args.no_cpu_baseline and world == 1 and not tsm:
                result["cpu_baseline"] = cpu_baseline(weights, gen=gen, device=dev)
            else:
                result["cpu_baseline"] = None
            if args.dtype == "f32" and not tsm and not args.no_secondary:
                result["f32x3"] = secondary_f32x3(weights, local_rank, inp, uv, outs[0], B, args, world, timed if not distributed else None,
                                                  with_parity=(world == 1 and not args.no_cpu_baseline), lanes=lanes)
            if args.loop and world == 1:
                from blindshadowremoval_amd.loop_bench import loop_bench
                result["loop"] = loop_bench(args.loop, os.path.join(ROOT, "tests", "golden"), gen)
        line = json.dumps(result) + "\n"
        if real_stdout is not None:
            sys.stdout.flush()
            os.write(real_stdout, line.encode())
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
    if peer is not None:
        peer.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if real_stdout is not None:
        # fd 1 is NOT pointed back at the real stdout: RCCL's banner sits in the C stdio buffer until the process exits and would be
        # flushed into it after the JSON line
        os.close(real_stdout)
    return result


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 32; 8 for --workload tsm512)")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of K steps each; `value` comes from the first")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=("f32", "f32x3", "f16"), default="f32",
                    help="f32 = the measured path (BASELINE configs[1], fp32 matrix cores); f32x3 = split-precision fp32 on the fp16 matrix cores "
                         "(3x3-conv path; fp32-class accuracy); f16 = fp16 operands on the 3x3-conv path (configs[3])")
    ap.add_argument("--no-secondary", action="store_true", help="skip the f32x3 side measurement the default f32 run appends")
    ap.add_argument("--no-sustained", action="store_true", help="skip the seconds-long `sustained` region (and its clock trace)")
    ap.add_argument("--workload", choices=("gsc256", "tsm512"), default="gsc256",
                    help="gsc256 = BASELINE configs[1]/[3]; tsm512 = the per-rank shape of configs[4] (TSM generator, 512x512 frames)")
    ap.add_argument("--streams", type=int, choices=(1, 2), default=2,
                    help="forwards in flight per GPU: 2 = consecutive steps alternate between two handles on two HIP streams, so one step's kernels "
                         "fill the tails and ramps of the other's one-round launches — reported as `two_in_flight` BESIDE `value`, which is always one forward at a time; "
                         "1 = skip that side measurement")
    ap.add_argument("--no-gather", action="store_true", help="skip the output all-gather (N>1)")
    ap.add_argument("--gather", choices=("rccl", "peer"), default="rccl",
                    help="N>1: how the outputs are re-assembled on every rank.  rccl = one asynchronous all_gather_into_tensor per step (RCCL kernels over "
                         "xGMI); peer = full-mesh direct peer-to-peer copies of the shard into every rank's buffer (copy engines, no compute units; "
                         "blindshadowremoval_amd/peer_gather.py) — the fallback should RCCL's kernels contend with the forward's one-round launches")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="gloo: with --stub (CPU test of the rank logic), or with --device (all ranks on one GPU)")
    ap.add_argument("--device", type=int, default=None,
                    help="run ALL ranks on this one GPU (implies --backend gloo --gather peer): the N > 1 step — packed forward, double-buffered gather, "
                         "verification — with the REAL generator where only one GPU exists; the rate it prints is N forwards sharing a chip, not a scaling figure")
    ap.add_argument("--stub", action="store_true", help="CPU stand-in generator: tests the launcher / sharding / JSON contract, measures nothing")
    ap.add_argument("--loop", choices=("ffhq", "ucb"), default=None,
                    help="also time the end-to-end FSRNet.testFFHQ / FSRNet.test loop (host prep + H2D + forward + post + PNG) on the shipped fixtures")
    args = ap.parse_args(argv)
    if args.batch is None:
        args.batch = 8 if args.workload == "tsm512" else 32
    if args.device is not None:
        if args.stub:
            ap.error("--device runs the real generator: not with --stub")
        args.backend, args.gather = "gloo", "peer"      # RCCL cannot place two ranks on one device; the peer-copy gather can
    if args.backend == "gloo" and not args.stub and args.device is None:
        ap.error("--backend gloo needs --stub (the product path has no CPU fallback) or --device (all ranks on one GPU)")
    if args.stub and args.backend != "gloo":
        ap.error("--stub runs on CPU: pass --backend gloo")
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    return args


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launcher: nothing in this process has touched (or will touch) the GPU
        sys.exit(launch_ranks(args.gpus, argv, check_devices=not args.stub, one_device=args.device))
    return run_rank(args)


if __name__ == "__main__":
    main()
