#!/usr/bin/env python3
"""bench.py — images/sec of the GSC generator batched forward at 256x256 on N MI355X (BASELINE.json).

A step = one forward of the hot path over one synthetic batch of 32 images per GPU (BASELINE config 2:
"Batch=32 synthetic 256x256x3, full GSC generator fp32"), inputs already resident in HBM.  For N > 1 the
batch shards one-process-per-GPU (weak scaling: 32 images per rank) and each step ends with the RCCL
all-gather that re-assembles the consumed outputs (con_rgb + dif) on every rank, overlapped with the next
step's compute.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--no-cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMAGE = 18.104        # 9 052.06 MMAC per 256x256 image
GFLOP_3X3_PER_IMAGE = 11.017    # 3x3 conv + transposed 3x3 ("3x3-conv path")
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, = fp32 vector peak
PEAK_F16_MFMA_TFLOPS = 2500.0   # same guide: BF16/F16 MFMA dense (only used for the opt-in --dtype f16 line)


def cpu_baseline(weights, seconds_budget=25.0, gen=None, device=None):
    """Oracle (torch-CPU restatement of the reference's TF graph; TF itself is not installable here) timed on
    the host cores of this box, bounded sample of the same synthetic workload.  The thread count is the best of a
    short sweep (oneDNN does not scale monotonically on a 2-socket host); `cores` reports the count actually used."""
    import torch
    from oracle.gsc_oracle import GeneratorOracle
    cores = os.cpu_count() or 1
    oracle = GeneratorOracle(weights)
    torch.manual_seed(0)
    b = 8
    inp, uv = torch.rand(b, 256, 256, 3), torch.rand(b, 256, 256, 3)
    t_all = time.perf_counter()

    def run_once():
        t0 = time.perf_counter()
        oracle(inp, uv)
        return time.perf_counter() - t0
    best_threads, best_t = 1, float("inf")
    for threads in sorted({max(1, cores // 8), max(1, cores // 4), max(1, cores // 2)}):
        torch.set_num_threads(threads)
        run_once()                               # warm-up (oneDNN primitive creation for this thread count)
        t = run_once()
        if t < best_t:
            best_threads, best_t = threads, t
        if time.perf_counter() - t_all > seconds_budget * 0.6:
            break
    torch.set_num_threads(best_threads)
    times = [best_t]
    while len(times) < 5 and (time.perf_counter() - t_all) < seconds_budget:
        times.append(run_once())
    times.sort()
    med = times[len(times) // 2]
    out = {"value": round(b / med, 3), "unit": "images/sec", "cores": best_threads, "kind": "port",
           "sample": "%d forwards of %d synthetic 256x256 images, median (oracle-CPU torch/oneDNN fp32, proxy for the TF2-CPU path; "
                     "thread count = best of a sweep on a %d-thread host)" % (len(times), b, cores)}
    if gen is not None:
        # the checker role of the oracle (BASELINE metric: "... PSNR vs TF2 ref"): the HIP outputs of the same sample against it,
        # compared given the same 32x32 threshold mask (SURVEY F7 protocol, tests/parity_util.py)
        hip = [t.cpu() for t in gen(inp.to(device), uv.to(device))]
        bmask = gen.probe("bmask").cpu()
        pr = {}
        oracle(inp, uv, probes=pr)
        flips = int((bmask != pr["bmask"]).sum())
        ref = oracle(inp, uv, bmask_override=bmask)
        err = max(float((a - r).abs().max()) for a, r in zip(hip, ref))
        mse = float(((hip[1].double().clamp(0, 1) - ref[1].double().clamp(0, 1)) ** 2).mean())
        out["parity"] = {"max_abs_err": err, "psnr_db_con_rgb": (round(-10.0 * __import__("math").log10(mse), 2) if mse > 0 else None),
                         "bmask_flips": flips, "sample": "the %d images of the CPU sample, all four outputs" % b}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=("f32", "f16"), default="f32",
                    help="f32 = the measured path (BASELINE configs[1]); f16 = opt-in fp16 MFMA on the 3x3-conv path (configs[3])")
    ap.add_argument("--no-gather", action="store_true", help="skip the output all-gather (N>1)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from blindshadowremoval_amd import Generator, init_weights

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    distributed = world > 1 or os.environ.get("BSR_BENCH_FORCE_DIST") == "1"      # the latter exercises the RCCL path on one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    B = args.batch
    weights = init_weights(1)
    gen = Generator(device=local_rank, dtype=args.dtype).load_weights(weights)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    inp = torch.rand(B, 256, 256, 3, generator=g).to(dev)       # synthetic, resident in HBM before timing
    uv = torch.rand(B, 256, 256, 3, generator=g).to(dev)
    outs = [tuple(torch.empty((B, 256, 256, c), device=dev) for c in (1, 3, 3, 1)) for _ in range(2)]
    packed = [torch.empty((B, 256, 256, 4), device=dev) for _ in range(2)]          # con_rgb | dif: what callers consume
    gathered = [torch.empty((world * B, 256, 256, 4), device=dev) for _ in range(2)] if distributed else None
    pending = [None, None]

    def step(i):
        slot = i & 1
        if pending[slot] is not None:           # buffers of step i-2 are free once its gather completed
            pending[slot].wait()
            pending[slot] = None
        o = gen(inp, uv, out=outs[slot])
        if distributed and not args.no_gather:
            torch.cat((o[1], o[3]), dim=3, out=packed[slot])
            pending[slot] = dist.all_gather_into_tensor(gathered[slot], packed[slot], async_op=True)

    def drain():
        for s in (0, 1):
            if pending[s] is not None:
                pending[s].wait()
                pending[s] = None

    for i in range(args.warmup):
        step(i)
    drain()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        # roofline of the dominant kernel (igemm_conv_kernel on the 3x3 / transposed-3x3 layers): HIP events around
        # every launch of a few extra forwards on the same stream (event overhead stays out of `value`)
        gen.set_timing(True)
        acc, n_rep = {}, 3
        for _ in range(n_rep):
            gen(inp, uv, out=outs[0])
            torch.cuda.synchronize()
            for k, (ms, n) in gen.get_timing().items():
                a = acc.setdefault(k, [0.0, 0])
                a[0] += ms / n_rep
                a[1] = n
        gen.set_timing(False)
        t33 = (acc["conv3x3"][0] + acc["convT3x3"][0] + acc["convT3x3_ni2"][0]) * 1e-3
        n33 = acc["conv3x3"][1] + acc["convT3x3"][1] + acc["convT3x3_ni2"][1]
        path_tflops = GFLOP_3X3_PER_IMAGE * B / t33 / 1e3         # the whole 3x3-conv path
        # the dominant kernel: igemm_conv_kernel<3,3,1,true,4,32,4,1,1,2,32,1> = up2, up3, clr_up3 (SURVEY Appendix C MMACs)
        dom_gflop = 2e-3 * (377.49 + 1207.96 + 905.97) * B       # algorithmic GFLOP of its 3 launches
        t_dom, n_dom = acc["convT3x3_ni2"][0] * 1e-3, acc["convT3x3_ni2"][1]
        achieved = dom_gflop / t_dom / 1e3                        # TFLOP/s
        t_all = sum(v[0] for v in acc.values()) * 1e-3
        peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_F16_MFMA_TFLOPS
        traffic = None          # HBM bytes of the same launches, from the committed PMC passes (tools/pmc_traffic.py)
        tpath = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
        if os.path.isfile(tpath) and B == 32 and args.dtype == "f32":
            with open(tpath) as ft:
                traffic = json.load(ft).get("dominant_kernel_hbm_bytes_per_launch")
        result = {
            "metric": "images/sec at 256x256 batch inference (GSC generator forward)",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: batch=32 synthetic 256x256x3 per GPU, full GSC generator fp32 "
                                    "(seeded random-init weights in the ckpt-94 variable layout)" if args.dtype == "f32" else
                                    "BASELINE configs[3]: batch=32 synthetic 256x256x3 per GPU, fp16 MFMA (fp32 accumulate/storage) on the "
                                    "3x3-conv path, fp32 elsewhere; NOT the headline configuration"),
                       "images_per_gpu_per_step": B, "global_batch": world * B, "height": 256, "width": 256,
                       "parallelism": "dp%d" % world,
                       "collective": ("all_gather(con_rgb|dif) per step, async" if distributed and not args.no_gather else "none")},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_note": "HBM bytes per launch of the same kernel (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes, profiles/r1_pmc_traffic.csv); "
                                         "algorithmic activation bytes of its launches (input read once, output written once): 5.87e8 per launch on average",
                         "kernel": "igemm_conv_kernel<3,3,1,true,4,32,4,1,1,2,32,1> (transposed 3x3: up2, up3, clr_up3) — the largest single kernel, 24 % of the forward",
                         "launches_per_forward": n_dom, "avg_launch_ms": round(t_dom * 1e3 / n_dom, 4),
                         "algorithmic_gflop_per_launch": round(dom_gflop / n_dom, 2),
                         "path_3x3": {"achieved": round(path_tflops, 2), "frac": round(path_tflops / peak, 4), "launches": n33,
                                      "ms": round(t33 * 1e3, 4), "algorithmic_gflop": round(GFLOP_3X3_PER_IMAGE * B, 2)},
                         "all_kernels_tflops": round(GFLOP_PER_IMAGE * B / t_all / 1e3, 2),
                         "class_ms": {k: round(v[0], 4) for k, v in acc.items()}},
        }
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(weights, gen=gen, device=dev)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
