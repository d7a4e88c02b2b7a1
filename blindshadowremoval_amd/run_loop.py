"""`python -m blindshadowremoval_amd.run_loop` — the reference's `main()` (train_test_GSC.py:934-955: Config -> Dataset -> FSRNet ->
fsr.testFFHQ / fsr.test) as a command, data-parallel when launched with one process per GPU:

    python -m blindshadowremoval_amd.run_loop --loop ffhq --data 'sample_imgs/*' --checkpoint-dir log/run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 \\
        -m blindshadowremoval_amd.run_loop --loop ucb --data 'UCB/train/input/*' --mask-root . --checkpoint-dir log/run

Under a launcher (RANK / LOCAL_RANK / WORLD_SIZE in the environment) every process takes GPU LOCAL_RANK, joins the RCCL process group
and `FSRNet` shards `dataset.name_list` contiguously over the ranks (fsrnet.py); each rank writes the PNG strips of its own items
into the same `<checkpoint-dir>/test/`, rank 0 prints the progress and the final running means over ALL items, and one JSON line
with the loop's rate.  The process group is created by THIS process before it touches the GPU; nothing is re-exec'ed.
Weights: the newest `ckpt-N` under --checkpoint-dir (tf_bundle.py), or `--random-weights SEED` (the reference ships no data shards).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--loop", choices=("ffhq", "ucb"), required=True, help="ffhq = FSRNet.testFFHQ, ucb = FSRNet.test (post-processing + SSIM / PSNR)")
    ap.add_argument("--data", action="append", required=True, help="glob of item folders (Config.DATA_DIR_TEST entry); repeatable")
    ap.add_argument("--checkpoint-dir", required=True, help="Config.CHECKPOINT_DIR: weights are restored from it, PNG strips go to <dir>/test/")
    ap.add_argument("--mask-root", default=".", help="parent of the UCB_input_images_*_masks_* folders (--loop ucb)")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dtype", choices=("f32", "f32x3", "f16"), default="f32")
    ap.add_argument("--random-weights", type=int, default=None, metavar="SEED", help="seeded random-init weights in the checkpoint layout instead of a restore")
    ap.add_argument("--host-prep", action="store_true", help="prepare the rows on the host (default: on the device, prep.py)")
    ap.add_argument("--host-post", action="store_true", help="rounds 2-4 forms: UCB post-processing in worker processes and PNG encoding on the host "
                                                             "(default: both on the device — ucb_post_gpu.py, gpu_png.py)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl")
    ap.add_argument("--device", type=int, default=None, help="GPU index of this rank (default: LOCAL_RANK).  With --backend gloo several ranks may share one "
                                                              "GPU — how the world-2 loop is exercised on a one-GPU box (tests/test_fsrnet.py)")
    args = ap.parse_args(argv)

    # Before ANYTHING initialises the HIP / HSA runtime (torch.cuda.is_available() below already does): the runtime reads this at
    # start-up — the host driver only supports dmabuf IPC, and RCCL's buffer registration fails without it.  Set here, not after the
    # device probe; never by re-exec'ing the process.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29641")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.device is not None:
        if args.backend == "nccl" and world > 1:
            sys.stderr.write("run_loop: --device with several ranks needs --backend gloo (RCCL wants one GPU per rank)\n")
            return 2
        local_rank = args.device
    import torch
    if not torch.cuda.is_available():
        sys.stderr.write("run_loop: no ROCm GPU visible — the generator has no CPU path\n")
        return 2
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write("run_loop: rank %d wants GPU %d but only %d are visible\n" % (rank, local_rank, torch.cuda.device_count()))
        return 2
    torch.cuda.set_device(local_rank)
    grouped = world > 1 or os.environ.get("BSR_LOOP_FORCE_DIST") == "1"      # the latter: a ONE-rank process group (exercises the collective path on one GPU)
    if grouped:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from .dataset import Dataset, cpu_share
    from .fsrnet import Config, FSRNet
    from .weights import init_weights
    cfg = Config(local_rank)
    cfg.DATA_DIR_TEST = list(args.data)
    cfg.CHECKPOINT_DIR = args.checkpoint_dir
    cfg.UCB_MASK_ROOT = args.mask_root
    os.makedirs(os.path.join(cfg.CHECKPOINT_DIR, "test"), exist_ok=True)
    ucb = args.loop == "ucb"
    ncpu = cpu_share()                      # this rank's share of the node's usable CPUs
    # worker counts: sweeps on the 16-CPU GPU box (loop_bench.py).  With post-processing and PNG encoding on the device (the default) the
    # loader's workers — PNG decode, Delaunay meshes, the UCB masks — are the only host stage: 3/4 (UCB) / 5/8 (FFHQ) of this rank's share of the CPUs (the loop's own thread and its file writers need the rest)
    ds_kw = dict(workers=max(1, (ncpu * 5 // 8 if ucb else ncpu * 7 // 8) if args.host_post else max(1, ncpu * 3 // 4 if ucb else ncpu * 5 // 8)))
    if not args.host_prep:
        ds_kw.update(device_prep=local_rank, device_batch=args.batch)
    ds = Dataset(cfg, "test", ucb=ucb, **ds_kw)
    fsr = FSRNet(cfg, weights=init_weights(args.random_weights) if args.random_weights is not None else None, dtype=args.dtype)
    fsr.post_device = fsr.log.gpu_png = not args.host_post
    fsr.post_workers = max(2, ncpu) if ucb and args.host_post else 0
    fsr.post_inflight = 3
    fsr.return_figs = False
    fsr.log.png_workers = max(1, ncpu * 7 // 8) if args.host_post and not ucb else 0
    rc = 0
    try:
        ds.warm()
        fsr.log.warm()
        fsr.warm_pools(batch=args.batch)
        if grouped:
            import torch.distributed as dist
            dist.barrier()
        t0 = time.perf_counter()
        res = fsr.test(ds, batch=args.batch) if ucb else fsr.testFFHQ(ds, batch=args.batch)
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        dt = time.perf_counter() - t0
        if rank == 0:
            n = len(fsr.all_losses)
            means = {k: s / max(c, 1) for k, (s, c) in fsr.log.losses.items()}
            print("\n" + json.dumps({"loop": "FSRNet.test" if ucb else "FSRNet.testFFHQ", "items": n, "ranks": world, "process_group": (args.backend if grouped else None), "items_this_rank": len(res),
                                     "images_per_sec": round(n / dt, 2), "seconds": round(dt, 3), "batch": args.batch, "dtype": args.dtype,
                                     "cpus_per_rank": ncpu, "post_and_png": "host" if args.host_post else "device", "means": means}))
    finally:
        ds.close()
        fsr.close()
        if grouped:
            import torch.distributed as dist
            dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
