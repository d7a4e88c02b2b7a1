#!/usr/bin/env python3
"""Per-stage host cost of the reference's test loops on THIS box: every host stage of FSRNet.testFFHQ / FSRNet.test run ALONE — through
the loops' own _SelectPool (one worker process per usable CPU unless --workers says otherwise) with the loops' own job formats as
aggregate items/s, and once in this process as uncontended CPU milliseconds per item; plus what the stage costs the DRIVING thread
per item (pickling, pipe I/O, select loop), which all stages of a real loop share.  The table bench.py --loop figures are read
against (profiles/r5_loop_stage_table.json): a loop cannot be faster than its slowest stage alone.  No GPU is used.

    python tools/loop_stage_table.py [--out gpurun_out/r5_loop_stage_table.json] [--items 400]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_stage(pool, jobs):
    """submit every job, wait for all; -> (seconds, CPU seconds of THIS process: pickling, pipe I/O, select loop)"""
    t0, c0 = time.perf_counter(), time.process_time()
    tickets = [pool.submit(j) for j in jobs]
    for t in tickets:
        pool.result(t)
    return time.perf_counter() - t0, time.process_time() - c0


def alone_ms(fn, jobs, n=6):
    """CPU milliseconds per item of the stage's job run in THIS process, nothing else running (the uncontended cost)"""
    fn(jobs[0])
    c0 = time.process_time()
    for j in jobs[:n]:
        fn(j)
    return round((time.process_time() - c0) / min(n, len(jobs)) * 1e3, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--items", type=int, default=400)
    ap.add_argument("--workers", type=int, default=0, help="worker processes (default: one per usable CPU)")
    args = ap.parse_args()
    import numpy as np
    from blindshadowremoval_amd.dataset import Dataset, _SelectPool, usable_cpus
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    golden = os.path.join(ROOT, "tests", "golden")
    ncpu = usable_cpus()
    nw = args.workers or ncpu
    out_dir = tempfile.mkdtemp(prefix="bsr_stage_")
    cfg = Config(0)
    cfg.CHECKPOINT_DIR = out_dir
    cfg.DATA_DIR_TEST = [os.path.join(golden, "UCB", "train", "input", "*")]
    cfg.UCB_MASK_ROOT = os.path.join(golden, "UCB_masks")
    n = args.items
    res = {"usable_cpus": ncpu, "worker_processes": nw, "items_per_stage": n, "stages": {}}
    pool = _SelectPool(nw)
    shm = None
    try:
        pool.warm("rows")
        pool.warm("post")
        # ---- stage 1: the loader's host half with device preparation (PNG decode of image + ground truth, crop box, three Delaunay
        # triangulations + plane coefficients): the job Dataset(device_prep=...) hands its workers
        ds = Dataset(cfg, "test", ucb=True, device_prep=0)          # only used for its job list (no GPU call is made)
        base = list(ds._jobs())
        jobs = [base[i % len(base)] for i in range(n)]
        from blindshadowremoval_amd.dataset import build_element
        run_stage(pool, jobs[:nw])
        dt, cpu = run_stage(pool, jobs)
        res["stages"]["loader_host_half"] = {"items_per_sec": round(n / dt, 1), "driver_cpu_ms_per_item": round(cpu / n * 1e3, 3), "job_cpu_ms_alone": alone_ms(build_element, jobs),
                                             "what": "prep.host_part per item (device preparation: inflate + C scanline reconstruction + triangulate)"}
        # ---- stage 1a (round 5): the same job + the item's seven segmentation masks (decoded, bit-packed): what FSRNet.test's loader does now that
        # the post-processing itself runs on the device
        ds_m = Dataset(cfg, "test", ucb=True, device_prep=0)
        ds_m.ucb_mask_files = FSRNet(cfg)._ucb_masks()
        base_m = list(ds_m._jobs())
        jobs_m = [base_m[i % len(base_m)] for i in range(n)]
        run_stage(pool, jobs_m[:nw])
        dt, cpu = run_stage(pool, jobs_m)
        res["stages"]["loader_host_half_with_masks"] = {"items_per_sec": round(n / dt, 1), "driver_cpu_ms_per_item": round(cpu / n * 1e3, 3),
                                                        "job_cpu_ms_alone": alone_ms(build_element, jobs_m),
                                                        "what": "prep.host_part per item incl. the seven UCB masks (C scanline reconstruction, bit-packed)"}
        # ---- stage 1b (round 6): the UCB loop's job as it runs now — the worker stops after the inflate and writes the item's filtered scanlines
        # (two photographs, seven masks) + triangle tables into its slot of the ring (a plain file here); reconstruction happens on the device
        from blindshadowremoval_amd.prep import RING_CAP
        ring_path = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else out_dir, "bsr_stage_ring_%d.bin" % os.getpid())      # (the loops' ring lives in /dev/shm)
        nslots = 64
        with open(ring_path, "wb") as fh:
            fh.truncate(nslots * RING_CAP)
        jobs_r = [j + ((ring_path, i % nslots, RING_CAP, True),) for i, j in enumerate(jobs_m)]
        run_stage(pool, jobs_r[:nw])
        dt, cpu = run_stage(pool, jobs_r)
        res["stages"]["loader_host_half_with_masks_device_unfilter"] = {
            "items_per_sec": round(n / dt, 1), "driver_cpu_ms_per_item": round(cpu / n * 1e3, 3), "job_cpu_ms_alone": alone_ms(build_element, jobs_r),
            "what": "prep.host_part_ring per item with the scanline reconstruction left to the device (inflate only; filtered scanlines of image, ground truth and the seven masks into the ring slot)"}
        os.remove(ring_path)
        # ---- stage 1c (round 5): writing the PNG files the DEVICE built (gpu_png): four writer threads in the loop's process, 256x768 and 256x1792 strips
        from concurrent.futures import ThreadPoolExecutor
        from blindshadowremoval_amd.pngio import stored_layout
        for label, wpx in (("file_write_ffhq", 768), ("file_write_ucb", 1792)):
            nbytes = stored_layout(256, wpx)[4]
            blob = np.random.default_rng(1).integers(0, 256, (16, nbytes), dtype=np.uint8)

            def put(i):
                with open(os.path.join(out_dir, "f%05d.png" % i), "wb", buffering=0) as fh:
                    fh.write(memoryview(blob[i % 16]))
            with ThreadPoolExecutor(max_workers=4) as ex:
                list(ex.map(put, range(32)))
                t0, c0 = time.perf_counter(), time.process_time()
                list(ex.map(put, range(n)))
                dt, cpu = time.perf_counter() - t0, time.process_time() - c0
            res["stages"][label] = {"items_per_sec": round(n / dt, 1), "driver_cpu_ms_per_item": round(cpu / n * 1e3, 3), "bytes_per_png": int(nbytes),
                                    "what": "write() of one device-built PNG file (stored deflate) from pinned memory, 4 threads in the loop's process"}
        # ---- stage 1b: the loader's full host path (build_row: everything prepared on the CPU)
        ds_h = Dataset(cfg, "test", ucb=True)
        base_h = list(ds_h._jobs())
        nh = max(nw * 4, n // 8)
        jobs_h = [base_h[i % len(base_h)] for i in range(nh)]
        run_stage(pool, jobs_h[:nw])
        dt, cpu = run_stage(pool, jobs_h)
        res["stages"]["loader_full_host"] = {"items_per_sec": round(nh / dt, 1), "driver_cpu_ms_per_item": round(cpu / nh * 1e3, 3), "job_cpu_ms_alone": alone_ms(build_element, jobs_h, 3),
                                             "what": "dataset.build_element per item (no device preparation)"}
        # ---- stage 2: PNG strips of testFFHQ (256 x 768 RGB), read from one shared-memory batch file as the loop parks them
        rng = np.random.default_rng(0)
        import test_ucb_post as T
        key, row, box, masks, con, dif = next(iter(T.cases()))
        strip = np.rint(np.clip(np.concatenate([row[..., 0:3], con, np.repeat(dif, 3, axis=2) * 2], axis=1), 0, 1) * 255).astype(np.uint8)
        strips = np.stack([strip] * 16)
        fd, shm = tempfile.mkstemp(prefix="bsr_stage_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        os.close(fd)
        strips.tofile(shm)
        jobs = [("png", os.path.join(out_dir, "s%05d.png" % i), (shm, tuple(strips.shape), i % 16)) for i in range(n)]
        from blindshadowremoval_amd.pngio import write_png
        run_stage(pool, jobs[:nw])
        dt, cpu = run_stage(pool, jobs)
        res["stages"]["png_strip"] = {"items_per_sec": round(n / dt, 1), "driver_cpu_ms_per_item": round(cpu / n * 1e3, 3),
                                      "job_cpu_ms_alone": alone_ms(lambda j: write_png(j[1], strips[j[2][2]]), jobs),
                                      "what": "pngio.write_png of one 256x768 strip (real image content)",
                                      "bytes_per_png": os.path.getsize(os.path.join(out_dir, "s00000.png"))}
        # ---- stage 3: the reference's UCB post-processing of one item (train_test_GSC.py:424-748) incl. its seven-figure PNG strip
        mf = FSRNet(cfg)._ucb_masks()[0]
        block = np.concatenate([row[..., 0:3], row[..., 3:6], con, dif], axis=2).astype(np.float32)[None].repeat(16, axis=0)
        block.tofile(shm)
        np_ = max(nw * 4, n // 4)
        jobs = [("ucb_post", {"shm": shm, "shape": tuple(block.shape), "index": i % 16, "box": np.asarray(box, np.float32).reshape(-1)[:4], "masks": mf,
                              "png": os.path.join(out_dir, "p%05d.png" % i), "return_figs": False}) for i in range(np_)]
        from blindshadowremoval_amd.ucb_post import run_post_job
        import torch
        torch.set_num_threads(1)
        run_stage(pool, jobs[:nw])
        dt, cpu = run_stage(pool, jobs)
        res["stages"]["ucb_post"] = {"items_per_sec": round(np_ / dt, 1), "driver_cpu_ms_per_item": round(cpu / np_ * 1e3, 3), "job_cpu_ms_alone": alone_ms(lambda j: run_post_job(j[1]), jobs),
                                     "what": "ucb_post.run_post_job per item (masks, thresholds, components, composite, SSIM / PSNR, PNG strip)"}
    finally:
        pool.shutdown()
        shutil.rmtree(out_dir, ignore_errors=True)
        if shm:
            try:
                os.unlink(shm)
            except OSError:
                pass
    st = res["stages"]
    res["reading"] = {
        "testFFHQ": "round 5 (PNG files built on the device): host stages = loader_host_half + file_write_ffhq (+ the driving thread); slowest stage alone: %.0f /s.  "
                    "Rounds 3-4 (host encoder): loader_host_half + png_strip, slowest %.0f /s"
                    % (min(st["loader_host_half"]["items_per_sec"], st["file_write_ffhq"]["items_per_sec"]), min(st["loader_host_half"]["items_per_sec"], st["png_strip"]["items_per_sec"])),
        "test (UCB)": "round 6 (scanline reconstruction on the device too): host stages = loader_host_half_with_masks_device_unfilter + file_write_ucb; slowest stage "
                      "alone: %.0f /s.  Round 5 (post-processing + PNG on the device): loader_host_half_with_masks + file_write_ucb, slowest %.0f /s.  "
                      "Rounds 2-4 (host post-processing): loader_host_half + ucb_post (its PNG strip included), slowest %.0f /s"
                      % (min(st["loader_host_half_with_masks_device_unfilter"]["items_per_sec"], st["file_write_ucb"]["items_per_sec"]),
                         min(st["loader_host_half_with_masks"]["items_per_sec"], st["file_write_ucb"]["items_per_sec"]), st["ucb_post"]["items_per_sec"]),
        "cpu_sum_ms_per_item_uncontended": {"testFFHQ": round(st["loader_host_half"]["job_cpu_ms_alone"] + st["png_strip"]["job_cpu_ms_alone"], 2),
                                            "test (UCB)": round(st["loader_host_half"]["job_cpu_ms_alone"] + st["ucb_post"]["job_cpu_ms_alone"], 2)},
        "note": "items_per_sec = the stage ALONE through a pool of %d worker processes on %d usable CPUs; job_cpu_ms_alone = the same job run once in one "
                "process with nothing else on the box (what it costs uncontended); the gap between usable_cpus / job_cpu_ms_alone and items_per_sec is "
                "contention between concurrent jobs (memory bandwidth, SMT siblings, the cgroup quota's throttling) plus the driving thread" % (nw, ncpu)}
    line = json.dumps(res, indent=1)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(line + "\n")
    print(line)


if __name__ == "__main__":
    main()
