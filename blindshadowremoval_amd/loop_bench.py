"""End-to-end rate of the reference's test loops on this box (bench.py --loop ffhq|ucb): host input preparation (Dataset, with and
without the worker pool) -> host-to-device -> generator forward -> the reference's post-processing -> PNG strips, i.e.
`FSRNet.testFFHQ` (train_test_GSC.py:840-890) / `FSRNet.test` (:360-748) as a user runs them — NOT the `value` of bench.py, which
times the forward alone on resident inputs.  The fixtures bench.py points it at (tests/golden) are the reference's own sample data
(sample_imgs/02165; the first 20 UCB items with their seven masks), repeated to a list of ~100 items (BASELINE configs[2]: the
UCB test set has 100 items)."""
from __future__ import annotations

import contextlib
import io
import os
import shutil
import tempfile
import time

from .dataset import Dataset
from .fsrnet import Config, FSRNet
from .weights import init_weights



def loop_bench(kind: str, data_root: str, gen=None, items: int = 100, batch: int = 16, workers: int = -1, dtype: str = "f32") -> dict:
    """``data_root``: a folder laid out like the reference's working directory — ``sample_imgs/*``, ``UCB/train/input/*`` and
    ``UCB_masks/UCB_input_images_*`` (bench.py passes the repo's tests/golden fixtures)."""
    GOLDEN = data_root
    ucb = kind == "ucb"
    cfg = Config(0)
    out_dir = tempfile.mkdtemp(prefix="bsr_loop_")
    cfg.CHECKPOINT_DIR = out_dir
    cfg.DATA_DIR_TEST = [os.path.join(GOLDEN, "UCB", "train", "input", "*") if ucb else os.path.join(GOLDEN, "sample_imgs", "*")]
    cfg.UCB_MASK_ROOT = os.path.join(GOLDEN, "UCB_masks")
    fsr = FSRNet(cfg, weights=init_weights(1), dtype=dtype)
    from .dataset import usable_cpus as _ucpu
    from .dist import rank_world
    rank, world = rank_world()
    res = {"usable_cpus": _ucpu(), "rank": rank, "world": world, "loop": "FSRNet.test (UCB, batch %d, test_step's post-processing + SSIM/PSNR: on the host in the first three modes, on the device in device_post)" % batch if ucb else "FSRNet.testFFHQ (batch %d; device_png_batch32: %d)" % (batch, 2 * batch),
           "dtype": dtype}
    try:
        from .dataset import cpu_share
        ncpu = cpu_share()                                               # affinity + cgroup quota (not the CPUs the box merely shows), divided by the ranks of this node
        modes = [("serial_loader", dict(workers=0), {}), ("pooled_loader", dict(workers=workers), {}),
                 # round 3: rows prepared ON THE DEVICE (prep.py: the workers only decode PNGs and triangulate), PNG strips assembled
                 # on the device, UCB post-processing in worker processes one batch behind the GPU
                 # worker counts from sweeps on the GPU box (16-CPU quota): loader ~1 per usable CPU, PNG 3/4 of that; UCB: loader 1/2,
                 # post 5/4 (the post-processing alone peaks at ONE process per usable CPU — scratch/post_scaling.py), three batches in flight
                 # round 4 (pipelined loop, shared pinned ring): FFHQ loader 7/8 + PNG 7/8 of the usable CPUs; UCB loader 5/8 + post-processing
                 # ONE process per usable CPU (more only adds contention: 20 workers 300 /s, 16 workers 356 /s) and no PNG pool (the
                 # post-processing workers write the strips themselves)
                 ("device_prep", dict(workers=max(2, ncpu * 5 // 8) if ucb else max(2, ncpu * 7 // 8), device_prep=fsr.gen._device, device_batch=batch),
                  dict(post_workers=max(2, ncpu), png_workers=0 if ucb else max(2, ncpu * 7 // 8), post_inflight=3, gpu_png=False))]
        if ucb:
            # round 5: test_step's post-processing + the PNG encoding run on the device (ucb_post_gpu / gpu_png); the loader's workers also
            # decode the seven masks of every item
            modes[-1][2]["post_device"] = False
            # worker counts of the device modes: 3/4 (UCB: two images + seven masks per item) and 5/8 (FFHQ) of the usable CPUs — with the C scanline reconstruction and the shared-memory ring the
            # loop's own thread and its four file-writer threads need the rest (16 workers on 16 CPUs: -25 %; scratch/loop_workers_sweep.py)
            modes.append(("device_post", dict(workers=max(2, ncpu * 3 // 4), device_prep=fsr.gen._device, device_batch=batch),
                          dict(post_workers=0, png_workers=0, post_inflight=3, gpu_png=True, post_device=True)))
        if not ucb:
            # round 5: the PNG files themselves are built on the device (gpu_png.py): no encoder pool, every usable CPU decodes / triangulates
            modes.append(("device_png", dict(workers=max(2, ncpu * 5 // 8), device_prep=fsr.gen._device, device_batch=batch),
                          dict(post_workers=0, png_workers=0, post_inflight=3, gpu_png=True)))
        if not ucb:
            # the same loop with 32 items per forward (testFFHQ's batch is the caller's choice; 16 is configs[2]'s UCB batch): the loop's
            # own thread pays its per-batch Python once per 32 items and the forward runs at its B = 32 rate
            modes.append(("device_png_batch32", dict(workers=max(2, ncpu * 5 // 8), device_prep=fsr.gen._device, device_batch=2 * batch),
                          dict(post_workers=0, png_workers=0, post_inflight=3, gpu_png=True, batch=2 * batch)))
        for label, ds_kw, fsr_kw in modes:
            mb = int(fsr_kw.get("batch", batch))                         # items per forward in this mode
            ds = Dataset(cfg, "test", ucb=ucb, **ds_kw)
            fsr.post_workers = fsr_kw.get("post_workers", 0)
            fsr.post_inflight = fsr_kw.get("post_inflight", fsr.post_inflight)
            fsr.return_figs = not fsr_kw                                 # the device_prep mode measures the loop as a user who wants the PNGs + metrics runs it
            fsr.log.png_workers = fsr_kw.get("png_workers", 0)
            fsr.log.gpu_png = bool(fsr_kw.get("gpu_png", False))        # the earlier modes keep the host encoders they were measured with
            fsr.post_device = bool(fsr_kw.get("post_device", False))
            base = list(ds.name_list)
            fast = bool(fsr_kw.get("gpu_png"))                          # the round-5 device modes: ~1 s of loop needs 2 000 / 4 000 items
            n_items = items * (((20 if fast else 10) if ucb else (40 if fast else 20)) if fsr_kw else 1)     # the fast modes need a longer list for a steady-state rate
            reps = (n_items + len(base) - 1) // len(base)
            ds.name_list = (base * reps)[:n_items]
            # item i of the repeated list is evaluated against mask i of the equally repeated mask list (FSRNet.test indexes strictly)
            masks = (fsr._ucb_masks()[:len(base)] * reps)[:n_items] if ucb else None
            if fsr_kw:                                                   # worker start-up (python + torch imports) is not part of the loop's rate
                ds.warm()
                fsr.log.warm()
                fsr.warm_pools(batch=mb)                                 # pinned staging buffers; worker pools / device kernels of the post-processing; the workspace of this batch
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                out = fsr.test(ds, batch=mb, mask_files=masks) if ucb else fsr.testFFHQ(ds, batch=mb)
            dt = time.perf_counter() - t0
            tm = dict(fsr.timings)
            res[label] = {"workers": ds.workers, "items": len(out), "images_per_sec": round(len(out) / dt, 2), "seconds": round(dt, 3),
                          "split_s": {k: round(v, 3) for k, v in tm.items() if k.endswith("_s")}, "forwards": tm.get("forwards")}
            if fsr_kw:
                t_first = tm.get("first_batch_done_s", 0.0)
                res[label].update(post_workers=fsr.post_workers if ucb else 0, png_workers=fsr.log.png_workers, gpu_png=fsr.log.gpu_png, post_device=fsr.post_device,
                                  batch=mb, steady_images_per_sec=round((len(out) - mb) / max(dt - t_first, 1e-9), 2),
                                  note="worker processes started and warmed before the clock; steady_images_per_sec = items after the first batch / time after it")
            ds.close()
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
        fsr.close()
    return res
