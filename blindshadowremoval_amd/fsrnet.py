"""Counterpart of the reference's inference harness: ``Config``, ``FSRNet.testFFHQ`` / ``FSRNet.test`` and the
``Logging`` sink (/root/reference/train_test_GSC.py:18-79, 118-151, 360-422, 840-890; /root/reference/utils.py:196-233)
with the same call surface, driving the HIP generator instead of the Keras model.

Differences from the reference, all deliberate:
* only the generator is constructed (the reference also builds 3 discriminators, 2 optimizers and downloads VGG19
  even for testing: train_test_GSC.py:121-128);
* a dataset element is ``(img[1,10,256,256,16], box[1,4], name)`` exactly as ``dataset.py`` yields it; the
  reference runs the generator on all 10 rows and keeps row 0 (train_test_GSC.py:866-871, utils.py:231).  Rows are
  independent at inference, so by default only row 0 of each element is computed and rows of several elements are
  batched into one forward (``batch`` argument); ``all_rows=True`` reproduces the reference's 10-row forward;
* ``test`` batches the generator over several UCB items and then applies the reference's per-image post-processing
  (train_test_GSC.py:424-748: mask heuristics, connected components, composite, SSIM / PSNR) on the host — ``ucb_post.py``;
  the seven mask images per item are read from ``Config.UCB_MASK_ROOT`` (the reference reads them from the working directory).
"""
from __future__ import annotations

import contextlib
import os
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .model import Generator

SPLIT_FFHQ = (3, 3, 3, 6, 1)      # img, gt, uv, reg, face  (train_test_GSC.py:419,870)


class Config(object):
    """Class-attribute configuration, as in the reference (train_test_GSC.py:18-51)."""
    GPU_INDEX = 0
    DATA_DIR_TEST = ['sample_imgs/*']
    IMG_SIZE = 256
    MAP_SIZE = 32
    FIG_SIZE = 128
    BATCH_SIZE = 1
    CHECKPOINT_DIR = './log/test'
    UCB_MASK_ROOT = '.'              # parent of the UCB_input_images_*_masks_* folders (train_test_GSC.py:372,386-392)

    def __init__(self, gpu_idx: Optional[int] = None):
        if gpu_idx is not None:
            self.GPU_INDEX = gpu_idx

    def compile(self) -> None:
        os.makedirs(os.path.join(self.CHECKPOINT_DIR, 'test'), exist_ok=True)
        print("\nConfigurations:")
        for a in dir(self):
            if not a.startswith("__") and not callable(getattr(self, a)) and a[0].isupper():
                print("{:30} {}".format(a, getattr(self, a)))
        print("\n")


class Logging(object):
    """utils.Logging counterpart (utils.py:127-253): running-mean text + PNG strips."""

    def __init__(self, config: Config, png_threads: int = 0, png_workers: int = 0):
        self.config = config
        # png_workers > 0: strips are encoded by that many worker PROCESSES instead of threads (PIL's encoder loop takes and drops
        # the GIL dozens of times per strip; a dozen encoder threads starve the loop's own thread at a few hundred images per second)
        self.png_workers = png_workers
        self._png_pool = None
        self._png_tickets: List[int] = []
        self._shm_batches: List[Tuple[str, List[int]]] = []
        self.losses: Dict[str, List[float]] = {}
        self.quiet = False
        self.saved: List[str] = []
        # PNG encoding (zlib) releases the GIL: with png_threads > 0 (the FSRNet loops use 4) strips are encoded by background
        # threads while the loop goes on and flush() — called by the loops before they return — waits for them; the default
        # (0) writes synchronously, as the reference's cv2.imwrite does
        self._png_threads = png_threads
        self._pool = None
        self._pending: List = []
        # gpu_png (round 5): the strips become complete PNG FILES on the device (gpu_png.py / csrc/png_kernels.h: stored deflate, checksums
        # computed there) and come over as bytes; the host only write()s them, from `file_threads` threads (write releases the GIL).  The
        # pipelined loops switch it on when the generator lives on a GPU; png_threads / png_workers are the host encoders it replaces.
        self.gpu_png = False
        self.file_threads = 4
        self._encoders: Dict[int, object] = {}
        self._file_pool = None

    @staticmethod
    def accumulate(acc: Dict[str, List[float]], losses: Dict[str, float]) -> None:
        for k, v in losses.items():
            a = acc.setdefault(k, [0.0, 0])
            a[0] += float(v)
            a[1] += 1

    @staticmethod
    def format_line(acc: Dict[str, List[float]], step: int, allstep: int) -> str:
        """The reference's progress line (utils.py:152-167): running means as name:%.3g."""
        txt = ''.join('%s:%.3g, ' % (k, s / max(c, 1)) for k, (s, c) in acc.items())
        return '\r Testing ' + str(step + 1) + '/' + str(allstep) + ': ' + txt + '     '

    def display(self, losses: Dict[str, float], epoch: int, step: int, training: bool, allstep: int) -> None:
        self.accumulate(self.losses, losses)
        if not self.quiet:                       # data-parallel loops: only rank 0 talks (FSRNet._loop_body)
            print(self.format_line(self.losses, step, allstep), end='', flush=True)

    @staticmethod
    def get_imgs(fig: Sequence[torch.Tensor]) -> np.ndarray:
        """clip*255, grey -> 3 channels, take batch row 0, concatenate horizontally (utils.py:217-233).
        Returned as RGB uint8 (the reference swaps to BGR only because cv2.imwrite expects BGR)."""
        column = []
        for img in fig:          # numpy on the host: tiny torch CPU ops cost milliseconds each on a many-core box (intra-op thread pool)
            a = (img[:1].detach().float().cpu().numpy() if isinstance(img, torch.Tensor) else np.asarray(img, np.float32)[:1])
            a = np.clip(a, 0.0, 1.0) * np.float32(255)
            if a.shape[3] == 1:
                a = np.repeat(a, 3, axis=3)
            column.append(a[0, :, :, :3])
        return np.rint(np.concatenate(column, axis=1)).astype(np.uint8)

    @staticmethod
    def strips_from_batch(figs: Sequence[torch.Tensor]) -> np.ndarray:
        """get_imgs for a whole batch, on whatever device the figures live: [B,S,S,C] each -> uint8 [B,S,S*len(figs),3].  The same
        float32 arithmetic (clip, * 255, round half to even) as get_imgs, one device-to-host copy of bytes instead of one float copy
        per figure and item."""
        return Logging.strips_on_device(figs).cpu().numpy()

    @staticmethod
    def strips_on_device(figs: Sequence[torch.Tensor]) -> torch.Tensor:
        """strips_from_batch without the copy to the host: uint8 [B,S,S*len(figs),3] on the figures' device (the pipelined loops
        copy it into pinned memory asynchronously)."""
        cols = []
        for f in figs:
            a = torch.clamp(f.detach().float(), 0.0, 1.0) * 255.0
            cols.append(a.expand(-1, -1, -1, 3) if a.shape[3] == 1 else a[..., :3])
        return torch.round(torch.cat(cols, dim=2)).to(torch.uint8)

    def files_on_device(self, figs: Sequence[torch.Tensor]) -> torch.Tensor:
        """strips_on_device + the PNG encoding itself on the device: uint8 [B, file_bytes] — row j is the complete PNG file of item j's
        strip (decodes to exactly get_imgs' pixels).  A figure may be a tuple (tensor, one-channel multiplier | None, scale)."""
        dev = None
        for f in figs:
            t = f[0] if isinstance(f, tuple) else f
            dev = t.device.index if t.is_cuda else None
            break
        if dev is not None:
            # round 5: the kernel reads the figures themselves (bsr_png_encode_figs) — no uint8 strip tensor, none of its eleven elementwise /
            # concatenation launches (0.26 ms of GPU time per 16 strips); same bytes.  A figure layout it does not address -> the strip path.
            if dev not in self._encoders:
                from .gpu_png import StripEncoder
                self._encoders[dev] = StripEncoder(dev)
            files = self._encoders[dev].encode_figs(figs)
            if files is not None:
                return files
        plain = [(f[0] * f[1] * f[2] if f[1] is not None else f[0] * f[2]) if isinstance(f, tuple) else f for f in figs]
        return self.encode_strips(self.strips_on_device(plain))

    def encode_strips(self, strips: torch.Tensor) -> torch.Tensor:
        """uint8 [B,S,W,3] strips on a GPU -> their PNG files [B, file_bytes] on that GPU (gpu_png.StripEncoder, one per device)."""
        dev = strips.device.index
        if dev not in self._encoders:
            from .gpu_png import StripEncoder
            self._encoders[dev] = StripEncoder(dev)
        return self._encoders[dev].encode(strips)

    def save_files(self, files: np.ndarray, names: Sequence[str]) -> List:
        """Write the PNG files of one batch (`files`: uint8 [B, file_bytes], e.g. a view of pinned memory the device copy landed in) under
        the names save_img would give them.  Returns the write futures: the caller must not reuse the memory behind `files` before they
        are done (flush() waits for all of them too)."""
        if self._file_pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._file_pool = ThreadPoolExecutor(max_workers=max(1, self.file_threads), thread_name_prefix="bsr-file")
            os.makedirs(os.path.join(self.config.CHECKPOINT_DIR, 'test'), exist_ok=True)

        def put(path, row):
            # unbuffered (the file is one ~0.6-1.4 MB write), but a raw write may be SHORT (disk filling up, a signal): loop until every byte is
            # out, and raise — through the future, in flush() — when the device stops taking bytes.  The stored-deflate files are raw size + 0.3 %:
            # at 3-4.5 k files/s the loops emit 3-4 GB/s, so a full disk is a real way for this to fail.
            mv = memoryview(row).cast("B")
            with open(path, "wb", buffering=0) as f:
                done = 0
                while done < len(mv):
                    n = f.write(mv[done:])
                    if not n:
                        raise OSError("short write: %d of %d bytes of %s" % (done, len(mv), path))
                    done += n
        futs = []
        for j, name in enumerate(names):
            out = self._png_path(name)
            self.saved.append(out)
            futs.append(self._file_pool.submit(put, out, files[j]))
        self._pending.extend(futs)
        if len(self._pending) > 4096:            # keep the list of finished futures short (exceptions surface here or in flush())
            done = [f for f in self._pending if f.done()]
            for f in done:
                f.result()
            self._pending = [f for f in self._pending if not f.done()]
        return futs

    def _png_path(self, fname: str) -> str:
        parts = fname.replace('\\', '/').split('/')
        stem = (parts[-2] + '_' if len(parts) > 1 else '') + parts[-1].split('.')[0]
        return os.path.join(self.config.CHECKPOINT_DIR, 'test', stem + '-result.png')

    def save_strip(self, strip: np.ndarray, fname: str) -> str:
        """save_img for an already assembled uint8 strip (strips_from_batch)."""
        out = self._png_path(fname)
        self.saved.append(out)
        if self.png_workers > 0:
            if self._png_pool is None:
                from .dataset import _SelectPool
                self._png_pool = _SelectPool(self.png_workers)
            self._png_tickets.append(self._png_pool.submit(("png", out, np.ascontiguousarray(strip))))
            self._png_pool._pump(block=False)
            return out
        os.makedirs(os.path.dirname(out), exist_ok=True)
        if self._png_threads > 0:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=self._png_threads, thread_name_prefix="bsr-png")
            self._pending.append(self._pool.submit(self._write_png, strip, out))
        else:
            self._write_png(strip, out)
        return out

    def save_strips(self, strips: np.ndarray, names: Sequence[str], parked: Optional[Tuple[str, object]] = None) -> None:
        """save_strip for a whole batch [B,S,W,3]; with png_workers > 0 the batch goes to the workers through one shared-memory file.
        ``parked`` = (shared-memory file that ALREADY holds `strips`, callback to run when every strip of it is written): the
        pipelined loops copy device -> that file directly (_ShmPinnedRing), so nothing is copied here."""
        if self.png_workers <= 0:
            for strip, name in zip(strips, names):
                self.save_strip(strip, name)
            if parked is not None:
                parked[1]()
            return
        if self._png_pool is None:
            from .dataset import _SelectPool
            self._png_pool = _SelectPool(self.png_workers)
        while len(self._shm_batches) >= 4:                   # bound what sits in /dev/shm: wait for the oldest batch
            self._reap(block=True)
        if parked is None:
            shm, done = _shm_file("bsr_png_"), None
            np.ascontiguousarray(strips).tofile(shm)
        else:
            shm, done = parked
        tickets = []
        for j, name in enumerate(names):
            out = self._png_path(name)
            self.saved.append(out)
            tickets.append(self._png_pool.submit(("png", out, (shm, tuple(strips.shape), j))))
        self._shm_batches.append((shm, tickets, done))
        self._reap(block=False)

    def _reap(self, block: bool) -> None:
        """Collect finished PNG batches (oldest first) and unlink their shared-memory files."""
        while self._shm_batches:
            shm, tickets, done = self._shm_batches[0]
            if not block:
                self._png_pool._pump(block=False)
                if not all(t in self._png_pool._done for t in tickets):
                    return
            first = None
            try:
                for t in tickets:                # every ticket of the batch is collected even when one strip failed (disk full, bad path) ...
                    try:
                        self._png_pool.result(t)
                    except RuntimeError as e:
                        first = first or e
            finally:
                self._shm_batches.pop(0)
                if done is not None:             # a ring slot: goes back to its owner
                    done()
                else:
                    try:
                        os.unlink(shm)
                    except OSError:
                        pass
            if first is not None:                # ... and the first failure is reported once the batch's shared-memory file is gone
                raise first
            block = False

    def warm(self) -> None:
        """Start the PNG worker processes (png_workers > 0) before the clock runs."""
        if self.png_workers > 0 and self._png_pool is None:
            from .dataset import _SelectPool
            self._png_pool = _SelectPool(self.png_workers)
            self._png_pool.warm("rows")

    def save_img(self, fig: Sequence[torch.Tensor], fname: str) -> str:
        """Returns the PNG's path.  With png_threads > 0 the file is written ASYNCHRONOUSLY: it exists (and `saved` is accurate)
        only after flush() / close() returned."""
        strip = self.get_imgs(fig)
        out = self._png_path(fname)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        self.saved.append(out)
        if self._png_threads > 0:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=self._png_threads, thread_name_prefix="bsr-png")
            self._pending.append(self._pool.submit(self._write_png, strip, out))
        else:
            self._write_png(strip, out)
        return out

    @staticmethod
    def _write_png(strip: np.ndarray, out: str) -> None:
        from .pngio import write_png
        write_png(out, strip)                                     # cv2.imwrite's defaults (Sub filter, RLE strategy, level 1); pixels are identical

    def flush(self) -> None:
        """Wait for every queued PNG (re-raises a writer's exception)."""
        tickets, self._png_tickets = self._png_tickets, []
        first = None
        for t in tickets:                        # PNG worker processes: a failed job is reported after ALL of them were waited for
            try:
                self._png_pool.result(t)
            except RuntimeError as e:
                first = first or e
        while self._shm_batches:
            try:
                self._reap(block=True)
            except RuntimeError as e:
                first = first or e
        pending, self._pending = self._pending, []
        for f in pending:                        # wait for ALL of them even when one failed, then report the first failure
            try:
                f.result()
            except BaseException as e:           # noqa: BLE001
                first = first or e
        if first is not None:
            raise first

    def close(self) -> None:
        """flush() + shut the writer threads down."""
        try:
            self.flush()
        finally:
            if self._pool is not None:
                self._pool.shutdown(wait=True)
                self._pool = None
            if self._file_pool is not None:
                self._file_pool.shutdown(wait=True)
                self._file_pool = None
            if self._png_pool is not None:
                self._png_pool.shutdown()
                self._png_pool = None
            for shm, _, done in self._shm_batches:
                if done is None:
                    try:
                        os.unlink(shm)
                    except OSError:
                        pass
            self._shm_batches = []


def _shm_file(prefix: str) -> str:
    """A fresh file name in shared memory (/dev/shm) — batches handed to worker processes travel as ONE memory-speed file instead of
    megabytes per item through 64-KB pipes; the creator unlinks it when the batch's jobs are done."""
    import tempfile
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    fd, path = tempfile.mkstemp(prefix=prefix, dir=d)
    os.close(fd)
    return path


class _ShmPinnedRing:
    """Staging buffers that are BOTH pinned host memory (the target of an asynchronous device-to-host copy) and shared-memory files a
    worker process opens by name: a batch's PNG strips / post-processing inputs land where the workers read them, with no copy in
    the loop's thread in between (parking a batch with ndarray.tofile cost 3 ms per 16 strips and 13 ms per 16 post-processing
    items — a quarter of the loops' wall time).  Each slot is a file in /dev/shm mapped MAP_SHARED and registered with the HIP
    runtime (hipHostRegister).  `acquire(nbytes)` hands out a free slot, `release(slot)` returns it once its readers are done.
    If the runtime refuses the registration the constructor raises and the loops fall back to private pinned buffers + tofile."""

    def __init__(self, n: int, nbytes: int):
        import mmap
        self.nbytes = int(nbytes)
        self.slots = []                          # (path, mmap, uint8 tensor)
        self.free: List[int] = []
        # a mapping of a tmpfs file that the file system cannot back kills the process with SIGBUS on first touch (truncate() succeeds
        # on a full /dev/shm): refuse up front unless the slots fit with room to spare
        probe = _shm_file("bsr_ring_probe_")
        try:
            vfs = os.statvfs(os.path.dirname(probe))
        finally:
            os.unlink(probe)
        if vfs.f_bavail * vfs.f_frsize < int(1.25 * n * self.nbytes) + (64 << 20):
            raise RuntimeError("shared memory has %.0f MB free, the ring needs %.0f MB" % (vfs.f_bavail * vfs.f_frsize / 1e6, n * self.nbytes / 1e6))
        self._rt = torch.cuda.cudart()
        try:
            for i in range(n):
                path = _shm_file("bsr_ring_")
                with open(path, "r+b") as f:
                    f.truncate(self.nbytes)
                    mm = mmap.mmap(f.fileno(), self.nbytes, flags=mmap.MAP_SHARED)
                t = torch.frombuffer(mm, dtype=torch.uint8)
                self.slots.append([path, mm, t, False])
                rc = self._rt.cudaHostRegister(t.data_ptr(), self.nbytes, 1)
                if int(rc) != 0:
                    raise RuntimeError("hipHostRegister of a /dev/shm mapping failed (%s)" % rc)
                self.slots[-1][3] = True
                self.free.append(i)
        except BaseException:
            self.close()
            raise

    def acquire(self) -> Optional[int]:
        return self.free.pop(0) if self.free else None

    def release(self, slot: int) -> None:
        self.free.append(slot)

    def view(self, slot: int, dtype, shape) -> torch.Tensor:
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        return self.slots[slot][2][:n].view(dtype).reshape(shape)

    def path(self, slot: int) -> str:
        return self.slots[slot][0]

    def close(self) -> None:
        for path, mm, t, registered in self.slots:
            try:
                if registered:
                    self._rt.cudaHostUnregister(t.data_ptr())
            except Exception:
                pass
            try:
                os.unlink(path)
            except OSError:
                pass
        self.slots, self.free = [], []


def _name(x) -> str:
    if isinstance(x, bytes):
        return x.decode()
    if isinstance(x, np.ndarray):
        return _name(x.reshape(-1)[0])
    return str(x)


class FSRNet(object):
    def __init__(self, config: Config, weights: Optional[Dict[str, np.ndarray]] = None, dtype: str = "f32", group=None, gen=None):
        """``group``: a torch.distributed process group; with one (or with the default group initialised — `torchrun`, one process per
        GPU) ``test`` / ``testFFHQ`` run DATA-PARALLEL: rank r takes the contiguous shard r of ``dataset.name_list``, writes the PNG
        strips of its own items, and the per-item losses are gathered at the end (``all_losses``; rank 0 prints the running means over
        ALL items exactly like the single-process loop).  ``gen``: a ready generator object to drive instead of constructing one
        (the multi-process CPU tests pass a stand-in: the product path has no CPU generator)."""
        self.config = config
        self.group = group
        if gen is not None:
            self.gen = gen
        else:
            self.gen = Generator(device=config.GPU_INDEX if torch.cuda.is_available() else None, dtype=dtype)
            if weights is not None:
                self.gen.load_weights(weights)
        self.log = Logging(config, png_threads=4)
        # PNG strips are built as complete files ON THE DEVICE when the generator lives on one (gpu_png.py); False: the host encoders
        # (png_threads / png_workers) of rounds 2-4 — same pixels either way
        self.log.gpu_png = getattr(self.gen, "_device", None) is not None and torch.cuda.is_available()
        self.shm_ring = True                     # device-to-host copies of pool-bound batches land in pinned shared-memory slots (_ShmPinnedRing); False: private pinned buffers + a file copy
        self.gpu_inflight = 2                    # batches whose forward + device-to-host copy may be outstanding while the loop feeds the next one
        self.all_losses: List[Tuple[str, Dict[str, float]]] = []      # (name, losses) of EVERY item in list order — on every rank after a data-parallel loop
        self.timings: Dict[str, float] = {}      # wall-clock split of the last test / testFFHQ loop (see _loop)
        self._pin_pool: List[Optional[torch.Tensor]] = []      # pinned device-to-host staging buffers of the loops (kept across loops; warm_pools() pre-allocates)
        self.post_device = True                  # FSRNet.test's post-processing on the GPU when the generator lives on one (ucb_post_gpu); False: the host forms below
        self.post_threads = 8                    # threads that post-process the items of one UCB batch (post_workers == 0)
        self.post_workers = 0                    # > 0: UCB post-processing in that many worker PROCESSES, pipelined one batch behind the GPU
        self.post_inflight = 2                   # batches whose post-processing may be outstanding
        self.return_figs = True                  # False: FSRNet.test returns (name, None, losses) — the figures are only written as PNG strips
        self._post_writes_png = False            # set by test(): the post-processing workers write the PNG strips themselves
        self._post_pool, self._post_pool_n = None, 0

    # -- worker pools ----------------------------------------------------------------------
    def _get_post_pool(self):
        if self._post_pool is None or self._post_pool_n != self.post_workers:
            self.close_pools()
            from .dataset import _SelectPool
            self._post_pool, self._post_pool_n = _SelectPool(self.post_workers), self.post_workers
        return self._post_pool

    def warm_pools(self, batch: Optional[int] = None) -> None:
        """Start the post-processing workers (post_workers > 0) and let them import torch / scipy now, not inside the timed loop; with the
        device post-processing (post_device on a GPU) run its kernels and the PNG encoder once on a dummy item — the first launch of a
        kernel family loads its code object, the first use of a torch operator its module: ~1 s in all on a fresh process."""
        dev = getattr(self.gen, "_device", None)
        if batch and dev is not None and torch.cuda.is_available():
            # one forward of the loop's batch: the library sizes its workspace for the largest batch it has seen (a hipMalloc + the first
            # launches of the batch-dependent instantiations: ~0.15 s that would otherwise sit in the loop's first batch)
            s_ = self.config.IMG_SIZE
            z = torch.zeros((int(batch), s_, s_, 3), dtype=torch.float32, device="cuda:%d" % dev)
            self.gen(z, z, None, chuck=1, training=False)
            torch.cuda.synchronize(dev)
            try:
                self.gen.check_range()             # zeros in, nothing out of range: leaves the flag clear
            except Exception:
                pass
        if dev is not None and torch.cuda.is_available():      # the loops' pinned staging buffers: gpu_inflight + 1 of a batch of 16 seven-figure strips each
            while len(self._pin_pool) < int(self.gpu_inflight) + 1:
                self._pin_pool.append(None)
            for k in range(int(self.gpu_inflight) + 1):
                if self._pin_pool[k] is None:
                    self._pin_pool[k] = torch.empty(24 << 20, dtype=torch.uint8).pin_memory()
        if self.post_device and dev is not None and torch.cuda.is_available():
            from .prep import unpack_masks
            from .ucb_post_gpu import UcbPostDevice
            s = self.config.IMG_SIZE
            d = torch.device("cuda", dev)
            bits = np.zeros((7, s * s // 8), np.uint8)
            bits[:, : s * s // 16] = 255
            masks = unpack_masks([("bits", bits, s)], d)
            rows = torch.rand((1, s, s, 10), device=d)
            _, strips, _, status = UcbPostDevice(dev).run(rows, masks, torch.tensor([[0, 0, s - 16, s - 16]], dtype=torch.float32, device=d))
            self.log.encode_strips(strips)
            torch.cat([strips.reshape(-1)[:8], status.view(torch.uint8)]).cpu()
        elif self.post_workers > 0:
            self._get_post_pool().warm("post")

    def close_pools(self) -> None:
        pool, self._post_pool = self._post_pool, None
        if pool is not None:
            pool.shutdown()

    def close(self) -> None:
        self.close_pools()
        self.log.close()
        self.gen.close()

    # -- checkpoint -------------------------------------------------------------------------
    def _restore(self) -> int:
        last_epoch = self.gen.restore(self.config.CHECKPOINT_DIR) if self.gen._handle is None else -1
        if not self.log.quiet:
            print('**********************************************************')
            print('Restore from Epoch ' + (str(last_epoch) if last_epoch >= 0 else '(weights supplied)'))
            print('**********************************************************')
        if self.gen._handle is None:
            raise RuntimeError("no generator weights: checkpoint data shard missing under %s" % self.config.CHECKPOINT_DIR)
        return last_epoch

    # -- steps ------------------------------------------------------------------------------
    def _split(self, img: torch.Tensor, rows: Optional[int]) -> Tuple[torch.Tensor, ...]:
        s = self.config.IMG_SIZE
        img = torch.as_tensor(np.asarray(img) if not isinstance(img, torch.Tensor) else img, dtype=torch.float32)
        img = img.reshape(-1, s, s, img.shape[-1])                         # [10,256,256,16] (train_test_GSC.py:866)
        if rows is not None:
            img = img[:rows]
        return torch.split(img, list(SPLIT_FFHQ), dim=3)

    def test_step_FFHQ(self, img, box=None, training: bool = False, all_rows: bool = False):
        """train_test_GSC.py:863-890 for one dataset element."""
        im, gt, uv, reg, face = self._split(img, None if all_rows else 1)
        dev = "cuda:%d" % self.gen._device
        _, con_rgb, _, mask_pred = self.gen(im.contiguous().to(dev), uv.contiguous().to(dev), reg, chuck=1, training=training)
        mask_pred = mask_pred * face.to(dev)
        con_rgb = torch.clamp(con_rgb, 0, 1)
        return {}, [im.to(dev), con_rgb, mask_pred * 2]

    def test_step(self, img, box=None, *masks, training: bool = False, all_rows: bool = False):
        """Head of the UCB step (train_test_GSC.py:411-422): returns the generator outputs for the element."""
        im, gt, uv, reg, face = self._split(img, None if all_rows else 1)
        dev = "cuda:%d" % self.gen._device
        gs, con_rgb, _, mask_pred = self.gen(im.contiguous().to(dev), uv.contiguous().to(dev), reg, chuck=4, training=training)
        return {}, [im.to(dev), gs, con_rgb, mask_pred, gt.to(dev), face.to(dev)]

    # -- loops ------------------------------------------------------------------------------
    def _ucb_masks(self) -> List[Dict[str, str]]:
        """Per-item mask file paths in the reference's order: the sorted listing of the with-hair folder, the same file name in
        the other six (train_test_GSC.py:372,386-392)."""
        from .ucb_post import MASK_DIRS
        root = self.config.UCB_MASK_ROOT
        first = os.path.join(root, MASK_DIRS["face_hair"])
        if not os.path.isdir(first):
            raise FileNotFoundError("FSRNet.test needs the UCB mask folders under Config.UCB_MASK_ROOT (%s is missing)" % first)
        return [{k: os.path.join(root, d, f) for k, d in MASK_DIRS.items()} for f in sorted(os.listdir(first))]

    @staticmethod
    def _read_masks(paths: Dict[str, str]) -> Dict[str, np.ndarray]:
        from PIL import Image
        out = {}
        for k, path in paths.items():
            a = np.asarray(Image.open(path).convert("L"), np.float64) / 255.0          # cv2.imread(...)/255.0, 3 equal channels
            out[k] = np.repeat(a[:, :, None], 3, axis=2)
        return out

    def _loop(self, dataset, batch: int, ucb: bool, postprocess: bool = True, mask_files=None):
        try:
            return self._loop_body(dataset, batch, ucb, postprocess, mask_files)
        finally:
            # also on an exception mid-loop (loader failure, missing mask): no queued PNG writer is left unobserved
            self.log.flush()

    def _loop_body(self, dataset, batch: int, ucb: bool, postprocess: bool, mask_files):
        """The reference's loops (train_test_GSC.py:360-408, 840-860), batched, data-parallel and pipelined:

        * rank r of a process group works on the contiguous shard r of ``dataset.name_list`` (no data-path collective: items are
          independent; one ``all_gather_object`` of the per-item losses at the end);
        * the loop's own thread only FEEDS: it pulls elements from the loader, submits a batch's forward + strip assembly (or the
          tensors the host post-processing reads) + the device-to-host copy into pinned memory, records an event, and goes back to
          the loader; up to ``gpu_inflight`` batches are outstanding, and a batch is completed — PNG strips / post-processing jobs
          handed to their worker pools, results appended — in submission order, so item order and every output bit are those of the
          serial loop."""
        from .dist import rank_world, shard_bounds
        names = list(dataset.name_list)
        num_list = len(names)
        rank, world = rank_world(self.group)
        lo, hi = shard_bounds(num_list, world)[rank] if world > 1 else (0, num_list)
        feed_is_sharded = False
        if world > 1 and hasattr(dataset, "shard"):
            dataset.shard(lo, hi)               # the loader prepares only this rank's items (a plain .feed is consumed and skipped instead)
            feed_is_sharded = True
        self.log.quiet = rank != 0
        results = []
        # where the loop's wall time goes: waiting for the loader, submitting / waiting for the GPU (host->device, forward,
        # device->host of what the host reads), handing work to the post-processing pool, PNG encoding
        tm = {"prep_wait_s": 0.0, "forward_s": 0.0, "post_s": 0.0, "png_s": 0.0, "forwards": 0, "items": 0, "rank": rank, "world": world}
        self.timings = tm
        pending: List[Tuple[int, str, torch.Tensor, object]] = []
        # the reference indexes its mask lists by the loop counter (train_test_GSC.py:386-396: masks[count]): item `step` is
        # evaluated against mask file `step`, and a shorter mask list is an error, never a wrap-around
        if ucb and postprocess:
            mask_files = self._ucb_masks() if mask_files is None else list(mask_files)
            if len(mask_files) < num_list:
                raise ValueError("FSRNet.test: %d items but only %d mask files" % (num_list, len(mask_files)))
        else:
            mask_files = None
        self._restore()
        start = time.time()
        dev = "cuda:%d" % self.gen._device if getattr(self.gen, "_device", None) is not None else "cpu"
        on_gpu = dev != "cpu"
        depth = max(0, int(self.gpu_inflight)) if on_gpu else 0
        # pinned staging buffers, one per batch that may be outstanding (+ the one being filled).  They belong to the FSRNet object and
        # survive the loop: page-locking a buffer costs ~60 ms (measured: five of them were a quarter of a 2 000-item loop's wall time)
        while len(self._pin_pool) < depth + 1:
            self._pin_pool.append(None)
        pins = self._pin_pool
        pin_busy: List[List] = [[] for _ in range(max(depth + 1, len(self._pin_pool)))]          # file writes still reading a pinned buffer (gpu_png): waited for before its turn comes again
        gpu_png = on_gpu and self.log.gpu_png
        gpu_q: List[Tuple] = []             # submitted batches whose device-to-host copy may still be running, oldest first
        d2h = torch.cuda.Stream(device=self.gen._device) if on_gpu else None
        post_stream = torch.cuda.Stream(device=self.gen._device) if on_gpu and os.environ.get("BSR_POST_SIDE", "1") != "0" else None
        turn = [0]
        # batches bound for a worker pool (PNG strips, UCB post-processing) are copied device -> a pinned SHARED-MEMORY slot the workers
        # read in place (_ShmPinnedRing); [ring | None, already tried]
        ring_state = [None, not (on_gpu and self.shm_ring)]

        # round 5: the per-item post-processing of test_step runs ON THE DEVICE (ucb_post_gpu / csrc/ucb_kernels.h), the figure strips become
        # PNG files there too, and what comes back per item is its file + two losses; post_workers / post_threads are the host forms
        post_dev = None
        pend_masks: Dict[int, object] = {}
        if ucb and postprocess and on_gpu and self.post_device:
            from .ucb_post_gpu import UcbPostDevice
            post_dev = UcbPostDevice(self.gen._device)
            if hasattr(dataset, "ucb_mask_files") and getattr(dataset, "device_prep", None) is not None and not getattr(dataset, "_started", False):
                dataset.ucb_mask_files = mask_files      # the loader's workers decode the masks next to the images (bit-packed through the pipe)
        post_pool = self._get_post_pool() if ucb and postprocess and self.post_workers > 0 and post_dev is None else None
        inflight: List[Tuple] = []          # UCB batches whose post-processing runs in the worker pool: (pending items, futures)

        poll = getattr(dataset, "poll", lambda: None)      # lets the loader's workers hand over finished elements while this thread is busy elsewhere

        def finish(batch_items, post):
            """log + save + collect one batch, in item order (post = [(losses, figs | None)] for the UCB post-processing mode)"""
            for j, (step, name, _, box) in enumerate(batch_items):
                t1 = time.perf_counter()
                losses, f = post[j]
                self.log.display(losses, 0, step, False, num_list)
                if f is not None and not self._post_writes_png:
                    self.log.save_img([torch.from_numpy(a) for a in f], name)
                elif self._post_writes_png:
                    self.log.saved.append(self.log._png_path(name))
                tm["png_s"] += time.perf_counter() - t1
                tm["items"] += 1
                results.append((name, [torch.from_numpy(a) for a in f] if f is not None else None, losses))

        def drain(keep: int):
            while len(inflight) > keep:
                t1 = time.perf_counter()
                batch_items, futs, shm, slot_ = inflight.pop(0)
                try:
                    post = [post_pool.result(fu) for fu in futs]
                finally:
                    if slot_ is not None:
                        ring_state[0].release(slot_)
                    else:
                        try:
                            os.unlink(shm)
                        except OSError:
                            pass
                tm["post_s"] += time.perf_counter() - t1
                finish(batch_items, post)

        def ring_slot(nbytes: int, to_pool: bool, nrows: int):
            """a free slot of the shared pinned ring for a batch that goes to a worker pool, or None (no pool / ring unavailable)"""
            if not to_pool:
                return None
            if ring_state[0] is None and not ring_state[1]:
                ring_state[1] = True
                try:
                    full = nbytes // max(1, nrows) * max(nrows, batch)                  # sized for a full batch
                    ring_state[0] = _ShmPinnedRing(depth + 6, full)
                except Exception as e:            # the runtime would not register a /dev/shm mapping: private pinned buffers + tofile (as in round 3)
                    tm["shm_ring_error"] = str(e)
            ring = ring_state[0]
            if ring is None or nbytes > ring.nbytes:
                return None
            slot = ring.acquire()
            while slot is None:                   # every slot is with a worker pool: collect the oldest batch
                if inflight:
                    drain(keep=len(inflight) - 1)
                elif self.log._shm_batches:
                    self.log._reap(block=True)
                else:
                    return None
                slot = ring.acquire()
            return slot

        def to_host_async(payload: torch.Tensor, to_pool: bool = False, pin_id: bool = False):
            """-> (numpy view of the payload on the host, event | None, ring slot | None).  GPU: an asynchronous copy into this batch's
            pinned buffer; the view is valid once the event has completed and until the buffer's turn comes again (depth + 1
            submissions later) — or, in a ring slot, until the slot is released."""
            if not on_gpu:
                return payload.numpy(), None, None
            nbytes = payload.numel() * payload.element_size()
            slot = ring_slot(nbytes, to_pool, int(payload.shape[0]))

            def copy_out(view):
                """the device-to-host copy on its own stream, behind the kernels that wrote `payload`: the compute stream goes straight on
                with the next batch instead of idling for the ~10 MB of files / strips of this one"""
                ready = torch.cuda.Event()
                ready.record()
                with torch.cuda.stream(d2h):
                    d2h.wait_event(ready)
                    view.copy_(payload, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                payload.record_stream(d2h)
                return ev
            if slot is not None:
                view = ring_state[0].view(slot, payload.dtype, tuple(payload.shape))
                return view.numpy(), copy_out(view), slot
            k = turn[0]
            turn[0] = (k + 1) % (depth + 1)
            for fu in pin_busy[k]:                # the file writes of the batch that used this buffer depth + 1 submissions ago
                fu.result()
            pin_busy[k] = []
            if pins[k] is None or pins[k].numel() < nbytes:
                pins[k] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8).pin_memory()
            view = pins[k][:nbytes].view(payload.dtype).reshape(payload.shape)
            return view.numpy(), copy_out(view), (-1 - k if pin_id else None)       # pin_id: the "slot" names the private pinned buffer (negative), for pin_busy

        def submit():
            """one batch: rows -> device -> generator -> what the host needs, on its way to pinned memory; nothing here waits for the GPU"""
            if not pending:
                return
            t0 = time.perf_counter()
            rows_d = self._whole_batch(pending)                     # the loader's device tensor itself when the batch is exactly its group (no 64 MB concatenation)
            if rows_d is None:
                rows = torch.cat([self._split_row0(p[2]) for p in pending], dim=0)
                rows_d = rows.to(dev, non_blocking=True)            # ONE host-to-device copy of the packed rows (none when the loader prepared them on the device)
            im_d, gt_d, uv_d, _, face_d = torch.split(rows_d, list(SPLIT_FFHQ), dim=3)
            gs, con_rgb, _, mask_pred = self.gen(im_d, uv_d, None, chuck=4 if ucb else 1, training=False)
            items = list(pending)
            pending.clear()
            if post_dev is not None:
                # train_test_GSC.py:424-748 for the whole batch on the device; the seven figures leave as PNG files
                from .prep import pack_masks, unpack_masks
                packed = [pend_masks.pop(it[0], None) or pack_masks(mask_files[it[0]]) for it in items]      # a feed without masks: read here
                # The post-processing runs on its OWN stream behind this batch's forward (BSR_POST_SIDE=0: on the compute stream): its
                # per-item kernel is ONE workgroup per item — 16 of 256 CUs busy for 1.2 ms per batch of 16 (scratch/loop_trace.sh) — so
                # the next batch's forward shares the chip with it instead of waiting behind it
                side = post_stream is not None
                if side:
                    fwd_done = torch.cuda.Event()
                    fwd_done.record()
                    for t_ in (rows_d, con_rgb, mask_pred) + tuple(pk[1] for pk in packed if isinstance(pk[1], torch.Tensor)):
                        t_.record_stream(post_stream)      # allocated on other streams, read by this one
                with torch.cuda.stream(post_stream) if side else contextlib.nullcontext():
                    if side:
                        post_stream.wait_event(fwd_done)
                    boxes = torch.from_numpy(np.stack([np.asarray(it[3], np.float32).reshape(-1)[:4] for it in items])).to(dev, non_blocking=True)
                    losses_d, strips_d, figs_d, status_d = post_dev.run(torch.cat([im_d, gt_d, con_rgb, mask_pred], dim=3), unpack_masks(packed, dev), boxes,
                                                                        want_figs=self.return_figs)
                    files_d = self.log.encode_strips(strips_d)
                    nfile = files_d.shape[1]
                    payload = torch.cat([losses_d.view(torch.uint8).reshape(-1), status_d.view(torch.uint8).reshape(-1), files_d.reshape(-1)])      # 12 bytes per item, then the files
                    host, ev, slot = to_host_async(payload, to_pool=False, pin_id=True)
                figs_b = ("post_dev", nfile, figs_d)
            elif ucb and postprocess:
                # train_test_GSC.py:424-748 runs on the host, one independent item per call: what it reads comes over in ONE copy
                host, ev, slot = to_host_async(torch.cat([im_d, gt_d, con_rgb, mask_pred], dim=3), to_pool=post_pool is not None)      # [B,S,S,10]
                figs_b = None
            else:
                # FFHQ / raw-UCB: the figures stay on the device; the PNG strips of the whole batch are assembled there and come
                # over as bytes in one copy (Logging.strips_on_device = get_imgs per item, same arithmetic)
                lazy = gpu_png and not self.return_figs        # nobody gets the figures back: the encoder clips / multiplies while it reads them
                if ucb:
                    figs_b = [im_d, gs, con_rgb, mask_pred, gt_d, face_d]
                    shown_b = [im_d, con_rgb, (mask_pred, face_d, 2.0)] if lazy else [im_d, torch.clamp(con_rgb, 0, 1), mask_pred * face_d * 2]
                elif lazy:
                    figs_b = None
                    shown_b = [im_d, con_rgb, (mask_pred, face_d, 2.0)]                          # clip(con_rgb, 0, 1) is the encoder's own clamp
                else:
                    figs_b = [im_d, torch.clamp(con_rgb, 0, 1), mask_pred * face_d * 2]        # train_test_GSC.py:872-873,889
                    shown_b = figs_b
                if gpu_png:                       # complete PNG files built on the device; the host only writes them
                    host, ev, slot = to_host_async(self.log.files_on_device(shown_b), to_pool=False, pin_id=True)
                else:
                    host, ev, slot = to_host_async(self.log.strips_on_device(shown_b), to_pool=self.log.png_workers > 0)
            gpu_q.append((items, host, ev, figs_b, slot))
            tm["forward_s"] += time.perf_counter() - t0
            tm["forwards"] += 1
            poll()
            while len(gpu_q) > depth:
                complete(gpu_q.pop(0))

        def complete(entry):
            """the batch's device work has to be done now: wait for its event, then hand its host half on"""
            items, host, ev, figs_b, slot = entry
            ring = ring_state[0]
            t0 = time.perf_counter()
            if ev is not None:
                ev.synchronize()
            if getattr(self.gen, "dtype", "f32") != "f32":
                # 16-bit modes: an out-of-range activation is an error here, never a silent inf.  The batch's own event has completed, so
                # the host-visible flag already covers it: read it WITHOUT synchronising the stream (check_range would wait for the up to
                # gpu_inflight batches submitted after this one and drain the pipeline on every batch) and without clearing it (a later
                # batch's report must not be lost): a raised flag means this batch or one of those behind it — the loop stops either way
                try:
                    self.gen.peek_range()
                except RuntimeError as e:
                    raise type(e)("batch of items %s (or one of the <= %d batches submitted after it): %s" % ([it[1] for it in items][:4], depth, e)) from None
            tm["forward_s"] += time.perf_counter() - t0
            tm.setdefault("first_batch_done_s", time.time() - start)
            poll()
            t1 = time.perf_counter()
            if post_dev is not None:
                from .ucb_post_gpu import raise_for_status
                _, nfile, figs_d = figs_b
                nb = len(items)
                losses = host[:nb * 8].view(np.float32).reshape(nb, 2)
                status = host[nb * 8:nb * 12].view(np.int32)
                files = host[nb * 12:nb * 12 + nb * nfile].reshape(nb, nfile)
                raise_for_status(status, [it[1] for it in items])
                futs = self.log.save_files(files, [it[1] for it in items])
                if slot is not None and slot < 0:
                    pin_busy[-1 - slot] = futs
                else:
                    for fu in futs:
                        fu.result()
                figs_h = figs_d.cpu() if figs_d is not None else None
                for j, it in enumerate(items):
                    lo_ = {"ssim": float(losses[j, 0]), "psnr": float(losses[j, 1])}
                    self.log.display(lo_, 0, it[0], False, num_list)
                    tm["items"] += 1
                    results.append((it[1], [figs_h[j, k][None] for k in range(7)] if figs_h is not None else None, lo_))
                tm["post_s"] += time.perf_counter() - t1
                return
            if ucb and postprocess:
                if post_pool is not None:
                    post_pool._pump(block=False)
                    if slot is not None:
                        shm = ring.path(slot)                       # the copy from the device landed in the shared-memory slot itself
                    else:
                        shm = _shm_file("bsr_post_")
                        host.tofile(shm)                            # the arrays travel through a shared-memory file, not the pipe
                    shape = tuple(host.shape)

                    def job(j):
                        step, name, _, box = items[j]
                        return {"shm": shm, "shape": shape, "index": j, "box": np.asarray(box, np.float32).reshape(-1)[:4], "masks": mask_files[step],
                                "png": self.log._png_path(name) if self._post_writes_png else None, "return_figs": self.return_figs}
                    # worker PROCESSES (the post-processing is ~30 ms of small numpy / torch-CPU calls per item, GIL-bound in threads);
                    # results are collected up to post_inflight batches later
                    inflight.append((items, [post_pool.submit(("ucb_post", job(j))) for j in range(len(items))], shm, slot))
                    tm["post_s"] += time.perf_counter() - t1
                    drain(keep=self.post_inflight)
                    return
                host = np.array(host)                               # the pinned buffer is reused; the in-process results may keep views

                def job(j):                                         # noqa: F811
                    step, name, _, box = items[j]
                    return {"im": host[j, ..., 0:3], "gt": host[j, ..., 3:6], "con": host[j, ..., 6:9], "mp": host[j, ..., 9:10],
                            "box": np.asarray(box, np.float32).reshape(-1)[:4], "masks": mask_files[step],
                            "png": self.log._png_path(name) if self._post_writes_png else None, "return_figs": self.return_figs}
                from .ucb_post import run_post_job
                if len(items) > 1 and self.post_threads > 1:
                    from concurrent.futures import ThreadPoolExecutor
                    with ThreadPoolExecutor(max_workers=min(self.post_threads, len(items))) as ex:
                        post = list(ex.map(lambda j: run_post_job(job(j)), range(len(items))))
                else:
                    post = [run_post_job(job(j)) for j in range(len(items))]
                tm["post_s"] += time.perf_counter() - t1
                finish(items, post)
                return
            if gpu_png and slot is not None and slot < 0:
                pin_busy[-1 - slot] = self.log.save_files(host, [it[1] for it in items])
            elif slot is not None:                  # the strips already sit in a shared-memory slot: the PNG workers read them there
                self.log.save_strips(host, [it[1] for it in items], parked=(ring.path(slot), lambda k=slot: ring.release(k)))
            else:
                strips = host if self.log.png_workers > 0 else np.array(host)      # the worker path copies into shared memory at once; writer threads keep the array
                self.log.save_strips(strips, [it[1] for it in items])
            for j, (step, name, _, box) in enumerate(items):
                self.log.display({}, 0, step, False, num_list)
                tm["items"] += 1
                results.append((name, [f[j:j + 1] for f in figs_b] if figs_b is not None else None))
            tm["png_s"] += time.perf_counter() - t1

        try:
            for step, img_name in enumerate(names):
                mine = lo <= step < hi
                if not mine and feed_is_sharded:
                    continue
                t0 = time.perf_counter()
                element = next(dataset.feed)
                tm["prep_wait_s"] += time.perf_counter() - t0
                if not mine:                    # a plain iterator cannot be sharded: its other ranks' elements are dropped
                    continue
                img = element[0]
                pending.append((step, _name(img_name), img, element[1] if len(element) > 1 else None))
                if post_dev is not None:
                    pend_masks[step] = element[3] if len(element) > 3 else None      # the item's packed masks when the loader decoded them
                if len(pending) >= batch:
                    submit()
            submit()
            while gpu_q:
                complete(gpu_q.pop(0))
            drain(keep=0)
        except BaseException:
            if on_gpu:
                try:
                    torch.cuda.synchronize()        # device-to-host copies of still-queued batches may be writing into ring slots / pinned buffers
                except BaseException:       # noqa: BLE001
                    pass
            self.close_pools()              # outstanding jobs of a failed loop are dropped with their workers
            for _, _, shm, slot_ in inflight:
                if slot_ is None:
                    try:
                        os.unlink(shm)
                    except OSError:
                        pass
            if ring_state[0] is not None:
                try:
                    self.log.flush()        # PNG workers may still be reading ring slots
                except BaseException:       # noqa: BLE001
                    pass
                ring_state[0].close()
            raise
        t0 = time.perf_counter()
        self.log.flush()
        tm["png_s"] += time.perf_counter() - t0
        if ring_state[0] is not None:
            tm["shm_ring_slots"] = len(ring_state[0].slots)
            ring_state[0].close()
        self._gather_losses(results, names, lo, rank, world, num_list)
        tm["total_s"] = time.time() - start
        if rank == 0:
            print('\n*****Time for epoch {} is {} sec*****'.format(1, int(time.time() - start)))
        return results

    def _gather_losses(self, results, names, lo: int, rank: int, world: int, num_list: int) -> None:
        """``all_losses`` = (name, losses) of every item in list order.  Data-parallel: one all_gather_object of the shards' lists;
        the running means are then re-accumulated in LIST order — the same floating-point sums as the single-process loop — into
        ``log.losses`` on every rank, and rank 0 prints the final progress line (utils.py:152-167)."""
        mine = [(lo + k, r[0], (r[2] if len(r) > 2 else {})) for k, r in enumerate(results)]
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            self.all_losses = [(n, l) for _, n, l in mine]
            return
        # under a process group the gather runs whatever its size (a one-rank RCCL group exercises the same path: tests/test_fsrnet.py)
        shards = [None] * world
        dist.all_gather_object(shards, mine, group=self.group)
        flat = sorted((it for sh in shards for it in sh), key=lambda it: it[0])
        if [it[0] for it in flat] != list(range(num_list)):
            raise RuntimeError("data-parallel loop: the ranks' shards do not tile the name list (%d items gathered for %d names)" % (len(flat), num_list))
        self.all_losses = [(n, l) for _, n, l in flat]
        acc: Dict[str, List[float]] = {}
        for _, _, losses in flat:
            Logging.accumulate(acc, losses)
        self.log.losses = acc
        if rank == 0:
            print(Logging.format_line(acc, num_list - 1, num_list), end='', flush=True)

    @staticmethod
    def _whole_batch(pending):
        """The elements of a device-prepared loader are views `out[i:i+1][None]` of one [B,S,S,16] tensor per group: when a batch is
        exactly such a group, in order, that tensor is the batch."""
        first = pending[0][2]
        base = getattr(first, "_base", None) if isinstance(first, torch.Tensor) else None
        if base is None or base.dim() != 4 or base.shape[0] != len(pending) or base.dtype != torch.float32 or not base.is_contiguous():
            return None
        step = base.stride(0) * base.element_size()
        for i, p in enumerate(pending):
            t = p[2]
            tb = t._base if isinstance(t, torch.Tensor) else None
            if tb is None or tb.data_ptr() != base.data_ptr() or tb.shape != base.shape or t.data_ptr() != base.data_ptr() + i * step or t.numel() != base.stride(0):
                return None
        return base

    def _split_row0(self, img) -> torch.Tensor:
        s = self.config.IMG_SIZE
        t = torch.as_tensor(np.asarray(img) if not isinstance(img, torch.Tensor) else img, dtype=torch.float32)
        return t.reshape(-1, s, s, t.shape[-1])[:1]

    def testFFHQ(self, dataset_val, batch: int = 16):
        """train_test_GSC.py:840-860.  ``dataset_val`` needs ``.feed`` (iterator of (img[1,10,256,256,16], box, name))
        and ``.name_list`` (dataset.py:29-30)."""
        return self._loop(dataset_val, batch, ucb=False)

    def test(self, dataset_val, batch: int = 16, postprocess: bool = True, mask_files=None):
        """train_test_GSC.py:360-408 + test_step :411-748.  Returns [(name, figs, {'ssim','psnr'})] with the reference's seven
        figures per item; ``postprocess=False`` returns the raw generator outputs [(name, [img, gs, con_rgb, dif, gt, face])].
        ``mask_files``: optional explicit per-item list (as ``_ucb_masks()`` returns it) instead of the folder listing."""
        # return_figs=False: nobody gets the figures back, so whoever post-processes an item (a pool worker, or this process when
        # post_workers == 0) writes its PNG strip itself — the reference always writes the strip (utils.py:196-204)
        self._post_writes_png = bool(postprocess and not self.return_figs)
        return self._loop(dataset_val, batch, ucb=True, postprocess=postprocess, mask_files=mask_files)


def roc_auc_score(labels: np.ndarray, scores: np.ndarray) -> float:
    """Area under the ROC curve = Mann-Whitney U with average ranks for ties (what sklearn.metrics.roc_auc_score computes
    for binary labels; the reference calls it at train_with_TSM.py:701)."""
    labels = np.asarray(labels).reshape(-1) > 0.5
    scores = np.asarray(scores, np.float64).reshape(-1)
    order = np.argsort(scores, kind="mergesort")
    s = scores[order]
    ranks = np.empty(len(s), np.float64)
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    r = np.empty_like(ranks)
    r[order] = ranks
    n_pos, n_neg = int(labels.sum()), int((~labels).sum())
    if n_pos == 0 or n_neg == 0:
        raise ValueError("roc_auc_score needs both classes")
    return float((r[labels].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


class FSRNetTSM(object):
    """Inference harness of the temporal-sharing variant (/root/reference/train_with_TSM.py:619-748): `testsfw`
    (image + mirror, frame = 2, shadow-segmentation AUC) and `testsfw_video` (10 video frames, frame = 10)."""
    SPLIT_SFW = (3, 3, 1, 3, 6, 1)        # img, cmap, mask, uv, reg, face   (train_with_TSM.py:675)
    SPLIT_VIDEO = (3, 3, 6, 1)            # img, uv, reg, face               (train_with_TSM.py:727)

    def __init__(self, config: Config, weights: Optional[Dict[str, np.ndarray]] = None, dtype: str = "f32"):
        from .model import GeneratorTSM
        self.config = config
        self.gen = GeneratorTSM(device=config.GPU_INDEX if torch.cuda.is_available() else None, dtype=dtype)
        if weights is not None:
            self.gen.load_weights(weights)
        self.log = Logging(config, png_threads=4)

    def _prep(self, img, n: int, split):
        s = self.config.IMG_SIZE
        t = torch.as_tensor(np.asarray(img) if not isinstance(img, torch.Tensor) else img, dtype=torch.float32)
        return torch.split(t.reshape(n, s, s, t.shape[-1]), list(split), dim=3)

    def _groups(self, elements, n: int, split):
        """Stack k elements of n coupled frames each into one [k*n, S, S, C] batch, split by channels."""
        s = self.config.IMG_SIZE
        ts = [torch.as_tensor(np.asarray(e) if not isinstance(e, torch.Tensor) else e, dtype=torch.float32).reshape(n, s, s, -1) for e in elements]
        return torch.split(torch.cat(ts, dim=0), list(split), dim=3)

    def test_steps_sfw(self, elements, training: bool = False):
        """train_with_TSM.py:668-707 for k elements in ONE forward: each element = [2,256,256,17] (image + mirror) is its own
        frame = 2 group — the ShareLayer couples only the frames of a group (model_with_TSM.py:204-229), so the k groups are independent
        and each element's outputs are those of its own forward, bit for bit.  -> [(losses, figs)] per element."""
        k = len(elements)
        im, cmap, mask, uv, reg, face = self._groups(elements, 2, self.SPLIT_SFW)
        dev = "cuda:%d" % self.gen._device
        _, con_rgb, _, mask_pred = self.gen(im.contiguous().to(dev), uv.contiguous().to(dev), reg.contiguous().to(dev), frame=2, share=True,
                                            chuck=1, training=training)
        mask_pred = mask_pred * face.to(dev)
        con_rgb = torch.clamp(con_rgb, 0, 1)
        im_d = im.to(dev)
        pred_host = mask_pred.detach().cpu()
        out = []
        for j in range(k):
            g = slice(2 * j, 2 * j + 2)
            m0 = mask[2 * j]
            label = (m0 == 2).float()                                               # :685
            pred0 = pred_host[2 * j]
            mse = float(((m0 - pred0) ** 2).mean())
            losses = {"psnr": float(10 * np.log10(1.0 / mse)) if mse > 0 else float("inf")}
            lab = np.concatenate([[1, 0], label.numpy().reshape(-1)])               # :688-692: one forced sample of each class
            sc = np.concatenate([[1, 0], pred0.numpy().reshape(-1)])
            losses["auc"] = roc_auc_score(lab, sc)
            out.append((losses, [im_d[g], con_rgb[g], mask_pred[g] * 2, label.reshape(1, *label.shape).to(dev)]))
        return out

    def test_step_sfw(self, img, box=None, training: bool = False):
        """train_with_TSM.py:668-707: element = [2,256,256,17] (image + mirror)."""
        return self.test_steps_sfw([img], training=training)[0]

    def test_steps_sfw_video(self, elements, training: bool = False):
        """train_with_TSM.py:720-748 for k elements in one forward: each element = [10,256,256,13] is its own frame = 10 group."""
        k = len(elements)
        im, uv, reg, face = self._groups(elements, 10, self.SPLIT_VIDEO)
        dev = "cuda:%d" % self.gen._device
        _, con_rgb, _, mask_pred = self.gen(im.contiguous().to(dev), uv.contiguous().to(dev), reg.contiguous().to(dev), frame=10, share=True,
                                            chuck=1, training=training)
        im_d, con_rgb, shown = im.to(dev), torch.clamp(con_rgb, 0, 1), mask_pred * face.to(dev) * 2
        return [({}, [im_d[10 * j:10 * j + 10], con_rgb[10 * j:10 * j + 10], shown[10 * j:10 * j + 10]]) for j in range(k)]

    def test_step_sfw_video(self, img, box=None, training: bool = False):
        """train_with_TSM.py:720-748: element = [10,256,256,13] (10 consecutive frames share features)."""
        return self.test_steps_sfw_video([img], training=training)[0]

    def _loop(self, dataset_val, steps_fn, batch: int = 1):
        """The reference's loop (train_with_TSM.py:619-666, 709-718) with `batch` ELEMENTS per forward (1 = the reference's own
        element-by-element form; config[4]'s "batch = 64 frames" is 32 SFW pairs or 6 ten-frame video groups)."""
        if self.gen._handle is None and self.gen.restore(self.config.CHECKPOINT_DIR) == 0 and self.gen._handle is None:
            raise RuntimeError("no generator weights: checkpoint data shard missing under %s" % self.config.CHECKPOINT_DIR)
        if batch < 1:
            raise ValueError("batch must be >= 1 element per forward")
        start = time.time()
        names = list(dataset_val.name_list)
        results = []
        group: List[Tuple[int, str, object]] = []

        def flush():
            if not group:
                return
            for (step, name, _), (losses, figs) in zip(group, steps_fn([g[2] for g in group], training=False)):
                self.log.display(losses, 0, step, False, len(names))
                self.log.save_img(figs, name)
                results.append((name, losses, figs))
            group.clear()
        try:
            for step, img_name in enumerate(names):
                element = next(dataset_val.feed)
                group.append((step, _name(img_name), element[0]))
                if len(group) >= batch:
                    flush()
            flush()
        finally:
            self.log.flush()
        print('\n*****Time for epoch {} is {} sec*****'.format(1, int(time.time() - start)))
        return results

    def testsfw(self, dataset_val, batch: int = 1):
        return self._loop(dataset_val, self.test_steps_sfw, batch)

    def testsfw_video(self, dataset_val, batch: int = 1):
        return self._loop(dataset_val, self.test_steps_sfw_video, batch)
