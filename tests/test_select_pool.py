"""_SelectPool (dataset.py): the process pool behind the loader, the PNG writers and the UCB post-processing.  A job that fails
in its worker must raise ONCE, for its own ticket, and leave the worker's other tickets valid and in order (a PNG write error —
disk full, bad CHECKPOINT_DIR — must not hang or mis-credit the rest of the loop)."""
import os

import numpy as np
import pytest

from blindshadowremoval_amd.dataset import _SelectPool
from blindshadowremoval_amd.fsrnet import Config, Logging


def _strip(v):
    return np.full((8, 24, 3), v, np.uint8)


def _blocked(tmp_path):
    """A path no PNG can be written to: its parent 'directory' is a regular file (write_png creates missing directories itself)."""
    f = tmp_path / "a_file"
    f.write_text("not a directory")
    return str(f / "x.png")


@pytest.mark.timeout(120)
def test_failed_job_raises_once_and_the_rest_still_return(tmp_path):
    pool = _SelectPool(1)
    try:
        bad = pool.submit(("png", _blocked(tmp_path), _strip(1)))
        good1 = pool.submit(("png", str(tmp_path / "b.png"), _strip(2)))
        good2 = pool.submit(("png", str(tmp_path / "c.png"), _strip(3)))
        with pytest.raises(RuntimeError, match="loader worker failed"):
            pool.result(bad)
        assert pool.result(good1) is True
        assert pool.result(good2) is True                   # used to block forever: the failed reply never advanced the ticket queue
        assert os.path.isfile(tmp_path / "b.png") and os.path.isfile(tmp_path / "c.png")
        assert pool._load == [0] and pool._owner == [[]] and not pool._done
        # tickets collected out of order, failure in the middle
        t = [pool.submit(("png", str(tmp_path / ("d%d.png" % i)) if i != 1 else _blocked(tmp_path), _strip(i))) for i in range(4)]
        assert pool.result(t[3]) is True and pool.result(t[0]) is True
        with pytest.raises(RuntimeError):
            pool.result(t[1])
        assert pool.result(t[2]) is True
    finally:
        pool.shutdown()


@pytest.mark.timeout(120)
def test_imap_raises_at_the_failed_element_in_order(tmp_path):
    pool = _SelectPool(2)
    try:
        jobs = [("png", str(tmp_path / ("e%d.png" % i)) if i != 2 else _blocked(tmp_path), _strip(i)) for i in range(5)]
        got = []
        with pytest.raises(RuntimeError, match="loader worker failed"):
            for v in pool.imap(jobs, depth=4):
                got.append(v)
        assert got == [True, True]
    finally:
        pool.shutdown()


@pytest.mark.timeout(120)
def test_logging_flush_reports_a_failed_strip_without_hanging(tmp_path):
    cfg = Config(0)
    cfg.CHECKPOINT_DIR = str(tmp_path / "ckpt")
    os.makedirs(os.path.join(cfg.CHECKPOINT_DIR, "test"))
    log = Logging(cfg, png_workers=1)
    strips = np.stack([_strip(i) for i in range(3)])
    try:
        log.save_strips(strips, ["ok/one.npy", "ok/two.npy", "ok/three.npy"])
        log.flush()
        assert all(os.path.isfile(p) for p in log.saved)
        # make the target directory unwritable by replacing it with a file
        import shutil
        shutil.rmtree(os.path.join(cfg.CHECKPOINT_DIR, "test"))
        with open(os.path.join(cfg.CHECKPOINT_DIR, "test"), "w") as f:
            f.write("not a directory")
        log.save_strips(strips, ["bad/one.npy", "bad/two.npy", "bad/three.npy"])
        with pytest.raises(RuntimeError, match="loader worker failed"):
            log.flush()
        assert not log._shm_batches                             # the shared-memory file of the failed batch was released
        log.flush()                                             # nothing left to wait for
    finally:
        log.close()
