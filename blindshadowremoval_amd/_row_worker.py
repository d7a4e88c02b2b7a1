"""Worker process of the test-time loader (dataset.Dataset(workers=N)) and of the UCB post-processing pool
(FSRNet.test(post_workers=N)): reads pickled jobs from stdin, writes pickled results to stdout, both length-prefixed.  Started as
`python -m blindshadowremoval_amd._row_worker`, so it never depends on the parent's __main__ module.  Loader jobs need numpy / PIL /
matplotlib.tri only; a ("ucb_post", ...) job imports torch for its CPU resize / SSIM — the worker never touches a GPU (the pool
starts it with no visible device)."""
import os
import pickle
import struct
import sys


def _read_exact(f, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = f.read(n - len(buf))
        if not chunk:
            return None
        buf += chunk
    return bytes(buf)


def main() -> int:
    fin = os.fdopen(os.dup(0), "rb")
    fout = os.fdopen(os.dup(1), "wb")
    os.dup2(2, 1)                       # anything a library prints goes to stderr, never into the result stream
    from blindshadowremoval_amd.dataset import build_element
    while True:
        head = _read_exact(fin, 8)
        if head is None:
            return 0
        job = pickle.loads(_read_exact(fin, struct.unpack("<Q", head)[0]))
        try:
            if isinstance(job, tuple) and job and job[0] == "warm":                  # import what the jobs of this pool will need, before the clock runs
                import time
                if job[1] == "post":
                    import blindshadowremoval_amd.ucb_post  # noqa: F401
                else:
                    import matplotlib.tri  # noqa: F401
                    import PIL.Image  # noqa: F401
                    if len(job) > 2:                                                 # map the parent's shared-memory ring while its file still has a name
                        import numpy as np
                        from blindshadowremoval_amd import prep
                        prep._RING_VIEWS[job[2]] = np.memmap(job[2], np.uint8, "r+")
                time.sleep(0.2)                                                      # keeps this worker busy so that the pool starts the others too
                result = True
            elif isinstance(job, tuple) and job and job[0] == "png":                 # ("png", path, uint8 strip | (shm file, shape, index)): Logging(png_workers=N)
                import numpy as np
                from blindshadowremoval_amd.pngio import write_png
                strip = job[2]
                if isinstance(strip, tuple):                                         # one strip of a batch the parent parked in shared memory
                    shm, shape, idx = strip
                    n = int(np.prod(shape[1:]))
                    strip = np.fromfile(shm, np.uint8, count=n, offset=idx * n).reshape(shape[1:])
                write_png(job[1], strip)
                result = True
            elif isinstance(job, tuple) and job and job[0] == "ucb_post":
                from blindshadowremoval_amd.ucb_post import run_post_job
                result = run_post_job(job[1])
            else:
                result = build_element(job)
            payload = pickle.dumps(("ok", result), protocol=pickle.HIGHEST_PROTOCOL)
        except Exception as e:          # reported to the parent, which re-raises
            payload = pickle.dumps(("err", "%s: %s" % (type(e).__name__, e)), protocol=pickle.HIGHEST_PROTOCOL)
        fout.write(struct.pack("<Q", len(payload)))
        fout.write(payload)
        fout.flush()


if __name__ == "__main__":
    sys.exit(main())
