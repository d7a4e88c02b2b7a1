"""Dev: event-timed bsr_png_unfilter on 32 photographs of the UCB fixtures (python scratch/unf_time.py)."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from blindshadowremoval_amd import _lib, prep, pngio
lib = _lib.load()
files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "UCB", "train", "input", "*", "*.png")))[:32]
raws = [pngio.read_rgb_raw(f) for f in files]
for n in (16, 32):
    items = [(r.raw, r.h, r.w, r.c) for r in raws[:n]]
    tab = np.zeros(n, prep.UNFILTER_DTYPE); off = tab.nbytes
    for k, (raw, h, w, c) in enumerate(items):
        tab[k] = (off, 0, h, w, c, 0); off = (off + raw.size + 7) & ~7
    for k, (raw, h, w, c) in enumerate(items):
        tab[k]["out_off"] = off; off = (off + h * w * 3 + 7) & ~7
    blob = np.zeros(off, np.uint8); blob[:tab.nbytes] = tab.view(np.uint8)
    for k, (raw, h, w, c) in enumerate(items):
        blob[tab[k]["raw_off"]:tab[k]["raw_off"] + raw.size] = raw.reshape(-1)
    d = torch.from_numpy(blob).cuda()
    fn = lambda: lib.bsr_png_unfilter(0, d.data_ptr(), d.numel(), 0, n, torch.cuda.current_stream().cuda_stream)
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(20):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); print("bsr_png_unfilter, %d images of 256x256x3: median %.3f ms" % (n, ts[len(ts) // 2]))
