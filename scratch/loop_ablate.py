"""Dev: the FFHQ / UCB loop WITHOUT its loader (elements served from HBM by a stand-in dataset): what the loop's own thread + the GPU + the
file writers sustain when input preparation costs nothing.  python scratch/loop_ablate.py ffhq|ucb items [nowrite]"""
import contextlib, io, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from blindshadowremoval_amd.dataset import Dataset
from blindshadowremoval_amd.fsrnet import Config, FSRNet
from blindshadowremoval_amd.weights import init_weights
kind, items = sys.argv[1], int(sys.argv[2])
nowrite = len(sys.argv) > 3 and sys.argv[3] == "nowrite"
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ucb = kind == "ucb"
cfg = Config(0)
cfg.CHECKPOINT_DIR = tempfile.mkdtemp(prefix="bsr_la_")
cfg.DATA_DIR_TEST = [os.path.join(G, "UCB", "train", "input", "*") if ucb else os.path.join(G, "sample_imgs", "*")]
cfg.UCB_MASK_ROOT = os.path.join(G, "UCB_masks")
fsr = FSRNet(cfg, weights=init_weights(1))
fsr.return_figs = False
real = Dataset(cfg, "test", ucb=ucb, workers=4, device_prep=0, device_batch=16)
base = list(real.name_list)
masks = fsr._ucb_masks() if ucb else None
if ucb:
    real.ucb_mask_files = masks
real.name_list = (base * 16)[:16]
if ucb:
    real.ucb_mask_files = (masks * 16)[:16]
elems = [tuple(e) for e in real.feed]
elems = [(e[0].clone(), e[1], e[2]) + (((e[3][0], e[3][1].clone(), e[3][2]),) if len(e) > 3 else ()) for e in elems]
real.close()


class Served:
    def __init__(self, n):
        self.name_list = ["/x/item%05d.npy" % i for i in range(n)]
        self.feed = self._gen(n)
        self.device_prep = 0
        self._started = True

    def _gen(self, n):
        for i in range(n):
            e = elems[i % 16]
            yield (e[0], e[1], np.array([("/x/gt/item%05d.png" % i).encode()])) + tuple(e[3:])

    def poll(self):
        pass


if nowrite:
    import blindshadowremoval_amd.fsrnet as F
    orig = F.Logging.save_files if hasattr(F.Logging, "save_files") else None
    def fake(self, files, names):
        return []
    F.Logging.save_files = fake
fsr.log.warm(); fsr.warm_pools()
for rep in range(3):
    ds = Served(items)
    mk = (masks * (items // len(masks) + 1))[:items] if ucb else None
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        out = fsr.test(ds, batch=16, mask_files=mk) if ucb else fsr.testFFHQ(ds, batch=16)
    dt = time.perf_counter() - t0
    print(kind, "served from HBM", "nowrite" if nowrite else "", len(out), "items", round(len(out) / dt, 1), "/s", {k: round(v, 3) for k, v in fsr.timings.items() if k.endswith("_s")}, flush=True)
fsr.close()
