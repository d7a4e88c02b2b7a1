"""Dev: upper bound of what fusing the strip assembly into the PNG encoder can give the FFHQ loop: the same loop with strips_on_device
replaced by a constant tensor (no elementwise launches).  python scratch/strip_bound.py [items] [workers]"""
import contextlib, io, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import blindshadowremoval_amd.fsrnet as F
from blindshadowremoval_amd.dataset import Dataset
from blindshadowremoval_amd.weights import init_weights
items = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 10
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = F.Config(0)
cfg.CHECKPOINT_DIR = tempfile.mkdtemp(prefix="bsr_sb_")
cfg.DATA_DIR_TEST = [os.path.join(G, "sample_imgs", "*")]
fsr = F.FSRNet(cfg, weights=init_weights(1))
fsr.return_figs = False
orig = F.Logging.strips_on_device
const = {}
def fake(figs):
    B = figs[0].shape[0]
    if B not in const:
        const[B] = orig(figs)
    return const[B]
for rep in range(3):
    for label, fn in (("real", orig), ("constant strips", fake)):
        F.Logging.strips_on_device = staticmethod(fn)
        ds = Dataset(cfg, "test", workers=workers, device_prep=0, device_batch=16)
        base = list(ds.name_list)
        ds.name_list = (base * items)[:items]
        ds.warm(); fsr.log.warm(); fsr.warm_pools()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            out = fsr.testFFHQ(ds, batch=16)
        dt = time.perf_counter() - t0
        print(label, len(out), "items", round(len(out) / dt, 1), "/s", flush=True)
        ds.close()
fsr.close()
