"""Dev: cProfile of the loop's own thread in the round-5 device modes (python scratch/loop_profile.py ucb|ffhq [items])."""
import cProfile, contextlib, io, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blindshadowremoval_amd.dataset import Dataset, cpu_share
from blindshadowremoval_amd.fsrnet import Config, FSRNet
from blindshadowremoval_amd.weights import init_weights
kind = sys.argv[1] if len(sys.argv) > 1 else "ucb"
items = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
workers = int(sys.argv[3]) if len(sys.argv) > 3 else cpu_share()
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ucb = kind == "ucb"
cfg = Config(0)
cfg.CHECKPOINT_DIR = tempfile.mkdtemp(prefix="bsr_lp_")
cfg.DATA_DIR_TEST = [os.path.join(G, "UCB", "train", "input", "*") if ucb else os.path.join(G, "sample_imgs", "*")]
cfg.UCB_MASK_ROOT = os.path.join(G, "UCB_masks")
fsr = FSRNet(cfg, weights=init_weights(1))
fsr.return_figs = False
ds = Dataset(cfg, "test", ucb=ucb, workers=workers, device_prep=0, device_batch=16)
base = list(ds.name_list)
reps = (items + len(base) - 1) // len(base)
ds.name_list = (base * reps)[:items]
masks = (fsr._ucb_masks()[:len(base)] * reps)[:items] if ucb else None
ds.warm(); fsr.log.warm(); fsr.warm_pools()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    out = fsr.test(ds, batch=16, mask_files=masks) if ucb else fsr.testFFHQ(ds, batch=16)
pr.disable()
dt = time.perf_counter() - t0
print(kind, "workers", workers, len(out), "items", round(len(out) / dt, 1), "/s", {k: round(v, 3) for k, v in fsr.timings.items() if k.endswith("_s")})
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
ds.close(); fsr.close()
