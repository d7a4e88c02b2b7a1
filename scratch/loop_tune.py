"""Tuning probe for the end-to-end loops (not product code): python scratch/loop_tune.py ffhq|ucb workers png_threads post_workers inflight switch_us items"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tempfile, shutil
kind, workers, png_threads, post_workers, inflight, switch_us, items = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
if switch_us > 0:
    sys.setswitchinterval(switch_us * 1e-6)
from blindshadowremoval_amd.dataset import Dataset
from blindshadowremoval_amd.fsrnet import Config, FSRNet
from blindshadowremoval_amd.weights import init_weights
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ucb = kind == "ucb"
cfg = Config(0)
out_dir = tempfile.mkdtemp(prefix="bsr_loop_")
cfg.CHECKPOINT_DIR = out_dir
cfg.DATA_DIR_TEST = [os.path.join(G, "UCB", "train", "input", "*") if ucb else os.path.join(G, "sample_imgs", "*")]
cfg.UCB_MASK_ROOT = os.path.join(G, "UCB_masks")
fsr = FSRNet(cfg, weights=init_weights(1))
fsr.post_workers, fsr.post_inflight, fsr.return_figs = post_workers, inflight, False
fsr.log.png_workers = png_threads
ds = Dataset(cfg, "test", ucb=ucb, workers=workers, device_prep=0, device_batch=16)
base = list(ds.name_list)
reps = (items + len(base) - 1) // len(base)
ds.name_list = (base * reps)[:items]
masks = (fsr._ucb_masks()[:len(base)] * reps)[:items] if ucb else None
ds.warm()
fsr.log.warm()
if ucb:
    fsr.warm_pools()
prof = None
if os.environ.get("BSR_PROFILE"):
    import cProfile
    prof = cProfile.Profile()
    prof.enable()
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    out = fsr.test(ds, batch=16, mask_files=masks) if ucb else fsr.testFFHQ(ds, batch=16)
dt = time.perf_counter() - t0
if prof is not None:
    import pstats
    prof.disable()
    pstats.Stats(prof).sort_stats("tottime").print_stats(28)
tm = fsr.timings
print(kind, "workers", workers, "png", png_threads, "post", post_workers, "inflight", inflight, "switch_us", switch_us, "->", round(len(out) / dt, 1), "img/s; steady",
      round((len(out) - 16) / (dt - tm["first_batch_done_s"]), 1), {k: round(v, 2) for k, v in tm.items() if k.endswith("_s")})
shutil.rmtree(out_dir, ignore_errors=True)
fsr.close()
