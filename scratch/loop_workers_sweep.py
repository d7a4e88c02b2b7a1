"""Dev: rate of the round-5 device loops against the number of loader workers (python scratch/loop_workers_sweep.py ucb|ffhq items w1 w2 ...)."""
import contextlib, io, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blindshadowremoval_amd.dataset import Dataset, cpu_share
from blindshadowremoval_amd.fsrnet import Config, FSRNet
from blindshadowremoval_amd.weights import init_weights
BATCH = int(os.environ.get("LOOP_BATCH", "16"))
kind = sys.argv[1]
items = int(sys.argv[2])
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ucb = kind == "ucb"
cfg = Config(0)
cfg.CHECKPOINT_DIR = tempfile.mkdtemp(prefix="bsr_lw_")
cfg.DATA_DIR_TEST = [os.path.join(G, "UCB", "train", "input", "*") if ucb else os.path.join(G, "sample_imgs", "*")]
cfg.UCB_MASK_ROOT = os.path.join(G, "UCB_masks")
fsr = FSRNet(cfg, weights=init_weights(1))
fsr.return_figs = False
print("cpu_share", cpu_share())
for rep in range(2):
    for workers in [int(w) for w in sys.argv[3:]]:
        ds = Dataset(cfg, "test", ucb=ucb, workers=workers, device_prep=0, device_batch=BATCH)
        base = list(ds.name_list)
        reps = (items + len(base) - 1) // len(base)
        ds.name_list = (base * reps)[:items]
        masks = (fsr._ucb_masks()[:len(base)] * reps)[:items] if ucb else None
        ds.warm(); fsr.log.warm(); fsr.warm_pools()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            out = fsr.test(ds, batch=BATCH, mask_files=masks) if ucb else fsr.testFFHQ(ds, batch=BATCH)
        dt = time.perf_counter() - t0
        print(kind, "workers", workers, "items", len(out), round(len(out) / dt, 1), "/s", {k: round(v, 3) for k, v in fsr.timings.items() if k.endswith("_s")}, flush=True)
        ds.close()
fsr.close()
