import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blindshadowremoval_amd.dataset import Dataset
from blindshadowremoval_amd.fsrnet import Config
from blindshadowremoval_amd import prep
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = Config(0); cfg.DATA_DIR_TEST = [os.path.join(G, "sample_imgs", "*")]
W = int(sys.argv[1]); N = 1600
ds = Dataset(cfg, "test", workers=W, device_prep=0, device_batch=16)
ds.name_list = ds.name_list * N
ds.warm()
t0 = time.perf_counter(); parts = list(ds._iterate_host()); t1 = time.perf_counter()
print("host parts", round(N / (t1 - t0)), "items/s")
dp = prep.DevicePrep(0, 256)
torch.cuda.synchronize()
for label, sync_each in (("async", False), ("sync each", True)):
    tp = tk = 0.0
    t0 = time.perf_counter()
    for g in range(0, N, 16):
        a = time.perf_counter(); out, _ = dp.rows(parts[g:g + 16]); tp += time.perf_counter() - a
        if sync_each:
            a = time.perf_counter(); torch.cuda.synchronize(); tk += time.perf_counter() - a
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(label, "dp.rows over distinct parts:", round(N / dt), "items/s; host", round(tp / (N / 16) * 1e3, 2), "ms/batch, wait", round(tk / (N / 16) * 1e3, 2), "ms/batch")
same = [parts[0]] * 16
t0 = time.perf_counter()
for g in range(0, N, 16): out, _ = dp.rows(same)
torch.cuda.synchronize(); print("same part x16:", round(N / (time.perf_counter() - t0)), "items/s")
ds.close()
