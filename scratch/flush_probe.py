import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from blindshadowremoval_amd import Generator, init_weights
from blindshadowremoval_amd.fsrnet import Logging, Config
gen = Generator().load_weights(init_weights(1))
rows_list = [torch.rand(1, 1, 256, 256, 16, device="cuda") for _ in range(16)]
def sync(): torch.cuda.synchronize()
T = {}
def tick(name, t0):
    sync(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
for it in range(30):
    t0 = time.perf_counter(); rows = torch.cat([r.reshape(-1, 256, 256, 16)[:1] for r in rows_list], dim=0); tick("cat", t0)
    t0 = time.perf_counter(); im, gt, uv, _, face = torch.split(rows, [3, 3, 3, 6, 1], dim=3); gs, con, _, mp = gen(im, uv); tick("gen", t0)
    t0 = time.perf_counter(); figs = [im, torch.clamp(con, 0, 1), mp * face * 2]; tick("figs", t0)
    t0 = time.perf_counter()
    cols = []
    for f in figs:
        a = torch.clamp(f.detach().float(), 0.0, 1.0) * 255.0
        cols.append(a.expand(-1, -1, -1, 3) if a.shape[3] == 1 else a[..., :3])
    u8 = torch.round(torch.cat(cols, dim=2)).to(torch.uint8); tick("strip_gpu", t0)
    t0 = time.perf_counter(); h = u8.cpu(); tick("d2h", t0)
    t0 = time.perf_counter(); n = h.numpy(); tick("numpy", t0)
print({k: round(v / 30 * 1e3, 3) for k, v in T.items()}, "ms per batch of 16")
pin = torch.empty((16, 256, 768, 3), dtype=torch.uint8).pin_memory()
t0 = time.perf_counter()
for _ in range(30): pin.copy_(u8, non_blocking=True); sync()
print("d2h pinned ms", (time.perf_counter() - t0) / 30 * 1e3)
