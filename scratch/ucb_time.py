"""Dev: event-timed bsr_ucb_post on 16 items of the UCB fixtures (python scratch/ucb_time.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ucb_cases import cases
from blindshadowremoval_amd.ucb_post_gpu import UcbPostDevice, MASK_ORDER
base = list(cases())
batch = [base[i % len(base)] for i in range(16)]
mu8 = lambda masks: np.stack([np.rint(masks[k][:, :, 0] * 255.0).astype(np.uint8) for k in MASK_ORDER], axis=0)
rows10 = torch.from_numpy(np.stack([np.concatenate([row[..., 0:3], row[..., 3:6], con, dif], axis=2) for _, row, _, _, con, dif in batch])).cuda()
masks = torch.from_numpy(np.stack([mu8(m) for _, _, _, m, _, _ in batch])).cuda()
boxes = torch.from_numpy(np.stack([np.asarray(b, np.float32).reshape(4) for _, _, b, _, _, _ in batch])).cuda()
post = UcbPostDevice(0)
for figs in (False, True):
    fn = lambda: post.run(rows10, masks, boxes, want_figs=figs)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print("bsr_ucb_post, 16 items, want_figs=%s: median %.3f ms (min %.3f)" % (figs, ts[len(ts) // 2], ts[0]))
