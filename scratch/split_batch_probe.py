"""Round 5 probe: ONE batch of 32 as two half batches on two handles / two streams (16 + 16 resident = the same 32 images) against the
one-launch-sequence forward.  Decides whether an internal batch split is worth building into bsr_forward."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from blindshadowremoval_amd import Generator, init_weights
from blindshadowremoval_amd.lanes import concurrent_streams

w = init_weights(1)
res = {}
for dtype in ("f32", "f32x3", "f16"):
    g = [Generator(device=0, dtype=dtype).load_weights(w) for _ in range(2)]
    lanes, ok = concurrent_streams(0, 2)
    inp, uv = torch.rand(32, 256, 256, 3).cuda(), torch.rand(32, 256, 256, 3).cuda()
    out = [tuple(torch.empty((n, 256, 256, c), device="cuda") for c in (1, 3, 3, 1)) for n in (32, 16, 16)]

    def whole():
        g[0](inp, uv, out=out[0])

    def split():
        ev = torch.cuda.Event()
        ev.record()
        for k in range(2):
            with torch.cuda.stream(lanes[k]):
                lanes[k].wait_event(ev)
                g[k](inp[16 * k:16 * k + 16], uv[16 * k:16 * k + 16], out=out[1 + k])
        for k in range(2):
            torch.cuda.current_stream().wait_stream(lanes[k])

    r = {}
    for name, fn in (("whole", whole), ("split", split), ("whole2", whole), ("split2", split)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        r[name] = round(32 * 30 / (time.perf_counter() - t0), 1)
    whole(); split(); torch.cuda.synchronize()
    r["bit_identical"] = all(torch.equal(out[0][i][:16], out[1][i]) and torch.equal(out[0][i][16:], out[2][i]) for i in range(4))
    r["streams_overlap"] = ok
    res[dtype] = r
    for x in g:
        x.close()
print(json.dumps(res))
