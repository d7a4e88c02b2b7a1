"""Probe: how many warm-up forwards does a two-handle / two-stream loop need before the overlap gain shows?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blindshadowremoval_amd import Generator, init_weights
B, dtype = 32, (sys.argv[1] if len(sys.argv) > 1 else "f32")
w = init_weights(1)
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(1234)
inp = torch.rand(B, 256, 256, 3, generator=g).to(dev)
uv = torch.rand(B, 256, 256, 3, generator=g).to(dev)
for warm in (3, 3, 8):
    gens = [Generator(device=0, dtype=dtype).load_weights(w) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = [tuple(torch.empty((B, 256, 256, c), device=dev) for c in (1, 3, 3, 1)) for _ in range(2)]
    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i & 1]):
                gens[i & 1](inp, uv, out=outs[i & 1])
    run(warm)
    torch.cuda.synchronize()
    res = []
    for rep in range(6):
        t0 = time.perf_counter()
        run(10)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        res.append("%.0f (issue %.1f ms)" % (B * 10 / (time.perf_counter() - t0), t_issue * 1e3))
    print("warm-up %d forwards: regions of 10 steps -> %s" % (warm, ", ".join(res)), flush=True)
    del gens
