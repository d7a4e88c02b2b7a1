"""Probe: throughput of back-to-back forwards on ONE stream vs alternating over TWO handles on TWO streams (the second forward's
kernels fill the tails / ramps of the first's one-round launches).  python scratch/two_stream.py [B] [dtype] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blindshadowremoval_amd import Generator, init_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
w = init_weights(1)
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(1234)
inp = torch.rand(B, 256, 256, 3, generator=g).to(dev)
uv = torch.rand(B, 256, 256, 3, generator=g).to(dev)
for nstream in (1, 2, 3, 1, 2):
    gens = [Generator(device=0, dtype=dtype).load_weights(w) for _ in range(nstream)]
    streams = [torch.cuda.Stream(priority=0) for _ in range(nstream)]
    outs = [tuple(torch.empty((B, 256, 256, c), device=dev) for c in (1, 3, 3, 1)) for _ in range(nstream)]
    def run(n):
        for i in range(n):
            k = i % nstream
            with torch.cuda.stream(streams[k]):
                gens[k](inp, uv, out=outs[k])
    run(2 * nstream + 2)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        res.append(B * steps / (time.perf_counter() - t0))
    ref = outs[0]
    same = all(torch.equal(a, b) for o in outs[1:] for a, b in zip(ref, o))
    print("streams %d: %s images/s   (outputs identical across handles: %s)" % (nstream, ", ".join("%.0f" % r for r in res), same), flush=True)
    del gens
