"""Probe (round 4): does the PHASE between the two lanes matter?  Two forwards in flight run the same launch sequence; started together
they are in the 1/8-resolution trunk at the same time (one-round grids, attention's 150-KB workgroups cannot share a CU with each
other) and in the decoders at the same time.  A one-time spin on lane 1 before the first step shifts it by a fraction of a forward.
python scratch/lane_phase.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blindshadowremoval_amd import Generator, init_weights
from blindshadowremoval_amd.lanes import concurrent_streams
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = 32
w = init_weights(1)
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(1234)
inp = torch.rand(B, 256, 256, 3, generator=g).to(dev)
uv = torch.rand(B, 256, 256, 3, generator=g).to(dev)
gens = [Generator(device=0).load_weights(w) for _ in range(2)]
lanes, ok = concurrent_streams(0, 2)
outs = [tuple(torch.empty((B, 256, 256, c), device=dev) for c in (1, 3, 3, 1)) for _ in range(2)]
print("streams seen to overlap:", ok)
CYC_PER_MS = 2_000_000      # torch.cuda._sleep counts ~2 GHz cycles (lanes.py: 4e6 ~ 2 ms)


def run(n, offset_ms):
    if offset_ms > 0:
        with torch.cuda.stream(lanes[1]):
            torch.cuda._sleep(int(offset_ms * CYC_PER_MS))
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(lanes[k]):
            gens[k](inp, uv, out=outs[k])


for off in (0.0, 1.2, 2.4, 3.6, 0.0, 2.4, 4.8):
    run(6, 0.0)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        run(steps, off)
        torch.cuda.synchronize()
        res.append(B * steps / (time.perf_counter() - t0))       # wall clock: lane 0 works while lane 1 spins, so the spin is NOT subtracted
    print("lane-1 offset %.1f ms: %s images/s (wall clock)" % (off, ", ".join("%.0f" % r for r in res)), flush=True)
