"""Probe: which pairs of torch streams really run concurrently (distinct hardware queues)?  A spin kernel on each of two streams:
elapsed ~1x = concurrent, ~2x = serialised on one queue.  Then forwards on (normal, normal) vs (normal, high-priority) pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda", 0)
torch.cuda._sleep(1000); torch.cuda.synchronize()
def pair_time(a, b, cyc=40_000_000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a): torch.cuda._sleep(cyc)
    with torch.cuda.stream(b): torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
ss = [torch.cuda.Stream() for _ in range(10)]
hi = [torch.cuda.Stream(priority=-1) for _ in range(4)]
one = pair_time(ss[0], ss[0]) / 2
print("one spin kernel: %.1f ms" % one)
print("normal pairs (i, i+1):", ["%.2f" % (pair_time(ss[i], ss[i + 1]) / one) for i in range(9)])
print("normal pairs (0, j):  ", ["%.2f" % (pair_time(ss[0], ss[j]) / one) for j in range(1, 10)])
print("normal i / high j:    ", ["%.2f" % (pair_time(ss[i], hi[j]) / one) for i in range(3) for j in range(4)])
print("default stream / normal j:", ["%.2f" % (pair_time(torch.cuda.default_stream(), ss[j]) / one) for j in range(6)])
from blindshadowremoval_amd import Generator, init_weights
w = init_weights(1)
B = 32
g = torch.Generator(device="cpu").manual_seed(1234)
inp = torch.rand(B, 256, 256, 3, generator=g).to(dev)
uv = torch.rand(B, 256, 256, 3, generator=g).to(dev)
gens = [Generator(device=0).load_weights(w) for _ in range(2)]
outs = [tuple(torch.empty((B, 256, 256, c), device=dev) for c in (1, 3, 3, 1)) for _ in range(2)]
def rate(lanes, n=30):
    def run(k):
        for i in range(k):
            with torch.cuda.stream(lanes[i & 1]):
                gens[i & 1](inp, uv, out=outs[i & 1])
    run(4); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(n); torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)
for name, lanes in [("ss0,ss1", (ss[0], ss[1])), ("ss2,ss3", (ss[2], ss[3])), ("ss4,ss5", (ss[4], ss[5])), ("ss0,hi0", (ss[0], hi[0])), ("ss1,hi1", (ss[1], hi[1])),
                    ("ss3,hi2", (ss[3], hi[2])), ("default,ss1", (torch.cuda.default_stream(), ss[1])), ("default,hi0", (torch.cuda.default_stream(), hi[0])), ("ss0,ss0", (ss[0], ss[0]))]:
    print("%-12s spin ratio %.2f  forwards %.0f images/s" % (name, pair_time(*lanes) / one, rate(lanes)), flush=True)
