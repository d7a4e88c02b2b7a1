"""Experiment: one B=32 forward vs two concurrent B=16 forwards on two HIP streams (two handles).  python scratch/two_stream.py f32x3"""
import sys, time, torch
sys.path.insert(0, ".")
from blindshadowremoval_amd import Generator, init_weights
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
w = init_weights(1)
dev = torch.device("cuda", 0)
def mk(B):
    g = torch.Generator().manual_seed(5)
    return torch.rand(B, 256, 256, 3, generator=g).to(dev), torch.rand(B, 256, 256, 3, generator=g).to(dev)
def outs(B):
    return tuple(torch.empty((B, 256, 256, c), device=dev) for c in (1, 3, 3, 1))
def bench(nsplit, steps=20):
    B = 32 // nsplit
    gens = [Generator(dtype=dtype).load_weights(w) for _ in range(nsplit)]
    ins = [mk(B) for _ in range(nsplit)]
    os_ = [outs(B) for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    def run(n):
        for _ in range(n):
            for j in range(nsplit):
                with torch.cuda.stream(streams[j]):
                    gens[j](*ins[j], out=os_[j])
    run(5); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / steps)
    for g in gens: g.close()
    return best * 1e3
for ns in (1, 2, 4):
    print(dtype, "streams", ns, "ms per 32 images %.3f" % bench(ns))
