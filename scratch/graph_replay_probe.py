import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blindshadowremoval_amd import Generator, init_weights


def main(B=32):
    gen = Generator().load_weights(init_weights(1))
    torch.manual_seed(0)
    inp = torch.rand(B, 256, 256, 3).cuda(); uv = torch.rand(B, 256, 256, 3).cuda()
    outs = tuple(torch.empty((B, 256, 256, c), device="cuda") for c in (1, 3, 3, 1))
    for _ in range(3): gen(inp, uv, out=outs)
    torch.cuda.synchronize()
    def timeit(fn, n=20):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
    print("B", B, "eager ms", timeit(lambda: gen(inp, uv, out=outs)))
    ref = [o.clone() for o in outs]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gen(inp, uv, out=outs)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            gen(inp, uv, out=outs)
    torch.cuda.current_stream().wait_stream(s)
    for o in outs: o.zero_()
    g.replay(); torch.cuda.synchronize()
    print("graph output equal:", all(torch.equal(a, b) for a, b in zip(outs, ref)))
    print("graph ms", timeit(lambda: g.replay()))
    print("eager ms", timeit(lambda: gen(inp, uv, out=outs)))


if __name__ == "__main__":
    for b in ([int(a) for a in sys.argv[1:]] or [32]):
        main(b)
