"""Dev script (GPU box): every dtype mode of the HIP forward vs the oracle — per-probe errors, flips, per-layer device time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blindshadowremoval_amd import Generator, init_weights
from oracle.gsc_oracle import GeneratorOracle

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f32", "f32x3", "f16"]
w = init_weights(1)
torch.manual_seed(0)
inp = torch.rand(B, 256, 256, 3); uv = torch.rand(B, 256, 256, 3)
pr = {}
ref = GeneratorOracle(w)(inp, uv, probes=pr)
names = {"x1": "x1", "x2": "x2", "x3": "x3", "x0": "x0", "res0": "res0", "res2": "res2", "up1": "up1", "up2": "up2", "y": "y", "d32": "d32",
         "res3": "res3", "res5": "res5", "f": "f"}
big_in, big_uv = torch.rand(32, 256, 256, 3).cuda(), torch.rand(32, 256, 256, 3).cuda()
for mode in modes:
    gen = Generator(dtype=mode).load_weights(w)
    out = gen(inp.cuda(), uv.cuda())
    torch.cuda.synchronize()
    print("==== mode", mode)
    for k, rk in names.items():
        a = gen.probe(k).cpu(); b = pr[rk]
        print("  %-6s max|ref| %8.3f  maxerr %.3e" % (k, b.abs().max(), (a - b).abs().max()))
    bm = gen.probe("bmask").cpu()
    flips = int((bm != pr["bmask"]).sum())
    ref2 = GeneratorOracle(w)(inp, uv, bmask_override=bm) if flips else ref
    for a, b, n in zip(out, ref2, ["gs", "con_rgb", "mask22", "dif"]):
        print("  %-8s maxerr %.3e (max|ref| %.3f)" % (n, (a.cpu() - b).abs().max(), b.abs().max()))
    print("  d32 margin %.3e  bmask flips %d" % (float((pr["d32"] - 0.1).abs().min()), flips))
    for _ in range(3): gen(big_in, big_uv)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): gen(big_in, big_uv)
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    print("  B=32  %.3f ms/forward  %.1f img/s" % (dt * 1e3, 32 / dt))
    gen.set_timing(True); gen(big_in, big_uv); torch.cuda.synchronize()
    print("  " + "  ".join("%s %.0f" % (n, ms * 1e3) for n, ms, _ in gen.get_launch_timing()))
    gen.close()
