"""Dev script: HIP forward vs oracle with per-probe errors (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from blindshadowremoval_amd import Generator, init_weights
from oracle.gsc_oracle import GeneratorOracle

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
w = init_weights(1)
torch.manual_seed(0)
inp = torch.rand(B, 256, 256, 3); uv = torch.rand(B, 256, 256, 3)
pr = {}
t = time.time(); ref = GeneratorOracle(w)(inp, uv, probes=pr); print("oracle s", time.time() - t)
gen = Generator().load_weights(w)
out = gen(inp.cuda(), uv.cuda())
torch.cuda.synchronize()
names = {"x1": "x1", "x2": "x2", "x3": "x3", "x0": "x0", "att0": "res_stack/0/non_local/att", "res0": "res0",
         "res1": "res1", "res2": "res2", "up1": "up1", "up2": "up2", "y": "y", "d32": "d32", "bmask": "bmask", "res3": "res3",
         "att5": "res_stack/5/non_local/att", "res5": "res5", "f": "f"}
for k, rk in names.items():
    a = gen.probe(k).cpu(); b = pr[rk]
    print("%-6s %-20s max|ref| %.3f  maxerr %.3e" % (k, tuple(a.shape), b.abs().max(), (a - b).abs().max()))
for a, b, n in zip(out, ref, ["gs", "con_rgb", "mask22", "dif"]):
    print("%-8s maxerr %.3e (max|ref| %.3f)" % (n, (a.cpu() - b).abs().max(), b.abs().max()))
print("d32 margin", float((pr["d32"] - 0.1).abs().min()), "bmask flips", int((gen.probe("bmask").cpu() != pr["bmask"]).sum()))
# timing
gen.set_timing(False)
for _ in range(2): gen(inp.cuda(), uv.cuda())
torch.cuda.synchronize(); t = time.time()
for _ in range(5): gen(inp.cuda(), uv.cuda())
torch.cuda.synchronize(); dt = (time.time() - t) / 5
print("B=%d  %.2f ms/forward  %.1f img/s" % (B, dt * 1e3, B / dt))
gen.set_timing(True); gen(inp.cuda(), uv.cuda()); torch.cuda.synchronize(); print(gen.get_timing())
