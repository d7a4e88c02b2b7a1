"""Dev: where the f16 mode's error enters — every probe of an f16 forward against the same probe of the fp32 forward (same weights / inputs):
max abs difference, and relative to the probe's own magnitude.   python scratch/f16_err.py [dtype]"""
import sys
sys.path.insert(0, ".")
import torch
from blindshadowremoval_amd import Generator, init_weights
dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
w = init_weights(1)
a, b = Generator().load_weights(w), Generator(dtype=dt).load_weights(w)
torch.manual_seed(0)
inp, uv = torch.rand(8, 256, 256, 3).cuda(), torch.rand(8, 256, 256, 3).cuda()
oa, ob = a(inp, uv), b(inp, uv)
names = ["x1", "x2", "x3", "res0", "res1", "res2", "up1", "up2", "y", "res3", "res4", "res5", "f"]
for n in names:
    try:
        pa, pb = a.probe(n).float(), b.probe(n).float()
    except RuntimeError as e:
        print(n, "n/a", str(e)[:60]); continue
    d = (pa - pb).abs()
    print("%-6s shape %-22s |x| max %8.3f rms %8.4f   err max %.3e rms %.3e   err rms / x rms %.2e" % (n, tuple(pa.shape), float(pa.abs().max()), float(pa.pow(2).mean().sqrt()), float(d.max()), float(d.pow(2).mean().sqrt()), float(d.pow(2).mean().sqrt() / pa.pow(2).mean().sqrt())))
for n, x, y in zip(("gs", "con_rgb", "mask22", "dif"), oa, ob):
    d = (x - y).abs()
    print("out %-8s err max %.3e rms %.3e   |x| max %.3f" % (n, float(d.max()), float(d.pow(2).mean().sqrt()), float(x.abs().max())))
