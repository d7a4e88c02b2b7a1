"""Dev: per-launch device time of one forward (HIP events around every launch: bsr_set_timing), mean of a few forwards after a warm-up.
python3 scratch/layer_times.py [dtype] [B]"""
import sys, time, torch
sys.path.insert(0, ".")
from blindshadowremoval_amd import Generator, init_weights
dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
gen = Generator(dtype=dtype).load_weights(init_weights(1))
torch.manual_seed(0)
inp, uv = torch.rand(B, 256, 256, 3).cuda(), torch.rand(B, 256, 256, 3).cuda()
def fwd():
    try:
        gen.check_range()          # diagnostic builds compute garbage: acknowledge the range guard's report so that the next forward still runs
    except RuntimeError:
        pass
    try:
        gen(inp, uv)
    except RuntimeError:
        pass
t0 = time.time()
while time.time() - t0 < 2.0:
    fwd()
torch.cuda.synchronize()
gen.set_timing(True)
acc, n = {}, 5
order = []
for _ in range(n):
    fwd(); torch.cuda.synchronize()
    for name, ms, _c in gen.get_launch_timing():
        if name not in acc: order.append(name)
        acc[name] = acc.get(name, 0.0) + ms / n
gen.set_timing(False)
tot = sum(acc.values())
print("dtype %s B %d: %d launches, %.4f ms" % (dtype, B, len(order), tot))
for name in order:
    print("  %-16s %8.1f us  %5.1f %%" % (name, acc[name] * 1e3, 100 * acc[name] / tot))
