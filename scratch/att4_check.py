"""Dev (round 6): the one-wave-per-SIMD split-precision attention kernel (csrc/attention_h16.h) alone, through bsr_debug_split_qkv /
bsr_debug_attention_split: error against fp64 softmax(QK^T)V and the time per launch at B = 32, 1024 tokens (median of rounds after a warm-up),
beside the fp32 kernel.   python3 scratch/att4_check.py [pv1]"""
import sys, time, torch
sys.path.insert(0, ".")
from blindshadowremoval_amd import _lib
lib = _lib.load()
B, T = 32, 1024
torch.manual_seed(0)
x = (torch.randn(B, T, 384) * 0.5).cuda()
y = torch.empty(B, T, 128, device="cuda")
xs = torch.empty(B, T, 384, device="cuda")
xd = x[:2].double()
q, k, v = xd[..., :128], xd[..., 128:256], xd[..., 256:]
ref = torch.softmax(q @ k.transpose(1, 2), dim=-1) @ v
_lib.check(lib.bsr_debug_split_qkv(x.data_ptr(), xs.data_ptr(), B, T, None), "split")
for pv1 in (0, 1):
    _lib.check(lib.bsr_debug_attention_split(xs.data_ptr(), y.data_ptr(), B, T, pv1, None), "att")
    torch.cuda.synchronize()
    print("pv1=%d max err vs fp64 %.3e" % (pv1, float((y[:2].double() - ref).abs().max())), flush=True)
_lib.check(lib.bsr_debug_attention_dtype(x.data_ptr(), y.data_ptr(), B, T, 0, None), "f32")
torch.cuda.synchronize()
print("fp32 kernel max err vs fp64 %.3e" % float((y[:2].double() - ref).abs().max()), flush=True)
def run(kind):
    if kind == "f32": lib.bsr_debug_attention_dtype(x.data_ptr(), y.data_ptr(), B, T, 0, None)
    else: lib.bsr_debug_attention_split(xs.data_ptr(), y.data_ptr(), B, T, int(kind), None)
t0 = time.time()
while time.time() - t0 < 3.0:
    for kind in ("0", "1", "f32"): run(kind)
    torch.cuda.synchronize()
res = {}
for rnd in range(7):
    for kind in ("0", "1", "f32"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(kind)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(kind, []).append(e0.elapsed_time(e1) * 50)
for kind, ts in res.items():
    ts = sorted(ts)
    print("kernel %-4s median %.1f us (min %.1f max %.1f)" % (kind, ts[len(ts) // 2], ts[0], ts[-1]), flush=True)
