"""Dev: event-timed PNG encoders on 16 FFHQ strips (3 x 256 wide) and 16 UCB strips (7 x 256): python scratch/png_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blindshadowremoval_amd.gpu_png import StripEncoder
enc = StripEncoder(0)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
rows = torch.rand(16, 256, 256, 16, device="cuda"); pk = torch.rand(16, 256, 256, 4, device="cuda")
figs = [rows[..., 0:3], pk[..., 0:3], (pk[..., 3:4], rows[..., 15:16], 2.0)]
s3 = (torch.rand(16, 256, 768, 3, device="cuda") * 255).to(torch.uint8)
s7 = (torch.rand(16, 256, 1792, 3, device="cuda") * 255).to(torch.uint8)
print("encode_figs 16 x (3 x 256): %.3f ms;  encode u8 16 x 768: %.3f ms;  encode u8 16 x 1792: %.3f ms" % (t(lambda: enc.encode_figs(figs)), t(lambda: enc.encode(s3)), t(lambda: enc.encode(s7))))
def tb(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("back to back (50 calls between two events): encode_figs %.3f ms;  u8 16 x 768 %.3f ms;  u8 16 x 1792 %.3f ms" % (tb(lambda: enc.encode_figs(figs)), tb(lambda: enc.encode(s3)), tb(lambda: enc.encode(s7))))
