"""Dev: print the per-step stamps of a -DBSR_AX3_STAMPS build of the role-split attention kernel (scratch/libatt_stamps.so)."""
import ctypes, os, torch
B, T = 32, 1024
torch.manual_seed(0)
x = (torch.randn(B, T, 384) * 0.5).cuda(); y = torch.zeros(B * T * 128 + 256, device="cuda")
f = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libatt_stamps.so")).bsr_debug_attention_dtype
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for _ in range(5): f(x.data_ptr(), y.data_ptr(), B, T, 2, None)
torch.cuda.synchronize()
st = y[B * T * 128:B * T * 128 + 128].cpu().view(torch.int64).view(2, 8, 4)
for role, name in ((0, "S-wave "), (1, "PV-wave")):
    print(name, "step: [start->after phase A] [phase A->before barrier] [barrier wait] | step total   (phase A = S MFMAs for S-wave, publish+fetch for PV-wave)")
    for t in range(8):
        a = st[role, t]
        nxt = st[role, t + 1, 0] if t + 1 < 8 else None
        print("   t=%2d  %6d %6d %6d | %s" % (t + 8, a[1] - a[0], a[2] - a[1], a[3] - a[2], (int(nxt - a[0]) if nxt is not None else "-")))
