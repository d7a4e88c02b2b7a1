"""Dev: time nonlocal_attention_x3_kernel variants built with -DAX3_DIAG_* (scratch/libatt_<name>.so) through bsr_debug_attention_dtype."""
import ctypes, sys, os, glob, torch
B, T = 32, 1024
x = (torch.randn(B, T, 384) * 0.5).cuda(); y = torch.empty(B, T, 128, device="cuda")
for so in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libatt_*.so"))):
    lib = ctypes.CDLL(so)
    f = lib.bsr_debug_attention_dtype
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for dt in (0, 2):
        for _ in range(3): f(x.data_ptr(), y.data_ptr(), B, T, dt, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f(x.data_ptr(), y.data_ptr(), B, T, dt, None)
        e1.record(); torch.cuda.synchronize()
        print("%-40s dtype %d %.1f us" % (os.path.basename(so), dt, e0.elapsed_time(e1) * 100))
