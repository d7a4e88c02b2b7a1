"""Dev: time attention kernel variants (scratch/libatt_<name>.so, built by scratch/build_att_variants.sh) through bsr_debug_attention_dtype
— fp32 kernel (dtype 0) and split-precision kernel (dtype 2) at B = 32, 1024 tokens — and check each against fp64 softmax(QK^T)V."""
import ctypes, sys, os, glob, torch
B, T = 32, 1024
torch.manual_seed(0)
x = (torch.randn(B, T, 384) * 0.5).cuda(); y = torch.empty(B, T, 128, device="cuda")
xd = x[:2].double()
q, k, v = xd[..., :128], xd[..., 128:256], xd[..., 256:]
ref = torch.softmax(q @ k.transpose(1, 2), dim=-1) @ v
for so in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libatt_*.so"))):
    lib = ctypes.CDLL(so)
    f = lib.bsr_debug_attention_dtype
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for dt in (0, 2):
        best = 1e9
        for rep in range(3):
            for _ in range(3): f(x.data_ptr(), y.data_ptr(), B, T, dt, None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f(x.data_ptr(), y.data_ptr(), B, T, dt, None)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 50)
        err = float((y[:2].double() - ref).abs().max())
        print("%-40s dtype %d %.1f us   max err vs fp64 %.2e" % (os.path.basename(so), dt, best, err), flush=True)
