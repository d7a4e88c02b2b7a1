"""Dev: time attention kernel variants (scratch/libatt_<name>.so, built by scratch/build_att_variants.sh) through bsr_debug_attention_dtype
— fp32 kernel (dtype 0) and split-precision kernel (dtype 2) at B = 32, 1024 tokens — and check each against fp64 softmax(QK^T)V.
The chip's clock drifts for seconds after it wakes up, so the variants are timed ROUND-ROBIN after a warm-up and the median is reported."""
import ctypes, sys, os, glob, time, torch
B, T = 32, 1024
torch.manual_seed(0)
x = (torch.randn(B, T, 384) * 0.5).cuda(); y = torch.empty(B, T, 128, device="cuda")
xd = x[:2].double()
q, k, v = xd[..., :128], xd[..., 128:256], xd[..., 256:]
ref = torch.softmax(q @ k.transpose(1, 2), dim=-1) @ v
libs = []
for so in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libatt_*.so"))):
    f = ctypes.CDLL(so).bsr_debug_attention_dtype
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    libs.append((os.path.basename(so), f))
dts = [int(a) for a in sys.argv[1:]] or [0, 2]
t0 = time.time()
while time.time() - t0 < 3.0:                       # warm-up: clocks, code objects
    for _, f in libs:
        for dt in dts: f(x.data_ptr(), y.data_ptr(), B, T, dt, None)
    torch.cuda.synchronize()
res = {}
for rnd in range(7):
    for name, f in libs:
        for dt in dts:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f(x.data_ptr(), y.data_ptr(), B, T, dt, None)
            e1.record(); torch.cuda.synchronize()
            res.setdefault((name, dt), []).append(e0.elapsed_time(e1) * 50)
for name, f in libs:
    for dt in dts:
        f(x.data_ptr(), y.data_ptr(), B, T, dt, None); torch.cuda.synchronize()
        err = float((y[:2].double() - ref).abs().max())
        ts = sorted(res[(name, dt)])
        print("%-40s dtype %d  median %.1f us  (min %.1f max %.1f)   max err vs fp64 %.2e" % (name, dt, ts[len(ts) // 2], ts[0], ts[-1], err), flush=True)
