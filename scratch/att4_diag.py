"""Dev (round 6): time compile-time variants of the split-precision attention kernel (scratch/libatt_<name>.so, built by scratch/build_att_variants.sh)
through bsr_debug_split_qkv / bsr_debug_attention_split at B = 32, 1024 tokens, round-robin after a warm-up (the clock drifts for seconds after the
chip wakes up); error against fp64 where the variant still computes the right thing.   python3 scratch/att4_diag.py [pv1]"""
import ctypes, glob, os, sys, time, torch
B, T = 32, 1024
pv1 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
torch.manual_seed(0)
x = (torch.randn(B, T, 384) * 0.5).cuda(); y = torch.zeros(B * T * 128 + 1024, device="cuda"); xs = torch.empty(B, T, 384, device="cuda")
xd = x[:2].double()
ref = torch.softmax(xd[..., :128] @ xd[..., 128:256].transpose(1, 2), dim=-1) @ xd[..., 256:]
libs = []
for so in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libatt_*.so"))):
    L = ctypes.CDLL(so)
    L.bsr_debug_split_qkv.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.bsr_debug_attention_split.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    libs.append((os.path.basename(so), L))
libs[0][1].bsr_debug_split_qkv(x.data_ptr(), xs.data_ptr(), B, T, None)
def run(L): L.bsr_debug_attention_split(xs.data_ptr(), y.data_ptr(), B, T, pv1, None)
t0 = time.time()
while time.time() - t0 < 3.0:
    for _, L in libs: run(L)
    torch.cuda.synchronize()
res = {}
for rnd in range(7):
    for name, L in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(L)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) * 50)
for name, L in libs:
    run(L); torch.cuda.synchronize()
    ts = sorted(res[name])
    print("%-36s median %.1f us (min %.1f max %.1f)  err vs fp64 %.2e" % (name, ts[len(ts) // 2], ts[0], ts[-1], float((y[:2 * T * 128].view(2, T, 128).double() - ref).abs().max())), flush=True)
    if "stamp" in name:
        st = y[B * T * 128:B * T * 128 + 192].cpu().view(torch.int32).view(6, 32).long()
        for i in range(8, 16):
            a = [int(st[k][i]) for k in range(6)]; nxt = int(st[0][i + 1])
            d = lambda x, y: (x - y) & 0xffffffff
            print("   iter %2d: phase A %5d  wait+barrier %5d  B gaps 0-7 %5d  8-15 %5d  16-23 %5d  tail %5d  total %5d cycles" % (i, d(a[1], a[0]), d(a[2], a[1]), d(a[5], a[2]), d(a[4], a[5]), d(a[3], a[4]), d(nxt, a[3]), d(nxt, a[0])))
