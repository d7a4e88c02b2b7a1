"""Dev: run ONE attention kernel variant N times (for rocprofv3 counter passes): python3 scratch/att_only.py <lib.so> <dtype> [reps]"""
import ctypes, sys, torch
so, dt = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B, T = 32, 1024
torch.manual_seed(0)
x = (torch.randn(B, T, 384) * 0.5).cuda(); y = torch.empty(B, T, 128, device="cuda")
f = ctypes.CDLL(so).bsr_debug_attention_dtype
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for _ in range(reps): f(x.data_ptr(), y.data_ptr(), B, T, dt, None)
torch.cuda.synchronize()
