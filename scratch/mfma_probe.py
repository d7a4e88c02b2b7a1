import ctypes, torch, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprobe.so"))
lib.probe_run.argtypes = [ctypes.c_void_p]*3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
torch.manual_seed(0)
for which in (32, 16):
    K = 64
    A = torch.randn(which, K, device="cuda"); B = torch.randn(K, which, device="cuda")
    C = torch.zeros(which, which, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    rc = lib.probe_run(A.data_ptr(), B.data_ptr(), C.data_ptr(), K, which, s)
    torch.cuda.synchronize()
    ref = (A.double() @ B.double()).float()
    print(which, "rc", rc, "maxerr", (C - ref).abs().max().item())
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)
import subprocess; print(subprocess.run("lscpu | head -20; nproc; free -g | head -2", shell=True, capture_output=True, text=True).stdout)
