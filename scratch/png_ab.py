"""Dev: time bsr_png_encode of two library builds (scratch/libpng_prev.so = three launches, scratch/libpng_new.so = one) on 16 strips, per call on an
idle stream and back to back; files compared byte for byte.   python scratch/png_ab.py"""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
s3 = (torch.rand(16, 256, 768, 3, device="cuda") * 255).to(torch.uint8)
s7 = (torch.rand(16, 256, 1792, 3, device="cuda") * 255).to(torch.uint8)
outs = {}
for name in ("prev", "new"):
    L = ctypes.CDLL(os.path.join(here, "libpng_%s.so" % name))
    L.bsr_png_file_bytes.restype = ctypes.c_size_t
    L.bsr_png_scratch_bytes.restype = ctypes.c_size_t
    L.bsr_png_encode.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    for tag, strip in (("768", s3), ("1792", s7)):
        b, h, w, _ = strip.shape
        nbytes = L.bsr_png_file_bytes(h, w)
        out = torch.zeros(b, nbytes, dtype=torch.uint8, device="cuda")
        scratch = torch.zeros(max(L.bsr_png_scratch_bytes(b), 4096) // 8 + 1, dtype=torch.int64, device="cuda")
        run = lambda: L.bsr_png_encode(0, strip.data_ptr(), b, h, w, out.data_ptr(), nbytes, scratch.data_ptr(), None)
        for _ in range(5): run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            a.record(); run(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
        ts.sort()
        a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100): run()
        e.record(); torch.cuda.synchronize()
        outs[(name, tag)] = out.cpu()
        print("%-5s 16 x %4s: per call on an idle stream %.4f ms, back to back %.4f ms" % (name, tag, ts[len(ts) // 2], a.elapsed_time(e) / 100), flush=True)
for tag in ("768", "1792"):
    print("files identical (%s):" % tag, bool(torch.equal(outs[("prev", tag)], outs[("new", tag)])))
