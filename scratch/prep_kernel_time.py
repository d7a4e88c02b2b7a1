"""Dev: event-timed bsr_prep_rows on 16 rows (the FFHQ sample and 15 UCB items), python scratch/prep_kernel_time.py"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blindshadowremoval_amd import prep, dataset as D
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
parts = [prep.host_part((os.path.join(G, "sample_imgs", "02165", "02165.npy"), None, 256))]
for lm in sorted(glob.glob(os.path.join(G, "UCB", "train", "input", "*", "*.npy")), key=D.natural_key)[:15]:
    pg = lm.replace("\\", "/").split("/")
    gt = os.path.splitext("/".join(pg[:-3] + ["gt"] + pg[-2:]))[0] + ".png"
    parts.append(prep.host_part((lm, gt, 256)))
dp = prep.DevicePrep(0, 256)
for _ in range(3): out, _ = dp.rows(parts)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); out, _ = dp.rows(parts); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
print("dp.rows 16 rows: min %.3f ms, median %.3f ms (copy + prep_rows_kernel + blur)" % (ts[0], ts[len(ts) // 2]), "checksum", float(out.nan_to_num().double().sum()))
