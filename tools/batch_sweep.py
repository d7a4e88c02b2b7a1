#!/usr/bin/env python3
"""Forward-only batch sweep of the GSC generator on one MI355X: images/s, ms per forward and launches per forward at
B in {1, 2, 4, 8, 10, 16, 32} (fp32, one forward at a time on one stream; B = 10 is the reference's literal element,
/root/reference/train_test_GSC.py:866-871, B = 16 BASELINE configs[2]).  Writes one JSON object.

    python tools/batch_sweep.py [--out gpurun_out/r4_batch_sweep.json] [--dtype f32] [--batches 1,2,4,8,10,16,32] [--seconds 0.5]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _groups(lt):
    """per-launch events folded by layer kind (res3.conv2 -> res*.conv2)"""
    out = {}
    for name, ms, _cls in lt:
        key = ("res*." + name.split(".", 1)[1]) if name.startswith("res") and "." in name else name
        out[key] = out.get(key, 0.0) + ms
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--batches", default="1,2,4,8,10,16,32")
    ap.add_argument("--seconds", type=float, default=0.5, help="timed region per point")
    args = ap.parse_args()
    import torch
    from blindshadowremoval_amd import Generator, init_weights
    from blindshadowremoval_amd.build import source_sha16
    w = init_weights(1)
    gen = Generator(device=0, dtype=args.dtype).load_weights(w)
    rows = []
    for B in [int(b) for b in args.batches.split(",")]:
        g = torch.Generator(device="cpu").manual_seed(1234)
        inp = torch.rand(B, 256, 256, 3, generator=g).cuda()
        uv = torch.rand(B, 256, 256, 3, generator=g).cuda()
        out = tuple(torch.empty((B, 256, 256, c), device="cuda") for c in (1, 3, 3, 1))
        for _ in range(5):
            gen(inp, uv, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gen(inp, uv, out=out)
        torch.cuda.synchronize()
        one = time.perf_counter() - t0
        n = max(5, int(args.seconds / max(one, 1e-4)))
        best = None
        for _rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                gen(inp, uv, out=out)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            best = dt if best is None else min(best, dt)
        gen.set_timing(True)
        gen(inp, uv, out=out)
        torch.cuda.synchronize()
        lt = gen.get_launch_timing()
        gen.set_timing(False)
        dev_ms = sum(ms for _n, ms, _c in lt)
        rows.append({"batch": B, "images_per_sec": round(B / best, 1), "ms_per_forward": round(best * 1e3, 4), "launches_per_forward": len(lt),
                     "device_ms_sum_of_launches": round(dev_ms, 4), "forwards_timed": n,
                     "layer_us": {k: round(v * 1e3, 1) for k, v in _groups(lt).items()}})
        print({k: v for k, v in rows[-1].items() if k != 'layer_us'}, file=sys.stderr)
    full = rows[-1]["images_per_sec"]
    for r in rows:
        r["rate_vs_largest_batch"] = round(r["images_per_sec"] / full, 4)
    res = {"what": "forward-only batch sweep, one forward at a time on one stream, inputs resident in HBM, best of 3 timed regions",
           "dtype": args.dtype, "height": 256, "width": 256, "kernel_src_sha16": source_sha16(), "rows": rows}
    line = json.dumps(res, indent=1)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(line + "\n")
    print(line)


if __name__ == "__main__":
    main()
