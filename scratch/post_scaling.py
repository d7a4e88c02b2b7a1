"""Probe (not product code): how the UCB post-processing job scales over processes on this box — k copies of the same job stream
running concurrently, items/s for each k.  python scratch/post_scaling.py [k ...]   (env BSR_PNG_WRITER=pil: PIL's encoder)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.chdir(ROOT)
    import tempfile
    import torch; torch.set_num_threads(1)
    import test_ucb_post as T
    from blindshadowremoval_amd import ucb_post
    from blindshadowremoval_amd.fsrnet import FSRNet, Config
    key, row, box, masks, con, dif = next(iter(T.cases()))
    cfg = Config(0); cfg.UCB_MASK_ROOT = os.path.join(ROOT, "tests/golden/UCB_masks")
    mf = FSRNet._ucb_masks(type("X", (), {"config": cfg})())[0]
    out = tempfile.mkdtemp()
    job = dict(im=row[..., 0:3], gt=row[..., 3:6], con=con, mp=dif, box=box, masks=mf, png=os.path.join(out, "a.png"), return_figs=False)
    ucb_post.run_post_job(job)
    print("ready", flush=True)
    sys.stdin.readline()
    n = int(sys.argv[2])
    t0, c0 = time.perf_counter(), time.process_time()
    for _ in range(n):
        ucb_post.run_post_job(job)
    print(time.perf_counter() - t0, time.process_time() - c0, flush=True)
    sys.exit(0)
n = 40
env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="")
for k in [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 24, 32]:
    ps = [subprocess.Popen([sys.executable, __file__, "--child", str(n)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env, text=True) for _ in range(k)]
    for p in ps:
        assert p.stdout.readline().strip() == "ready"
    t0 = time.perf_counter()
    for p in ps:
        p.stdin.write("go\n"); p.stdin.flush()
    res = [tuple(float(x) for x in p.stdout.readline().split()) for p in ps]
    dt = time.perf_counter() - t0
    for p in ps:
        p.wait()
    print("k=%d: %.1f items/s, wall per item %.1f ms, cpu per item %.1f ms" % (k, k * n / dt, sum(r[0] for r in res) / k / n * 1e3, sum(r[1] for r in res) / k / n * 1e3), flush=True)
