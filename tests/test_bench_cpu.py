"""bench.py's rank logic on a CPU box: `--gpus N` must start N ranks itself (the driver runs `python bench.py --gpus N`),
report n_gpus = N, and refuse to fall back to fewer ranks.  The generator is the labelled stub; nothing is measured."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_gpus2_launches_two_ranks_over_gloo():
    r = _run(["--gpus", "2", "--backend", "gloo", "--stub", "--batch", "2", "--steps", "2", "--warmup", "1", "--repeats", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["stub"] is True and j["steps"] == 2 and j["warmup"] == 1
    assert j["scaling"] == "weak" and j["higher_is_better"] is True and j["unit"] == "images/sec"
    assert j["config"]["global_batch"] == 4 and j["config"]["parallelism"] == "dp2"
    assert j["config"]["collective"].startswith("all_gather")
    ag = j["config"]["allgather"]
    assert ag["bytes_per_rank"] == 2 * 256 * 256 * 4 * 4 and ag["ms_alone"] > 0
    assert ag["verified"] is True and ag["backend"] == "gloo"          # every rank's shard of the gathered buffer checked against what that rank packed
    assert len(j["config"]["per_rank_images_per_sec"]) == 2
    assert abs(j["value"] - 4 * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 1e-3      # whole-job aggregate over both ranks
    assert j["repeats"]["n"] == 2 and j["repeats"]["ms_per_step_min"] > 0
    assert j["config"]["forwards_in_flight"] == 1 and "two_in_flight" not in j      # the CPU stand-in has no streams: the two-lane side measurement only exists on a GPU
    # `value` is always one forward at a time (BASELINE configs[1]: ONE batch of 32 resident), and every line says so
    assert j["value_mode"] == "one forward at a time" and j["single_stream_value"] == j["value"] and "roofline_in_flight" not in j


def test_gpus_without_devices_fails_loudly():
    """No silent fall-back to one GPU: on this CPU box `--gpus 2` on the real path must exit non-zero before starting ranks."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert "only 0 GPU" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_rejected():
    r = _run(["--gpus", "4", "--backend", "gloo", "--stub", "--batch", "1", "--steps", "1", "--warmup", "0"],
             env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE 1 != --gpus 4" in r.stderr


def test_stub_needs_gloo_and_gloo_needs_stub():
    assert _run(["--backend", "gloo"]).returncode != 0
    assert _run(["--stub"]).returncode != 0
    r = _run(["--gpus", "2", "--device", "0", "--stub", "--backend", "gloo"])          # --device runs the real generator
    assert r.returncode != 0 and "not with --stub" in r.stderr
    r = _run(["--gpus", "2", "--device", "0"])                                         # no GPU here: the launcher says so before starting ranks
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr


def test_roofline_tables_are_consistent():
    """The per-layer MMAC table bench.py prices launches with sums to SURVEY Appendix C's total and every layer sits in exactly one
    kernel group."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert abs(sum(bench.LAYER_MMAC.values()) - 9052.06) < 1.0
    grouped = [n for layers in bench.KERNEL_GROUPS.values() for n in layers]
    assert sorted(grouped) == sorted(list(bench.LAYER_MMAC) + list(bench.FUSED_LAUNCHES))      # every launch name sits in exactly one group
    for name, parts in bench.FUSED_LAUNCHES.items():                                            # a fused launch is priced as the sum of its layers
        assert abs(bench.launch_mmac(name) - sum(bench.LAYER_MMAC[p] for p in parts)) < 1e-9 and name in bench.LAYER_IO_ELEMS
    assert abs(2e-3 * sum(bench.LAYER_MMAC[n] for n in bench.LAYERS_3X3) - 2e-3 * (16.78 + 3.15) - bench.GFLOP_3X3_PER_IMAGE) < 0.01
    assert bench.physical_cores() >= 1
    # the same tables derived from the layer list (workload_tables) reproduce Appendix C, and re-pricing for configs[4]'s frames works in place
    m, io, ex = bench.workload_tables(False, 256)
    assert all(abs(m[k] - v) < 0.02 for k, v in bench.LAYER_MMAC.items()) and io == bench.LAYER_IO_ELEMS and set(m) == set(bench.LAYER_MMAC)
    bench.set_workload(True, 512)
    try:
        assert abs(bench.LAYER_MMAC["res0.attention"] - 2 * 4096 * 4096 * 128 / 1e6) < 1e-6          # quadratic in the 4096 tokens
        assert abs(bench.LAYER_MMAC["res3.conv1"] - 4096 * 877 * 128 / 1e6) < 1e-6 and abs(bench.LAYER_MMAC["up1"] - 4096 * 9 * 291 * 96 / 1e6) < 1e-6
        assert abs(bench.GFLOP_PER_IMAGE - 2e-3 * sum(bench.LAYER_MMAC.values())) < 1e-9 and 110 < bench.GFLOP_PER_IMAGE < 125
    finally:
        bench.set_workload(False, 256)
    assert abs(bench.GFLOP_PER_IMAGE - 18.104) < 1e-3 and abs(bench.GFLOP_3X3_PER_IMAGE - 11.017) < 1e-3


def test_under_torchrun_the_ranks_already_exist():
    """The driver's N > 1 form: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` — WORLD_SIZE is set, so
    bench.py must NOT start ranks of its own, and stdout must carry exactly one JSON line (rank 0's)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--backend", "gloo", "--stub", "--batch", "1", "--steps", "2", "--warmup", "1",
                        "--repeats", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["stub"] is True and j["config"]["global_batch"] == 2


def test_run_loop_refuses_to_run_without_a_gpu():
    """`python -m blindshadowremoval_amd.run_loop` (the torchrun entry of the data-parallel loops): no GPU -> exit code 2 with a message,
    never a CPU fallback; --device with several RCCL ranks is refused before anything else happens."""
    root = os.path.dirname(BENCH)
    env = dict(os.environ, PYTHONPATH=root)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "blindshadowremoval_amd.run_loop", "--loop", "ffhq", "--data", "x", "--checkpoint-dir", "/tmp/none"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 2 and "no CPU path" in r.stderr
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-m", "blindshadowremoval_amd.run_loop", "--loop", "ffhq", "--data", "x", "--checkpoint-dir", "/tmp/none", "--device", "0"],
                       capture_output=True, text=True, timeout=300, env=env2, cwd=root)
    assert r.returncode == 2 and "--backend gloo" in r.stderr


def test_kernel_group_traffic_from_the_committed_counter_passes(monkeypatch):
    """bench.attach_group_traffic: every kernel group gets its HBM rate from the committed counter passes (profiles/r6_pmc_traffic*.json)
    beside the algorithmic one — but only while the kernel sources hash to what the passes ran on (a stale figure is never quoted)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from blindshadowremoval_amd import build
    for dtype, sfx in (("f32", ""), ("f16", "_f16")):
        line = json.load(open(os.path.join(ROOT, "profiles", "r6_bench_n1%s.json" % sfx)))
        passes = json.load(open(os.path.join(ROOT, "profiles", "r6_pmc_traffic%s.json" % sfx)))
        rf = json.loads(json.dumps(line["roofline"]))
        for g in rf["kernel_groups"].values():
            g.pop("counter_GBps", None); g.pop("hbm_frac_counters", None); g.pop("traffic_ratio", None)
        monkeypatch.setattr(build, "source_sha16", lambda: "0" * 16)
        bench.attach_group_traffic(rf, 32, dtype)
        assert not any("counter_GBps" in g for g in rf["kernel_groups"].values())          # other sources: nothing quoted
        monkeypatch.setattr(build, "source_sha16", lambda: passes["kernel_src_sha16"])
        bench.attach_group_traffic(rf, 32, dtype)
        got = {k: g for k, g in rf["kernel_groups"].items() if "counter_GBps" in g}
        assert len(got) >= 8, sorted(rf["kernel_groups"])
        for k, g in got.items():
            assert 0.5 < g["traffic_ratio"] < 2.5 and 0 < g["hbm_frac_counters"] < 1, (k, g)
        bench.attach_group_traffic(rf, 16, dtype)                                          # another batch than the passes': untouched
