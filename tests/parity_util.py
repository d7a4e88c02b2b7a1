"""Shared parity protocol for the -m gpu tests.

The generator thresholds a 32x32 map in the middle of the network (``bmask = d32 > 0.1``,
/root/reference/model.py:256).  A cell whose d32 sits within rounding distance of 0.1 may flip between
two correct fp32 implementations and changes the output by O(1) downstream, so parity is checked as:
  1. d32 (pre-threshold) agrees to ``tol``;
  2. every cell whose bmask differs has |d32_oracle - 0.1| < ``flip_tol`` (a legitimate flip);
  3. the four outputs agree to ``tol`` with the oracle run on the SAME mask.
"""

from oracle.gsc_oracle import GeneratorOracle

TOL = 1e-3          # north_star: outputs within 1e-3 per pixel, fp32
FLIP_TOL = 2e-5


def run_and_compare(gen, weights, inp, uv, tol=TOL, want_probes=(), flip_tol=FLIP_TOL):
    dev = "cuda:%d" % gen._device if gen._device is not None else "cuda"
    out = [t.cpu() for t in gen(inp.to(dev), uv.to(dev))]
    d32 = gen.probe("d32").cpu()
    bmask = gen.probe("bmask").cpu()
    oracle = GeneratorOracle(weights)
    pr = {}
    ref = oracle(inp, uv, probes=pr)
    assert float((d32 - pr["d32"]).abs().max()) <= tol
    flips = bmask != pr["bmask"]
    nflip = int(flips.sum())
    if nflip:
        assert float((pr["d32"][flips] - 0.1).abs().max()) < flip_tol, "bmask differs away from the threshold"
        pr = {}
        ref = oracle(inp, uv, probes=pr, bmask_override=bmask)
    errs = {}
    for a, b, name in zip(out, ref, ("gs", "con_rgb", "mask22", "dif")):
        assert a.shape == b.shape, name
        errs[name] = float((a - b).abs().max())
        assert errs[name] <= tol, "%s: max abs err %g > %g" % (name, errs[name], tol)
    for k in want_probes:
        a, b = gen.probe(k).cpu(), pr[k]
        assert float((a - b).abs().max()) <= tol, "probe %s" % k
    return out, ref, errs, nflip
