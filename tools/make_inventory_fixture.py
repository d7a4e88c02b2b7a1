"""Generate tests/golden/gsc_ckpt94_inventory.json from the reference's own checkpoint index.

Run in the build container (needs /root/reference):  python tools/make_inventory_fixture.py
The fixture is DATA (variable names and shapes of ``generator/*`` in
/root/reference/log/<GSC run>/ckpt-94.index); it pins the weight layout the oracle and the HIP
packer chain through.  The TSM and RGB indices are recorded too.
"""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blindshadowremoval_amd.tf_bundle import generator_inventory  # noqa: E402

REF = "/root/reference/log"
out = {}
for idx in sorted(glob.glob(os.path.join(REF, "*", "ckpt-*.index"))):
    run = os.path.basename(os.path.dirname(idx))
    tag = "tsm" if run.endswith("with-TSM") else ("rgb" if run.endswith("RGB-model") else "gsc")
    inv = generator_inventory(idx)
    out[tag] = {"index": os.path.basename(idx), "run": run,
                "n_variables": len(inv), "n_params": int(sum(__import__("numpy").prod(s) for s in inv.values())),
                "variables": {k: list(v) for k, v in sorted(inv.items())}}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gsc_ckpt94_inventory.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
print({k: (v["n_variables"], v["n_params"]) for k, v in out.items()})
