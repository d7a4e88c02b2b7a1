"""Generate tests/golden/warp_sp_reference.npz with the REFERENCE's own scipy implementation of the offset warp
(`sp_batch_map_offsets` / `sp_batch_map_coordinates`, /root/reference/warp.py:61-68,118-131 — documented there as the
"reference implementation for tf_batch_map_offsets").  Build container only (imports /root/reference/warp.py with
tensorflow / cv2 stubbed; the scipy functions themselves are pure numpy/scipy).  Run: python tools/make_warp_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_sample_fixture import _install_stubs  # noqa: E402

_install_stubs()
sys.path.insert(0, "/root/reference")
import warp as ref_warp  # noqa: E402

rng = np.random.default_rng(165)
b, s = 3, 32
inp = rng.standard_normal((b, s, s)).astype(np.float32)
offsets = (rng.standard_normal((b, s, s, 2)) * 2.5).astype(np.float32)
offsets[0, :4] -= 6.0        # push some coordinates outside the map: exercises the clamp
offsets[1, :, -3:] += 7.0
offsets[2, 5, 5] = 0.0       # exact integer coordinates: floor == ceil
out = ref_warp.sp_batch_map_offsets(inp, offsets.reshape(b, -1, 2))
dst = os.path.join(ROOT, "tests", "golden", "warp_sp_reference.npz")
np.savez_compressed(dst, inp=inp, offsets=offsets, out=np.asarray(out, np.float32).reshape(b, s, s))
print(dst, out.shape, float(np.abs(out).max()))
