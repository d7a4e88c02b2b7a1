"""TSM variant (BASELINE config 5): the oracle's offset warp is pinned against the REFERENCE's own scipy implementation
(tests/golden/warp_sp_reference.npz, made by tools/make_warp_fixture.py from /root/reference/warp.py:61-68,118-131),
plus hand-computable cases for the ShareLayer (/root/reference/model_with_TSM.py:199-229)."""
import os

import numpy as np
import torch

from oracle import gsc_oracle as O


def test_warp_matches_reference_scipy_implementation(golden_dir):
    z = np.load(os.path.join(golden_dir, "warp_sp_reference.npz"))
    inp, off, want = torch.from_numpy(z["inp"]), torch.from_numpy(z["offsets"]), z["out"]
    b, s = inp.shape[0], inp.shape[1]
    ii, jj = torch.meshgrid(torch.arange(s), torch.arange(s), indexing="ij")
    grid = torch.stack([ii, jj], -1).float().reshape(1, -1, 2)
    got = O.batch_map_coordinates(inp[..., None], off.reshape(b, -1, 2) + grid).reshape(b, s, s)
    np.testing.assert_allclose(got.numpy(), want, atol=1e-5)      # scipy interpolates in float64


def test_map_offsets_resizes_and_scales_offsets():
    # offsets given at 4x the map size, constant (+0.25, -0.5) in image-fraction units -> shift by (+1, -2) cells on a 4x4 map
    x = torch.arange(16, dtype=torch.float32).reshape(1, 4, 4, 1)
    off = torch.zeros(1, 16, 16, 3)
    off[..., 0], off[..., 1], off[..., 2] = 0.25, -0.5, 9.0          # third channel is ignored (warp.py:139)
    y = O.batch_map_offsets(x, off)[0, :, :, 0]
    for i in range(4):
        for j in range(4):
            assert y[i, j] == x[0, min(i + 1, 3), max(j - 2, 0), 0]   # clamp-to-edge


def test_share_layer_group_max_mean_and_passthrough():
    torch.manual_seed(0)
    x = torch.rand(4, 8, 8, 3)
    reg = torch.zeros(4, 64, 64, 6)                                   # zero offsets: the warps are identities
    y = O.share_layer(x, reg, frame=2)
    assert y.shape == (4, 8, 8, 6)
    for g in range(2):
        mx = torch.maximum(x[2 * g], x[2 * g + 1])
        mean = (x[2 * g] + x[2 * g + 1]) / 2
        for f in range(2):
            np.testing.assert_allclose(y[2 * g + f, ..., :3].numpy(), mx.numpy(), atol=1e-7)
            np.testing.assert_allclose(y[2 * g + f, ..., 3:].numpy(), mean.numpy(), atol=1e-7)
    assert torch.equal(O.share_layer(x, reg, 2, share=False), torch.cat([x, x], 3))   # model_with_TSM.py:227
