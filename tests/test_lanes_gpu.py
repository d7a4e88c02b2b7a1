"""lanes.concurrent_streams: the streams it returns must be distinct and — when it says so — really overlap."""
import time

import pytest
import torch

from blindshadowremoval_amd.lanes import _SPIN_CYCLES, concurrent_streams


@pytest.mark.gpu
def test_streams_seen_to_overlap_do_overlap():
    lanes, ok = concurrent_streams(0, 2)
    assert len(lanes) == 2 and lanes[0] != lanes[1]
    single, _ = concurrent_streams(0, 1)
    assert len(single) == 1

    def spin(a, b):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(a):
            torch.cuda._sleep(4 * _SPIN_CYCLES)
        with torch.cuda.stream(b):
            torch.cuda._sleep(4 * _SPIN_CYCLES)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    if not ok:
        pytest.skip("concurrent_streams found no pair of streams it had seen overlapping on this box: nothing to check")
    # overlapping spin kernels take ~0.5x, serialised ones 1.0x of one stream's time.  A WALL-CLOCK ratio on a shared box, measured up
    # to three times: clearly overlapping (< 0.85) passes; FULLY serialised in every attempt (>= 0.95: streams that were reported as
    # overlapping share a hardware queue — the regression this test exists for) FAILS; only the ambiguous band in between skips.
    seen = []
    for _ in range(3):
        serial = min(spin(lanes[0], lanes[0]) for _ in range(5))
        both = min(spin(lanes[0], lanes[1]) for _ in range(5))
        seen.append((round(both * 1e3, 3), round(serial * 1e3, 3)))
        if both < 0.85 * serial:
            return
    assert not all(b >= 0.95 * s_ for b, s_ in seen), \
        "streams reported as overlapping ran fully serialised in three attempts (both vs serial, ms): %s" % seen
    pytest.skip("ambiguous: the two streams showed neither a clear overlap nor serialisation in three attempts (both vs serial, ms): %s" % seen)
