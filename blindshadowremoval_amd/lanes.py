"""Streams for keeping several forwards in flight on one GPU (bench.py --streams 2, a serving loop with two handles).

Every launch of the generator's 1/8-resolution trunk is ONE round of workgroups: alone on the chip, its tail and the next launch's
ramp leave most CUs idle.  A second, independent forward (its own handle = its own workspace) issued on another HIP stream fills
them: +4 % images/s at fp32, +8 % at f32x3 (scratch/two_stream.py).  That only happens when the two streams sit on DIFFERENT
hardware queues: the HIP runtime multiplexes all streams of a process over a few queues (4 by default), packets of one queue run
in order, and `torch.cuda.Stream()` hands out pool streams of which some pairs share a queue (measured: about one pair in four
serialises completely — scratch/stream_pairs.py).  `concurrent_streams` therefore TESTS candidates with a spin kernel instead of
trusting the pool.
"""
import time
from typing import List, Tuple

import torch

_SPIN_CYCLES = 4_000_000        # ~2 ms on MI355X: long against launch overhead, short enough for a handful of probes


def _spin_pair_ms(a, b) -> float:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(_SPIN_CYCLES)
    with torch.cuda.stream(b):
        torch.cuda._sleep(_SPIN_CYCLES)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def concurrent_streams(device: int = 0, n: int = 2, candidates: int = 8) -> Tuple[List["torch.cuda.Stream"], bool]:
    """-> (n streams on cuda:`device`, verified).  verified = every pair of them was SEEN to overlap (two spin kernels, one per
    stream, finish in clearly less than twice the time of two on one stream).  When no such set exists among `candidates` fresh
    streams the first n are returned with verified = False: correct, merely not concurrent."""
    if n < 1:
        raise ValueError("n must be >= 1")
    with torch.cuda.device(device):
        pool = [torch.cuda.Stream(device=device) for _ in range(max(candidates, n))]
        if n == 1:
            return pool[:1], True
        if not hasattr(torch.cuda, "_sleep"):                     # the spin kernel is a private torch helper: without it nothing can be verified
            return pool[:n], False
        torch.cuda._sleep(1000)                                   # load the spin kernel before anything is timed
        serial = min(_spin_pair_ms(pool[0], pool[0]) for _ in range(2))
        chosen = [pool[0]]
        for s in pool[1:]:
            if all(min(_spin_pair_ms(s, c) for _ in range(2)) < 0.7 * serial for c in chosen):
                chosen.append(s)
                if len(chosen) == n:
                    return chosen, True
        return pool[:n], False
