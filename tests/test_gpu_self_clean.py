"""The self-cleaning counter protocol of the product build: a pipeline's calls alternate between two sets of job counters, and every call's k_scan clears
the set the NEXT call will use (and k_pipe's hand-off words, and its own symbol-slot owners) instead of a fill kernel in front of every call.  The tools variant of the
library does not take this path, so the transitions are exercised here on the shipped one: on ONE pipeline, calls with different capture sets (more, fewer, none),
hipGraph replay switched on and off in between (a recorded graph carries its own fill), the symbol chain and the trellis kernel flipped, an empty call, a reset --
every call's rows and MPDU bytes equal the oracle's."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_util import batch, oracle_results, random_capture, same_results  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mhz", [20, 40])
def test_one_pipeline_through_every_transition(oracle, mhz):
    import torch
    import sora_amd
    rng = np.random.default_rng(6100 + mhz)
    sets = []
    for n in (5, 2, 7, 1, 3):
        caps = [random_capture(oracle, rng, mhz, multipath_p=0.2) for _ in range(n)]
        iq, descs = batch(caps)
        sets.append((torch.from_numpy(iq).cuda(), descs, oracle_results(oracle, caps, mhz)))
    cap_samples = max(int(d.shape[0]) for d, _, _ in sets)
    rx = sora_amd.Rx(max_captures=7, max_total_samples=cap_samples, sample_rate_mhz=mhz, max_frames_per_capture=4)
    rx.set_depth(1)                                                         # one pipeline: every call follows the one before it on the same counters
    calls = 0

    def run(k, times=1):
        nonlocal calls
        d, descs, want = sets[k]
        for _ in range(times):
            got = rx.results(ticket=rx.process_dev(d, descs))
            ok, why = same_results(got, want)
            assert ok, (calls, k, rx.front(), rx.trellis(), why)
            calls += 1
    # plain calls, the set changing every time
    for k in (0, 1, 2, 3, 4, 0):
        run(k)
    # an empty call between two others
    assert rx.results(ticket=rx.process_dev(sets[0][0], [])) == []
    run(2)
    # graph replay: the second identical call records, the third replays; then a different set (the graph is dropped), then off again
    rx.set_graph(1)
    run(1, 4); run(4, 3); run(1, 2)
    rx.set_graph(0)
    run(0); run(3)
    # every symbol chain and trellis kernel the handle can run, flipped between calls (and k_pipe when it fits: its hand-off words are part of the protocol)
    for front in (1, 3, 4, 0):
        for trellis in (64, 16, sora_amd.TRELLIS_WINDOWED, 0):
            rx.set_front(front); rx.set_trellis(trellis)
            run((front + trellis) % 5); run((front + trellis + 1) % 5)
    # graph on while the kernels flip
    rx.set_graph(1)
    for front in (3, 1, 4):
        rx.set_front(front)
        run(2, 3)
    rx.set_graph(0); rx.set_front(0); rx.set_trellis(0)
    rx.reset()
    run(0); run(1)
    assert calls > 60
    rx.close()
