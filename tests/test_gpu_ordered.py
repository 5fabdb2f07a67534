"""sora_rx_set_ordered (round 6): calls in flight complete in submission order -- a call's trellis kernel starts behind the previous call's -- and deliver what unordered calls deliver."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_util import make_capture, batch, same_results, oracle_results

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.fixture(scope="module")
def sora():
    import sora_amd
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    return sora_amd


def test_ordered_calls_complete_in_submission_order_with_the_oracles_rows(sora, torch_cuda, oracle):
    rates = (54000, 6000, 36000, 12000, 48000, 24000, 9000, 18000)
    caps = [make_capture(oracle, rates[i % 8], 120 + 37 * i, seed=900 + i, rate_mhz=20, sigma=40, tail=200)[0] for i in range(48)]
    iq, descs = batch(caps)
    d = torch_cuda.from_numpy(iq).cuda()
    want = oracle_results(oracle, caps, 20)
    per_call = 12
    for trellis in (0, 64, 16, 1):
        rx = sora.Rx(max_captures=per_call, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
        rx.set_depth(4); rx.set_trellis(trellis)
        assert rx.set_ordered(1) == 0 and rx.set_ordered() == 1
        for rep in range(3):                                                     # (a pipeline's second and third use: no recorded graph may get in the way)
            bufs = [sora.HostResults(per_call * 2, 1 << 20) for _ in range(4)]
            tickets = []
            for i in range(4):
                t = rx.process_dev(d, descs[i * per_call:(i + 1) * per_call]); rx.deliver_async(t, bufs[i]); tickets.append(t)
            order = [rx.wait_any() for _ in range(4)]
            assert order == tickets, (trellis, rep, order, tickets)              # the oldest finished call each time: with ordering, the submission order
            got = []
            for b in bufs:
                got += b.results(); b.close()
            by_cap = lambda rows: sorted(rows, key=lambda r: (r["capture_id"], r["start_sample"]))
            ok, why = same_results(by_cap(got), by_cap(want)); assert ok, (trellis, rep, why)
        assert rx.set_ordered(0) == 1
        t = rx.process_dev(d, descs[:per_call]); res = rx.results(ticket=t)       # ... and back
        ok, why = same_results(res, [r for r in want if r["capture_id"] < per_call]); assert ok, why
        rx.close()
