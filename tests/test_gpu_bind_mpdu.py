"""sora_rx_bind_mpdu (round 6): a call's frame sink writes every MPDU straight into the caller's page-locked array, and sora_rx_deliver_async with that array copies rows
and count only.  The delivered table -- rows, and the MPDU bytes the rows point at -- equals the oracle's through every kernel chain; the binding is for one call; an
array that is too small is refused by the process call."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_util import batch, oracle_results, random_capture, same_results  # noqa: E402

pytestmark = pytest.mark.gpu


def delivered(buf):
    n = int(buf.nrows[0]); out = []
    for r in buf.rows[:n]:
        d = {k: int(r[k]) for k in ("capture_id", "start_sample", "end_sample", "error_code", "rate_kbps", "length", "nsym", "crc32", "cfo_est", "flags")}
        d["error_code"] &= 0xFFFFFFFF
        ok = d["error_code"] in (1, 0x80000006)
        d["mpdu"] = bytes(buf.mpdu[int(r["mpdu_offset"]):int(r["mpdu_offset"]) + d["length"]]) if ok else b""
        out.append(d)
    return out


def norm(rows):
    out = []
    for r in rows:
        d = {k: r[k] for k in ("capture_id", "start_sample", "end_sample", "rate_kbps", "length", "nsym", "crc32", "cfo_est", "mpdu")}
        d["error_code"] = r["error_code"] & 0xFFFFFFFF
        out.append(d)
    return out


@pytest.mark.parametrize("mhz", [20, 40])
def test_bound_delivery_equals_the_oracle_through_every_chain(oracle, mhz):
    import torch
    import sora_amd
    rng = np.random.default_rng(990 + mhz)
    small = [random_capture(oracle, rng, mhz, multipath_p=0.2) for _ in range(2)]
    many = [random_capture(oracle, rng, mhz, multipath_p=0.2) for _ in range(40)]
    for caps, chains in ((small, ((4, 0), (3, 1), (1, 64))), (many, ((1, 16), (1, 1), (3, 1), (0, 0)))):
        iq, descs = batch(caps)
        want = norm(oracle_results(oracle, caps, mhz))
        d = torch.from_numpy(iq).cuda()
        rx = sora_amd.Rx(len(caps), len(iq), sample_rate_mhz=mhz, max_frames_per_capture=4)
        rx.set_depth(2)
        t0 = rx.process_dev(d, descs)
        buf = sora_amd.HostResults(len(caps) * 4, rx.mpdu_bytes(t0)); other = sora_amd.HostResults(len(caps) * 4, rx.mpdu_bytes(t0))
        for front, trellis in chains:
            rx.set_front(front); rx.set_trellis(trellis); rx.flush()
            buf.mpdu[:] = 0xEE
            rx.bind_mpdu(buf)
            t = rx.process_dev(d, descs); rx.deliver_async(t, buf); rx.wait(t)
            got = [{k: g[k] for k in want[0]} for g in delivered(buf)] if want else delivered(buf)
            assert got == want, (front, trellis, rx.front(), rx.trellis())
            # the binding was for that call: the next one is delivered by the copy, into another array, with the same bytes where frames are
            other.mpdu[:] = 0x11
            t = rx.process_dev(d, descs); rx.deliver_async(t, other); rx.wait(t)
            got2 = [{k: g[k] for k in want[0]} for g in delivered(other)] if want else delivered(other)
            assert got2 == want
        # an array that is too small is refused, and the call after it runs unbound
        tiny = sora_amd.HostResults(4, 64)
        rx.bind_mpdu(tiny)
        with pytest.raises(sora_amd.SoraError):
            rx.process_dev(d, descs)
        t = rx.process_dev(d, descs); rx.deliver_async(t, other); rx.wait(t)
        assert ([{k: g[k] for k in want[0]} for g in delivered(other)] if want else delivered(other)) == want
        tiny.close(); buf.close(); other.close(); rx.close()
