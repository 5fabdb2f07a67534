"""The window-parallel trellis (sora_rx_set_trellis(SORA_TRELLIS_WINDOWED): k_viterbi16w + k_win_redo (the proof, and the serial decode of what fails it),
sora_amd/csrc/k_vitwin.hip) against the oracle and the compiled reference graph: the same rows and the same MPDU bytes as the serial kernels,
whether a frame's units all pass their verification (every decodable frame) or not (frames that are noise behind a good SIGNAL symbol: the
serial kernel decodes them again).  T11aViterbi: /root/reference/kernel/bb/Brick11/src/viterbi.hpp:103-237, viterbicore.h:293-555."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_util import batch, make_capture, oracle_results, pad_capture, same_results  # noqa: E402
from oracle.pyoracle import RATES  # noqa: E402

pytestmark = pytest.mark.gpu
WINDOWED = 1


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.fixture(scope="module")
def sora():
    import sora_amd
    sora_amd.load()
    assert sora_amd.device_count() > 0
    return sora_amd


def run_windowed(sora, torch, caps, mhz, max_frames=4, depth=None):
    iq, descs = batch(caps)
    rx = sora.Rx(max_captures=max(1, len(caps)), max_total_samples=max(64, len(iq)), sample_rate_mhz=mhz, max_frames_per_capture=max_frames)
    assert rx.set_trellis(WINDOWED) in (0, 16, 64) and rx.trellis() == WINDOWED
    if depth:
        rx.set_depth(depth)
    rx.process_dev(torch.from_numpy(iq).cuda(), descs)
    res = rx.results()
    st = rx.window_stats()
    rx.close()
    return res, st


def test_single_frames_every_rate_and_length(sora, torch_cuda, oracle):
    """One frame per call -- every unit is one window (the planner's finest cut) -- at every rate and at lengths around the window schedule's
    corners: no window at all (a single final trace-back), exactly one, the final trace-back swallowing a window, 2500 bytes."""
    n = 0
    for i, rate in enumerate(RATES):
        for j, ln in enumerate((1, 5, 29, 30, 31, 33, 34, 35, 61, 62, 63, 64, 65, 66, 67, 100, 257, 1024, 1500, 2304, 2500)):
            if (i + j) % 3:
                continue
            cap = make_capture(oracle, rate, ln, seed=5100 + 40 * i + j, rate_mhz=20, sigma=40 + 10 * (j % 7), tail=160)[0]
            got, st = run_windowed(sora, torch_cuda, [cap], 20)
            ok, why = same_results(got, oracle_results(oracle, [cap], 20))
            assert ok, (rate, ln, why)
            assert st["frames_decoded_again"] == 0, (rate, ln, st)                # decodable frames: every verification holds
            n += 1
    assert n > 40


def test_random_captures_equal_the_oracle(sora, torch_cuda, oracle):
    """Random captures (all rates, lengths, noise up to decode failure, CFO, DC, several frames per capture, truncation, pure noise) at
    20 and 40 MHz: batches of very different size, so units of one, three and more windows."""
    from gpu_util import random_capture
    rng = np.random.default_rng(20260926)
    for mhz, n in ((20, 200), (40, 120), (20, 9), (40, 1)):
        caps = [random_capture(oracle, rng, mhz, multipath_p=0.2) for _ in range(n)]
        got, st = run_windowed(sora, torch_cuda, caps, mhz, max_frames=8)
        ok, why = same_results(got, oracle_results(oracle, caps, mhz))
        assert ok, (mhz, n, why, st)


def test_noise_behind_a_good_header_is_decoded_again(sora, torch_cuda, oracle):
    """Frames whose SIGNAL symbol is intact and whose data field is noise: the units' vectors do not meet, k_win_redo decodes the frame
    again serially, and the bytes (FCS failure and all) are the oracle's.  The proof's record says so."""
    rng = np.random.default_rng(77)
    caps = []
    for i, rate in enumerate(RATES):
        cap = make_capture(oracle, rate, 1200 + 37 * i, seed=900 + i, rate_mhz=20, sigma=30, tail=160)[0].astype(np.int32)
        d0 = 320 + 80                                                              # preamble + SIGNAL @20 MHz
        cap[d0:len(cap) - 160] = np.rint(rng.normal(0, 2500, (len(cap) - 160 - d0, 2)))
        caps.append(np.clip(cap, -32768, 32767).astype(np.int16))
    for group in (caps, caps[:1], caps[5:6]):
        got, st = run_windowed(sora, torch_cuda, group, 20)
        want = oracle_results(oracle, group, 20)
        ok, why = same_results(got, want)
        assert ok, why
        assert st["boundaries_failed"] > 0 and st["frames_decoded_again"] > 0, st
        assert all(r["error_code"] != 1 for r in got)


def test_full_batch_equals_the_reference_graph(sora, torch_cuda, oracle):
    """The bench workload (BASELINE configs[2], 4096 x 1500 B at 54 Mbps) window-parallel, one call in flight and three: every capture
    against the compiled reference graph; no frame needs the serial kernel."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    from oracle.pyoracle import ReferenceGraph
    if not ReferenceGraph().available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present")
    nfr = bench.FRAMES_PER_GPU
    iq, descs, _ = bench.make_workload(oracle, nfr, seed0=0)
    rx = sora.Rx(max_captures=nfr, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
    rx.set_depth(3)
    rx.set_trellis(WINDOWED)
    d = torch_cuda.from_numpy(iq).cuda(); dd = sora.Rx.captures(descs)
    tickets = [rx.process_dev(d, dd) for _ in range(3)]
    kind, want = bench.reference_rows(iq, nfr, oracle)
    for t in tickets:
        ok, why = bench.check_against_reference(rx.results(ticket=t), kind, want, range(nfr))
        assert ok, why
    st = rx.window_stats()
    assert st["units"] >= 3 * 4 * nfr and st["boundaries_failed"] == 0 and st["frames_decoded_again"] == 0, st
    rx.close()


def test_fsample6_single_capture(sora, torch_cuda, oracle, golden_dir):
    """kernel/test-data/fsample-6 (6 Mbps, 1392 bytes, 465 symbols) as one capture: 44 one-window units; MPDU sha256 as the survey pinned it."""
    import hashlib
    iq = np.load(os.path.join(golden_dir, "fsample6_40mhz_i8.npz"))["iq_i8"].astype(np.int16) << 8
    cap = pad_capture(iq, 40)
    got, st = run_windowed(sora, torch_cuda, [cap], 40)
    assert len(got) == 1 and got[0]["error_code"] == 1 and got[0]["length"] == 1392
    assert hashlib.sha256(got[0]["mpdu"]).hexdigest() == "5a13a47743867e307040a009e1172b916c9015cd34fac586cafb2d0f1fd64b62"
    assert st["units"] == 44 and st["boundaries"] == 43 and st["boundaries_failed"] == 0, st
