"""k_pipe (sora_rx_set_front(4), sora_amd/csrc/k_rx.hip): the symbol chain AND the window-parallel trellis of a handful of frames as ONE launch whose workgroups
hand symbols on inside it -- front workgroups -> a frame's tracker (chain in wave 0, soft values made by waves 1-3 behind it) -> trellis waves that wait for the soft
values they read.  Same rows and MPDU bytes as the oracle: random captures at both sample rates, frames of one to 835 symbols at every rate (the records' LDS ring wraps
from 512 on), several frames per capture, frames that are noise behind an intact SIGNAL symbol (the units' proof fails: k_win_redo decodes them again), repeated and
graph-replayed calls (every hand-off word is cleared by the call), several calls in flight; no frame is ever reported as SORA_E_INTERNAL_TIMEOUT."""
import hashlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_util import batch, make_capture, oracle_results, pad_capture, random_capture, same_results  # noqa: E402
from oracle.pyoracle import RATES  # noqa: E402

pytestmark = pytest.mark.gpu
E_INTERNAL_TIMEOUT = 0x8000F001


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


def run_pipe(sora, torch, caps, mhz, max_frames=4, depth=1):
    iq, descs = batch(caps)
    rx = sora.Rx(max_captures=max(1, len(caps)), max_total_samples=max(64, len(iq)), sample_rate_mhz=mhz, max_frames_per_capture=max_frames)
    rx.set_depth(depth)
    rx.set_front(4)
    assert rx.front() == 4 and rx.trellis() == sora.TRELLIS_WINDOWED, (rx.front(), rx.trellis())
    rx.process_dev(torch.from_numpy(iq).cuda(), descs)
    res = rx.results()
    rx.close()
    assert all(r["error_code"] != E_INTERNAL_TIMEOUT for r in res)
    return res


@pytest.mark.parametrize("depth", [1, 4])
def test_random_captures_equal_the_oracle(sora, torch_cuda, oracle, depth):
    """(depth 1: so few workgroups that the trellis role runs in its 64-lane form, two units per wave; depth 4: the sixteen-lane form, eight per wave)"""
    rng = np.random.default_rng(20260928 + depth)
    for mhz, n, reps in ((20, 1, 12), (40, 1, 12), (20, 3 if depth == 1 else 1, 6), (40, 2 if depth == 1 else 1, 6)):
        for _ in range(reps):
            caps = [random_capture(oracle, rng, mhz, multipath_p=0.2) for _ in range(n)]
            ok, why = same_results(run_pipe(sora, torch_cuda, caps, mhz, depth=depth), oracle_results(oracle, caps, mhz))
            assert ok, (mhz, n, why)


def test_every_rate_one_symbol_to_835(sora, torch_cuda, oracle):
    k = 0
    for i, rate in enumerate(RATES):
        for j, ln in enumerate((1, 7, 24, 47, 100, 511, 1500, 2500)):
            if (i + j) % 2:
                continue
            cap = make_capture(oracle, rate, ln, seed=9000 + 16 * i + j, rate_mhz=20, sigma=(30, 200, 700, 1500)[(i + j) % 4], tail=160, cfo_hz=(-70e3, 0, 35e3)[j % 3])[0]
            ok, why = same_results(run_pipe(sora, torch_cuda, [cap], 20, max_frames=2, depth=1 + 3 * (k % 2)), oracle_results(oracle, [cap], 20))
            assert ok, (rate, ln, why)
            k += 1
    assert k >= 30


def test_frames_of_noise_are_decoded_again(sora, torch_cuda, oracle):
    rng = np.random.default_rng(5)
    caps = []
    for i in range(4):
        c = make_capture(oracle, (54000, 6000, 24000, 36000)[i], 1500, seed=100 + i, rate_mhz=20, sigma=50, tail=160)[0].astype(np.int32)
        c[700:len(c) - 200] = np.rint(rng.normal(0, 2500, (len(c) - 900, 2)))   # the data field replaced by noise: SIGNAL stays intact
        caps.append(np.clip(c, -32768, 32767).astype(np.int16))
    iq, descs = batch(caps)
    rx = sora.Rx(max_captures=4, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
    rx.set_depth(1); rx.set_front(4)
    assert rx.front() == 4
    rx.process_dev(torch_cuda.from_numpy(iq).cuda(), descs)
    got = rx.results()
    st = rx.window_stats()
    rx.close()
    ok, why = same_results(got, oracle_results(oracle, caps, 20))
    assert ok, why
    assert st["frames_decoded_again"] >= 1 and st["boundaries_failed"] >= 1, st


def test_repeated_calls_graph_replay_and_calls_in_flight(sora, torch_cuda, oracle, golden_dir):
    iq = np.load(os.path.join(golden_dir, "fsample6_40mhz_i8.npz"))["iq_i8"].astype(np.int16) << 8
    cap = pad_capture(iq, 40)
    want = "5a13a47743867e307040a009e1172b916c9015cd34fac586cafb2d0f1fd64b62"
    d = torch_cuda.from_numpy(cap).cuda()
    one = [(0, len(cap), 0)]
    rx = sora.Rx(max_captures=1, max_total_samples=len(cap), sample_rate_mhz=40, max_frames_per_capture=2)
    assert rx.front() == 4 and rx.trellis() == sora.TRELLIS_WINDOWED             # a single capture, default depth: the automatic choice
    for graph in (0, 1):
        rx.set_graph(graph)
        for _ in range(6):
            res = rx.results(ticket=rx.process_dev(d, one))
            assert len(res) == 1 and res[0]["error_code"] == 1 and hashlib.sha256(res[0]["mpdu"]).hexdigest() == want
    rx.set_graph(0)
    tickets = [rx.process_dev(d, one) for _ in range(8)]                         # eight calls in flight, every one k_pipe on its own stream
    for t in tickets:
        res = rx.results(ticket=t)
        assert len(res) == 1 and res[0]["error_code"] == 1 and hashlib.sha256(res[0]["mpdu"]).hexdigest() == want
    st = rx.window_stats()
    assert st["boundaries_failed"] == 0 and st["frames_decoded_again"] == 0, st
    rx.close()
    # what does not fit runs as the three-kernel chain
    big = sora.Rx(max_captures=256, max_total_samples=1 << 22, sample_rate_mhz=20, max_frames_per_capture=2)
    big.set_depth(1); big.set_front(4)
    assert big.front() == 3
    big.close()


def test_edges_silent_tiny_truncated_overlong_and_the_row_limit(sora, torch_cuda, oracle):
    """What test_gpu_parity's edge cases hold, explicitly through k_pipe: a capture of four frames of four rates, silence, a capture shorter than a symbol, a frame cut
    by the capture's end, a frame of the longest length the header admits -- and a row limit below the number of frames found (the last row carries the flag)."""
    from gpu_util import awgn
    rng = np.random.default_rng(11)
    parts = []
    for i, rate in enumerate((54000, 6000, 36000, 48000)):
        parts.append(oracle.tx_capture(rng.integers(0, 256, 200 + 100 * i).astype(np.uint8).tobytes(), rate, lead=0, tail=400 + 52 * i))
    multi = pad_capture(awgn(np.concatenate(parts), 120, 3), 40)
    silent = np.zeros((2800, 2), np.int16)
    tiny = np.zeros((28, 2), np.int16)
    trunc = pad_capture(oracle.tx_capture(bytes(1000), 12000)[:9000], 40)
    overmtu = pad_capture(oracle.tx_capture(bytes(2497), 54000), 40)
    for caps, mf in (([multi, silent], 8), ([tiny, trunc, overmtu], 4), ([multi], 16)):
        ok, why = same_results(run_pipe(sora, torch_cuda, caps, 40, max_frames=mf), oracle_results(oracle, caps, 40))
        assert ok, why
    want = oracle_results(oracle, [multi], 40)
    assert len(want) == 4
    for mf in (1, 2, 3):
        got = run_pipe(sora, torch_cuda, [multi], 40, max_frames=mf)
        ok, why = same_results(got, want[:mf]); assert ok, (mf, why)
        assert [r["flags"] for r in got] == [0] * (mf - 1) + [1], (mf, [r["flags"] for r in got])


def test_the_automatic_choice_leaves_k_pipe_alone_while_a_batch_handle_keeps_the_chip_full(sora, torch_cuda, oracle, golden_dir):
    """k_pipe's workgroups take a whole CU's LDS each: beside a full chip the three-kernel chain is the faster one (tools/pipe_under_load.py), so a handle's automatic
    choice looks at the process's other handles on the device -- a batch-sized one that has just taken a call switches it to 3, and back a few milliseconds later."""
    import time
    iq = np.load(os.path.join(golden_dir, "fsample6_40mhz_i8.npz"))["iq_i8"].astype(np.int16) << 8
    cap = pad_capture(iq, 40)
    small = sora.Rx(max_captures=1, max_total_samples=len(cap), sample_rate_mhz=40, max_frames_per_capture=2)
    assert small.front() == 4
    caps = [make_capture(oracle, 54000, 300, seed=i, rate_mhz=20, sigma=50, tail=160)[0] for i in range(4)]
    big_iq, descs = batch(caps * 64)
    big = sora.Rx(max_captures=256, max_total_samples=len(big_iq), sample_rate_mhz=20, max_frames_per_capture=2)     # 8 x 256 x 2 rows in flight: a batch handle
    d_big = torch_cuda.from_numpy(big_iq).cuda()
    assert small.front() == 4                                                      # (a handle that has not taken a call does not count)
    # (no sleeps: the window is set, not waited for -- an hour while the batch handle counts as busy, zero for "that was long ago")
    old_window = sora.set_share_window_us(4000000000)
    try:
        big.wait(big.process_dev(d_big, descs))
        assert small.front() == 3
        t = small.process_dev(torch_cuda.from_numpy(cap).cuda(), [(0, len(cap), 0)])
        assert small.call_front(t) == 3 and small.call_front() == 3                # what the call was launched with is latched with its ticket
        res = small.results(ticket=t)                                              # ... and decodes through the three kernels
        assert len(res) == 1 and res[0]["error_code"] == 1
        small.set_front(4); assert small.front() == 4                              # an explicit request is an explicit request
        small.set_front(0)
        sora.set_share_window_us(0)
        assert small.front() == 4
        t2 = small.process_dev(torch_cuda.from_numpy(cap).cuda(), [(0, len(cap), 0)])
        assert small.call_front(t2) == 4
        res = small.results(ticket=t2); assert len(res) == 1 and res[0]["error_code"] == 1
    finally:
        sora.set_share_window_us(old_window)
    assert old_window == 20000
    big.close(); small.close()


@pytest.mark.parametrize("depth", [1, 4])
def test_a_wait_that_gives_up_still_delivers_the_reference_rows(sora, torch_cuda, oracle, golden_dir, depth):
    """k_pipe's safety net: with the bound of the waits inside the launch at zero every hand-off that is not there at first look gives up -- what
    happens to a launch whose workgroups cannot all be resident because somebody else holds the compute units.  The finishing kernel then makes the call's data field
    again with k_frame's code and the serial trellis: the rows and MPDU bytes are the oracle's (no SORA_E_INTERNAL_TIMEOUT, no missing frame), the record counts the calls,
    and the handle leaves k_pipe alone for its next calls."""
    rng = np.random.default_rng(606 + depth)
    want_sha = "5a13a47743867e307040a009e1172b916c9015cd34fac586cafb2d0f1fd64b62"
    iq6 = pad_capture(np.load(os.path.join(golden_dir, "fsample6_40mhz_i8.npz"))["iq_i8"].astype(np.int16) << 8, 40)
    sets = [([iq6], 40, 2)]
    for mhz, n in ((20, 1), (40, 1), (20, 3 if depth == 1 else 1), (40, 2 if depth == 1 else 1)):   # (what fits the chip depth times over)
        for _ in range(3):
            sets.append(([random_capture(oracle, rng, mhz, multipath_p=0.2) for _ in range(n)], mhz, 4))
    made_again = 0
    for caps, mhz, mf in sets:
        iq, descs = batch(caps)
        rx = sora.Rx(max_captures=len(caps), max_total_samples=max(64, len(iq)), sample_rate_mhz=mhz, max_frames_per_capture=mf)
        rx.set_depth(depth); rx.set_front(4)
        assert rx.set_pipe_wait_us(0) == 20000
        assert rx.front() == 4
        d = torch_cuda.from_numpy(iq).cuda()
        got = rx.results(ticket=rx.process_dev(d, descs))
        want = oracle_results(oracle, caps, mhz)
        ok, why = same_results(got, want)
        assert ok, (mhz, len(caps), why)
        assert all(r["error_code"] != E_INTERNAL_TIMEOUT for r in got)
        if caps[0] is iq6:
            assert len(got) == 1 and got[0]["error_code"] == 1 and hashlib.sha256(got[0]["mpdu"]).hexdigest() == want_sha
        st = rx.pipe_stats()
        # (a call without a data field has no hand-off to wait for, and a frame of a symbol or two may be through before anybody looks)
        assert st["calls_made_again"] in (0, 1) and st["backoffs"] == st["calls_made_again"], st
        if caps[0] is iq6:
            assert st["calls_made_again"] == 1, st                              # 465 symbols: the trellis waves certainly looked before the soft values were there
        if st["calls_made_again"]:
            made_again += 1
            assert rx.front() == 3                                              # the handle has learnt: the three-kernel chain for a while ...
            got2 = rx.results(ticket=rx.process_dev(d, descs))
            ok, why = same_results(got2, want); assert ok, why
            assert rx.pipe_stats()["calls_made_again"] == 1                     # ... and that call was not a k_pipe launch
        rx.close()
    assert made_again >= 6, made_again


def test_the_back_off_ends(sora, torch_cuda, golden_dir):
    cap = pad_capture(np.load(os.path.join(golden_dir, "fsample6_40mhz_i8.npz"))["iq_i8"].astype(np.int16) << 8, 40)
    d = torch_cuda.from_numpy(cap).cuda(); one = [(0, len(cap), 0)]
    rx = sora.Rx(max_captures=1, max_total_samples=len(cap), sample_rate_mhz=40, max_frames_per_capture=2)
    rx.set_depth(1); rx.set_pipe_wait_us(0)
    assert rx.front() == 4
    assert rx.results(ticket=rx.process_dev(d, one))[0]["error_code"] == 1
    assert rx.pipe_stats() == {"calls_made_again": 1, "backoffs": 1}
    rx.set_pipe_wait_us(20000)
    n = 0
    while rx.front() == 3 and n < 200:
        assert rx.results(ticket=rx.process_dev(d, one))[0]["error_code"] == 1
        n += 1
    assert 60 <= n <= 66, n                                                      # 64 calls on the chain, then k_pipe again
    assert rx.results(ticket=rx.process_dev(d, one))[0]["error_code"] == 1
    assert rx.pipe_stats() == {"calls_made_again": 1, "backoffs": 1}
    rx.close()
