"""The three-kernel symbol chain (sora_rx_set_front(3): k_sym_front -> k_track_lds -> k_sym_back, sora_amd/csrc/k_rx.hip) against the oracle: the chain for few
frames in flight, the pilot tracker's loop (freqoffset.hpp:28-30, pilot.hpp:166-233) running out of tables folded into LDS (dev_arith.h TrkTables).  Same rows,
same MPDU bytes as k_frame -- on random captures (frames meeting inside a wave's sixteen symbol slots, several frames per capture, truncation, CFO, noise up to
decode failure), at both sample rates, with either trellis kernel behind it; and the automatic choice follows the handle's capacity in flight."""
import hashlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_util import batch, make_capture, oracle_results, pad_capture, same_results  # noqa: E402
from oracle.pyoracle import RATES  # noqa: E402

pytestmark = pytest.mark.gpu


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


def run_split(sora, torch, caps, mhz, max_frames=8, trellis=None):
    iq, descs = batch(caps)
    rx = sora.Rx(max_captures=max(1, len(caps)), max_total_samples=max(64, len(iq)), sample_rate_mhz=mhz, max_frames_per_capture=max_frames)
    rx.set_front(3); assert rx.front() == 3
    if trellis is not None:
        rx.set_trellis(trellis)
    rx.process_dev(torch.from_numpy(iq).cuda(), descs)
    res = rx.results()
    rx.close()
    return res


@pytest.mark.parametrize("trellis", [64, 1])
def test_random_captures_equal_the_oracle(sora, torch_cuda, oracle, trellis):
    from gpu_util import random_capture
    rng = np.random.default_rng(20260927 + trellis)
    for mhz, n in ((20, 150), (40, 100), (40, 3)):
        caps = [random_capture(oracle, rng, mhz, multipath_p=0.2) for _ in range(n)]
        ok, why = same_results(run_split(sora, torch_cuda, caps, mhz, trellis=trellis), oracle_results(oracle, caps, mhz))
        assert ok, (mhz, n, why)


def test_every_rate_long_and_short_frames(sora, torch_cuda, oracle):
    """Frames from one symbol to 1366 symbols (6 Mbps x 2500 bytes is 835; 4095-byte lengths do not exist), the tracker's request ring crossing its frame's end at
    every phase, CFO so that the rotation angle wraps many times, heavy noise (the tracker's angles all over the table)."""
    caps = []
    for i, rate in enumerate(RATES):
        for j, ln in enumerate((1, 7, 23, 24, 25, 47, 100, 511, 1500, 2500)):
            if (i + j) % 2:
                continue
            caps.append(make_capture(oracle, rate, ln, seed=7000 + 16 * i + j, rate_mhz=20, sigma=(30, 200, 700, 1500)[(i + j) % 4], tail=160, cfo_hz=(-70e3, 0, 35e3)[j % 3])[0])
    ok, why = same_results(run_split(sora, torch_cuda, caps, 20, max_frames=2), oracle_results(oracle, caps, 20))
    assert ok, why
    ok, why = same_results(run_split(sora, torch_cuda, caps[:1], 20, max_frames=2, trellis=1), oracle_results(oracle, caps[:1], 20))
    assert ok, why


def test_fsample6_and_the_automatic_choice(sora, torch_cuda, golden_dir):
    iq = np.load(os.path.join(golden_dir, "fsample6_40mhz_i8.npz"))["iq_i8"].astype(np.int16) << 8
    cap = pad_capture(iq, 40)
    rx = sora.Rx(max_captures=1, max_total_samples=len(cap), sample_rate_mhz=40, max_frames_per_capture=2)
    rx.set_depth(1)
    assert rx.front() == 4 and rx.trellis() == sora.TRELLIS_WINDOWED             # a single capture: the chain and the trellis spread over the chip as one launch (k_pipe; tests/test_gpu_pipe.py)
    rx.process_dev(torch_cuda.from_numpy(cap).cuda(), [(0, len(cap), 0)])
    got = rx.results()
    assert len(got) == 1 and got[0]["error_code"] == 1 and hashlib.sha256(got[0]["mpdu"]).hexdigest() == "5a13a47743867e307040a009e1172b916c9015cd34fac586cafb2d0f1fd64b62"
    rx.close()
    big = sora.Rx(max_captures=4096, max_total_samples=1 << 20, sample_rate_mhz=20, max_frames_per_capture=2)
    big.set_depth(8)
    assert big.front() == 1 and big.trellis() == 16                               # the full batch, eight calls in flight: a frame per wave / eight per wave
    big.set_depth(1)
    assert big.front() == 1 and big.trellis() == sora.TRELLIS_WINDOWED            # one lone call of it: k_frame, the trellis cut into units
    big.close()
