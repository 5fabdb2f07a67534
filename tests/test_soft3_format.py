"""The packed soft stream (sora_amd/csrc/rx_types.h, dev_viterbi.h): three bits per soft value, value i in bits 3 i .. 3 i + 2 of a little-endian
bit stream.  A numpy model of what the producers (soft3_pack8 / soft3_store8) write and what a trellis lane does to read value i
(SoftCursor: 16 bits at byte (3 i) >> 3, shifted by (3 i) & 7) -- the format's contract, checked on a CPU; the kernels themselves are
covered by the GPU parity tests."""
import numpy as np
import pytest


def pack(values):
    """eight values -> three bytes, as soft3_pack8 / soft3_store8"""
    v = np.asarray(values, np.uint32).reshape(-1, 8)
    bits = v[:, 0] | v[:, 1] << 3 | v[:, 2] << 6 | v[:, 3] << 9 | v[:, 4] << 12 | v[:, 5] << 15 | v[:, 6] << 18 | v[:, 7] << 21
    out = np.empty((len(v), 3), np.uint8)
    out[:, 0] = bits & 0xFF; out[:, 1] = (bits >> 8) & 0xFF; out[:, 2] = (bits >> 16) & 0xFF
    return out.reshape(-1)


def fetch(stream, i, bits=3):
    """value i the way a trellis lane reads it: one 16-bit little-endian load at any byte address, a shift, a mask"""
    tb = i * bits
    w = int(stream[tb >> 3]) | int(stream[(tb >> 3) + 1]) << 8
    return (w >> (tb & 7)) & 7


@pytest.mark.parametrize("n", [8, 48, 288, 288 * 56, 104 * 3])
def test_every_value_comes_back(n):
    rng = np.random.default_rng(n)
    vals = rng.integers(0, 8, n)
    s = np.concatenate([pack(vals), np.full(2, 0xFF, np.uint8)])          # (a reader's 16-bit load reaches one byte past the stream: the buffers have slack)
    assert len(s) == 3 * n // 8 + 2
    assert all(fetch(s, i) == vals[i] for i in range(n))


def test_a_64qam_symbol_is_exactly_its_slot():
    assert len(pack(np.zeros(288, int))) == 108                            # kSoftBytesPerSlot: a frame's stream never leaves its own symbol slots


def test_constant_shift_per_chunk():
    """When a 12-step chunk is a whole number of bytes the bit offset of 'value k of the chunk' never changes (SoftCursor::kConst):
    3/4 and 1/2 at three bits, every rate at eight."""
    for bits, cw, const in ((3, 16, True), (3, 24, True), (3, 18, False), (8, 16, True), (8, 18, True), (8, 24, True)):
        shifts = {((c * cw + 5) * bits) & 7 for c in range(40)}
        assert (len(shifts) == 1) == const, (bits, cw)


def test_byte_stream_is_read_by_the_same_fetch():
    rng = np.random.default_rng(8)
    vals = rng.integers(0, 8, 100)
    s = np.concatenate([(vals | rng.integers(0, 32, 100) << 3).astype(np.uint8), np.zeros(2, np.uint8)])      # junk above the three bits
    assert all(fetch(s, i, bits=8) == vals[i] for i in range(100))
