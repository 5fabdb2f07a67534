"""The product's look-up tables are pinned: libsora_hip.so generates them with the host's libm when a process first
needs them, compares each table's sha256 with a constant compiled into the library and refuses to work on a mismatch.  Here: this build's tables
hash to their pins; the pins of the tables the oracle also holds are the digests tests/test_oracle_luts.py pins the oracle's copies with (those
were compared entry for entry with /root/reference/kernel/core/inc/intalglut.h:4,3648,7332, fft_lut_twiddle.h:61433-61600,
Brick11/src/demapper.h:55-130); and -- on the GPU -- the device-resident copies read back are those bytes."""
import hashlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def sora():
    import sora_amd
    sora_amd.load(build_if_missing=False)
    return sora_amd


def test_every_table_hashes_to_its_pin(sora):
    names = sora.table_names()
    assert len(names) == 20 and {"usin", "ucos", "rot", "uatan2", "demap", "tw64", "tw16", "tw128", "tw32", "tw8", "dsp_sincos", "dsp_atan"} <= set(names)
    for n in names:
        pin = sora.table_pin(n)
        assert pin and len(pin) == 64 and pin != "0" * 64, n
        assert sora.table_digest(n) == pin, n
    assert sora.table_pin("no such table") is None


def test_pins_are_the_reference_checked_digests(sora, oracle):
    import test_oracle_luts as t
    assert sora.table_pin("usin") == t.PINS["usin_lut"] and sora.table_pin("ucos") == t.PINS["ucos_lut"] and sora.table_pin("uatan2") == t.PINS["uatan2_lut"]
    # the demapper's four step tables are one 1024-byte array in the library
    assert sora.table_pin("demap") == hashlib.sha256(b"".join(oracle.demap_lut(w).tobytes() for w in range(4))).hexdigest()
    # twiddles: the library packs (re, im) as int16 pairs, k = 1..3 back to back
    for name, n in (("tw64", 64), ("tw16", 16), ("tw128", 128), ("tw32", 32)):
        want = np.concatenate([oracle.twiddle(n, k).astype(np.int16).reshape(-1) for k in (1, 2, 3)]).tobytes()
        assert sora.table_pin(name) == hashlib.sha256(want).hexdigest(), name
    assert sora.table_pin("sts") == hashlib.sha256(oracle.sts_pattern().astype(np.int16).tobytes()).hexdigest()
    # rot = {ucos, -usin} packed
    rot = np.stack([oracle.ucos_lut().astype(np.int16), (-oracle.usin_lut().astype(np.int32)).astype(np.int16)], 1)
    assert sora.table_pin("rot") == hashlib.sha256(rot.tobytes()).hexdigest()
    # the 802.11n graph's dsp_math tables against the oracle's
    import ctypes
    oracle.L.so_dsp_sincos_table.restype = ctypes.POINTER(ctypes.c_int16); oracle.L.so_dsp_atan_table.restype = ctypes.POINTER(ctypes.c_int16)
    sc = np.ctypeslib.as_array(oracle.L.so_dsp_sincos_table(), shape=(65536, 2)).copy()
    at = np.ctypeslib.as_array(oracle.L.so_dsp_atan_table(), shape=(4097,)).copy()
    assert sora.table_pin("dsp_sincos") == hashlib.sha256(sc.tobytes()).hexdigest()
    assert sora.table_pin("dsp_atan") == hashlib.sha256(at.tobytes()).hexdigest()


@pytest.mark.gpu
def test_device_resident_tables_are_the_pinned_bytes(sora):
    import torch
    assert torch.cuda.is_available()
    sora.fft64(torch.zeros((1, 64, 2), dtype=torch.int16, device="cuda"))        # (any stage call brings the tables up)
    for n in sora.table_names():
        assert hashlib.sha256(sora.table_read(n)).hexdigest() == sora.table_pin(n), n
