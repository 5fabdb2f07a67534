"""How a call's frames are cut into the units of the window-parallel trellis (sora_amd/csrc/dev_winplan.h: host-and-device inline functions) -- compiled here with the host
compiler and checked as plain arithmetic: every unit of a frame sits in exactly one slot, the three layouts are permutations, a lone frame's single-window units never share
a wave with another trace-back class and its last unit sits alone (what k_pipe's last wave relies on: DESIGN.md section 3.10)."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r"""
#define __host__
#define __device__
#include "dev_winplan.h"
extern "C" {
unsigned w_events(unsigned length, unsigned cr) { return sora::win_events(length, cr, 256u, 24u); }
unsigned w_units_per_frame(unsigned n, unsigned target) { return sora::win_units_per_frame(n, target); }
unsigned w_per_unit(unsigned nev, unsigned q) { return sora::win_per_unit(nev, q); }
unsigned w_unit_at(unsigned p, unsigned nun) { return sora::win_unit_at(p, nun); }
unsigned w_unit_lone(unsigned p, unsigned nun) { return sora::win_unit_lone(p, nun); }
unsigned w_slots(unsigned n, unsigned q) { return sora::win_slots(n, q); }
unsigned w_max_units() { return sora::kWinMaxUnits; }
unsigned w_lone_pad() { return sora::kWinLonePad; }
}
"""


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("winplan")
    src = d / "winplan.cpp"; src.write_text(SRC)
    so = d / "libwinplan.so"
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-I", os.path.join(ROOT, "sora_amd", "csrc"), str(src), "-o", str(so)])
    L = ctypes.CDLL(str(so))
    for f in ("w_events", "w_units_per_frame", "w_per_unit", "w_unit_at", "w_unit_lone", "w_slots", "w_max_units", "w_lone_pad"):
        getattr(L, f).restype = ctypes.c_uint
    return L


NONE = 0xFFFFFFFF


def test_a_frame_never_has_more_trace_backs_than_units_allowed(lib):
    assert lib.w_max_units() == 80
    for cr in (0, 1, 2):
        assert max(lib.w_events(n, cr) for n in range(1, 2501)) <= 80
        assert lib.w_events(1, cr) == 1


def test_units_per_frame_and_windows_per_unit(lib):
    assert lib.w_units_per_frame(1, 16384) == 80 and lib.w_units_per_frame(4096, 16384) == 4 and lib.w_units_per_frame(100000, 16384) == 1
    for nev in range(1, 81):
        for q in (1, 2, 4, 8, 80):
            m = lib.w_per_unit(nev, q)
            assert m >= 1 and (m == 1 or m % 3 == 0) and (nev + m - 1) // m <= max(q, 1)    # never more units than the frame's share


def test_sorted_layout_is_a_permutation(lib):
    for nun in range(1, 81):
        assert sorted(lib.w_unit_at(p, nun) for p in range(nun)) == list(range(nun))


def test_lone_layout_no_wave_mixes_classes_and_the_last_unit_sits_alone(lib):
    pad = lib.w_lone_pad()
    for nun in range(1, 81):
        slots = lib.w_slots(1, 80)
        assert slots == 80 + pad
        units = [lib.w_unit_lone(p, nun) for p in range(slots)]
        held = [u for u in units if u != NONE]
        assert sorted(held) == list(range(nun)), nun                                         # every unit exactly once, inside the slots a list of one frame gets
        for w in range(0, slots, 8):
            us = [u for u in units[w:w + 8] if u != NONE]
            if not us:
                continue
            if nun - 1 in us:
                assert us == [nun - 1], (nun, w, us)                                         # the frame's last unit: alone in its wave
            else:
                assert len({u % 3 for u in us}) == 1 and us == sorted(us), (nun, w, us)      # one trace-back class per wave
    assert lib.w_slots(7, 80) == 560                                                         # several frames: unit-major slots, no padding
