"""k_track_lds divides the pilot-angle differences by 28 (pilot.hpp:166-233: `(th3 - th1) / 28`, C division) with a float product: trunc((float)d * 0x3D124925).
Exact for every difference of two 16-bit angles -- shown here exhaustively in IEEE single precision (numpy's float32 multiply rounds to nearest like v_mul_f32;
the conversion back truncates towards zero like v_cvt_i32_f32)."""
import numpy as np


def test_float_division_by_28_is_c_division():
    d = np.arange(-65535, 65536, dtype=np.int64)
    want = np.sign(d) * (np.abs(d) // 28)
    c = np.array([0x3D124925], np.uint32).view(np.float32)[0]
    assert float(c) == 0.0357142873108387
    got = (d.astype(np.float32) * c).astype(np.int64)
    assert np.array_equal(got, want)


def test_division_by_4_towards_zero():
    s = np.arange(-131072, 131069, dtype=np.int64)
    assert np.array_equal((s + ((s >> 31) & 3)) >> 2, np.sign(s) * (np.abs(s) // 4))
