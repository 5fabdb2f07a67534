"""Failure rate of the window-parallel trellis's verification, by warm-up length and noise (tools/winmodel/winmodel.c)."""
import ctypes, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
L = ctypes.CDLL(os.path.join(HERE, "libwinmodel.so"))
u8p = ctypes.POINTER(ctypes.c_uint8); u32p = ctypes.POINTER(ctypes.c_uint32)
L.wm_sequential.argtypes = [u8p, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32, u8p]
L.wm_windowed.argtypes = [u8p, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, u8p, u32p, u32p]

def par(v):
    v = np.asarray(v); v = v ^ (v >> 4); v = v ^ (v >> 2); v = v ^ (v >> 1); return v & 1

def encode(bits, cr):
    """K=7 encoder in the oracle's convention (register r = oldest<<6 | ... | newest), punctured: the coded-bit stream."""
    r = 0; out = []
    for i, x in enumerate(bits):
        r = ((r << 1) | int(x)) & 127
        a = bin(r & 0o155).count("1") & 1; b = bin(r & 0o117).count("1") & 1
        if cr == 0: out += [a, b]
        elif cr == 1: out += [a, b] if i % 2 == 0 else [a]
        else: out += [a, b] if i % 3 == 0 else ([a] if i % 3 == 1 else [b])
    return np.array(out, np.uint8)

def soft_stream(rng, length, cr, ndbps, amp, sigma):
    nbits = 16 + 8 * length + 6
    nsym = -(-nbits // ndbps)
    bits = rng.integers(0, 2, nsym * ndbps, dtype=np.uint8); bits[nbits - 6:nbits] = 0
    c = encode(bits, cr).astype(np.float64)
    v = 3.5 + (2 * c - 1) * amp + rng.normal(0, sigma, len(c))
    return np.clip(np.round(v), 0, 7).astype(np.uint8)

def run(soft, cr, length, W, mwin):
    n = len(soft); out = np.zeros(length + 2 + 64, np.uint8); ref = np.zeros(length + 2 + 64, np.uint8)
    sp = soft.ctypes.data_as(u8p)
    r0 = L.wm_sequential(sp, n, cr, length, ref.ctypes.data_as(u8p))
    nu = ctypes.c_uint32(); ff = ctypes.c_uint32()
    f = L.wm_windowed(sp, n, cr, length, W, mwin, out.ctypes.data_as(u8p), ctypes.byref(nu), ctypes.byref(ff))
    assert f >= 0, f
    same = bool((out[:length + 2] == ref[:length + 2]).all())
    return f, nu.value, same, ref[:r0]

if __name__ == "__main__":
    from oracle.pyoracle import Oracle
    o = Oracle()
    rng = np.random.default_rng(1)
    # the model's serial decode is the oracle's
    for cr, nd in ((0, 24), (1, 192), (2, 216)):
        for length in (1, 7, 40, 100, 1500):
            s = soft_stream(rng, length, cr, nd, 2.0, 1.5)
            f, nu, same, ref = run(s, cr, length, 96, 1)
            want = o.viterbi_frame(s, cr, length)
            assert bytes(ref[:length + 2]) == bytes(want[:length + 2]), (cr, length)
    print("model == oracle (serial)")
    for cr, nd, length in ((2, 216, 1500), (0, 24, 1392), (1, 192, 1500)):
        for amp, sigma, label in ((3.5, 0.5, "clean"), (2.5, 1.0, "good"), (2.0, 1.5, "fair"), (1.5, 2.0, "poor"), (1.0, 2.5, "bad"), (0, 100, "noise")):
            for W in (48, 96, 144, 192):
                for mwin in (1, 3, 12):
                    nf = nb = nbad = nwrong = 0
                    for it in range(30 if label != "noise" else 10):
                        s = soft_stream(rng, length, cr, nd, amp, sigma) if label != "noise" else rng.integers(0, 8, (-(-(16 + 8 * length + 6) // nd)) * nd * (2 if cr == 0 else 1) * (1 if cr == 0 else 1) // 1, dtype=np.uint8)
                        if label == "noise": s = rng.integers(0, 8, len(soft_stream(rng, length, cr, nd, 1, 1)), dtype=np.uint8)
                        f, nu, same, _ = run(s, cr, length, W, mwin)
                        nf += f; nb += nu - 1; nbad += f > 0
                        if f == 0 and not same: nwrong += 1
                        assert not (f == 0 and not same), "verified but different!"
                    print(f"cr={cr} {label:6s} W={W:3d} mwin={mwin:2d}: boundaries failed {nf}/{nb}  frames with a failure {nbad}")
