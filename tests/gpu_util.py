"""Helpers shared by the GPU parity tests: synthetic captures (oracle TX + channel) and batching."""
import numpy as np


def awgn(cap, sigma, seed):
    rng = np.random.default_rng(seed)
    x = cap.astype(np.int32) + np.rint(rng.normal(0, sigma, cap.shape)).astype(np.int32)
    return np.clip(x, -32768, 32767).astype(np.int16)


def pad_capture(x, rate_mhz):
    """Whole source bursts only (28 raw samples @40 MHz, 14 @20 MHz) -- the capture contract of sora_rx."""
    q = 28 if rate_mhz == 40 else 14
    n = (len(x) + q - 1) // q * q
    if n == len(x):
        return x
    return np.concatenate([x, np.zeros((n - len(x), 2), np.int16)])


def make_capture(oracle, rate_kbps, length, seed, rate_mhz=40, sigma=0.0, lead=0, tail=200, cfo_hz=0.0):
    rng = np.random.default_rng(seed)
    mp = rng.integers(0, 256, length).astype(np.uint8).tobytes()
    cap = oracle.tx_capture(mp, rate_kbps, seed=1 + seed % 127 if seed % 3 else 0xFF, lead=lead, tail=tail)
    if cfo_hz:
        z = (cap[:, 0].astype(np.float64) + 1j * cap[:, 1]) * np.exp(2j * np.pi * cfo_hz * np.arange(len(cap)) / 40e6)
        cap = np.stack([np.rint(z.real), np.rint(z.imag)], 1).astype(np.int16)
    if sigma:
        cap = awgn(cap, sigma, seed)
    if rate_mhz == 20:
        cap = cap[::2].copy()
    return pad_capture(cap, rate_mhz), mp


def batch(caps):
    """Concatenate captures with 4-sample aligned offsets -> (iq [N,2] int16, [(offset, nsamples, id)])."""
    descs, parts, off = [], [], 0
    for i, c in enumerate(caps):
        descs.append((off, len(c), i))
        parts.append(c)
        off += len(c)
        padn = (-off) % 4
        if padn:
            parts.append(np.zeros((padn, 2), np.int16)); off += padn
    return np.concatenate(parts) if parts else np.zeros((0, 2), np.int16), descs


def oracle_results(oracle, caps, rate_mhz):
    out = []
    for i, c in enumerate(caps):
        for r in oracle.rx_capture(c, rate_mhz):
            r = dict(r); r["capture_id"] = i; out.append(r)
    return out


KEYS = ("capture_id", "start_sample", "end_sample", "error_code", "rate_kbps", "length", "nsym", "crc32", "cfo_est", "mpdu")


def same_results(got, want):
    if len(got) != len(want):
        return False, "count %d != %d" % (len(got), len(want))
    for i, (g, w) in enumerate(zip(got, want)):
        for k in KEYS:
            if g[k] != w[k]:
                return False, "frame %d field %s: %r != %r" % (i, k, g[k] if k != "mpdu" else g[k][:16].hex(), w[k] if k != "mpdu" else w[k][:16].hex())
    return True, ""
