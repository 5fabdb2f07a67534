"""ht40_rx_f64.py -- TEST INFRASTRUCTURE: a SECOND, independently written receiver for the 40 MHz HT two-stream captures the GPU path
(sora_ht40_*, k_ht40.hip) decodes.  **The 40 MHz extension stays "parity unpinned"** (the reference has no such receiver); what this file
adds is a cross-check that is not the author's own loop-back: VERDICT r3 #9 asked for a float64 receiver written from IEEE 802.11n-2009
clause 20 alone, decoding the same captures, PSDU bytes equal.

Independence: nothing here is imported from, or shares a line with, oracle/py_ht40.py (the model that GENERATES the captures and that the GPU
kernels were developed against) or the kernels.  Sequences, carrier plans, the interleaver, puncturing, scrambler, CRC-8 and the decoder are
written from the standard's equations and tables; the algorithms are deliberately different from the fixed-point path's:
  timing by cross-correlation with the L-LTF waveform (the GPU reuses the reference's carrier sense on the decimated primary channel),
  CFO from the L-LTF repetition at 40 MHz, least-squares channel estimates, float64 MMSE detection, max-log LLRs, a full-frame float
  Viterbi with one final trace-back (the GPU: 8-bit wrapping metrics, 192 / 36 windows).

The waveform is NOT a standard-conformant PPDU in every respect; the receiver follows these documented properties of the captures (they are
properties of the test signal, stated in DESIGN.md section 7 g1, not of either receiver):
  P1  each of the two spatial streams carries its OWN PSDU of HT-SIG LENGTH bytes through its own scrambler, K = 7 encoder, puncturer and
      interleaver (BASELINE.json configs[3]: "dual Viterbi"); the standard would stream-parse ONE encoder's output (N_ES = 1 below 300 Mbit/s);
  P2  no cyclic shifts on the second chain; HT-STF is a place-holder of 4 us;
  P3  the six pilots (+-11, +-25, +-53) are +1 on both streams in every symbol (no polarity sequence, no per-symbol pattern rotation);
  P4  constellation points are sent at a modulation-dependent scale relative to the HT-LTF that is not the standard's K_MOD; the receiver
      estimates the scale from the equalised symbols (a decision-free moment estimate), so it needs no knowledge of it;
  P5  both transmit chains send the same legacy preamble and SIG fields; the upper 20 MHz half is the lower one times j (20.3.9.3.3).
"""
import zlib

import numpy as np

# ---- sequences (IEEE 802.11n-2009 20.3.9.3.3 eq. 20-8/20-11 for the legacy parts, eq. 20-25 for HT-LTF at 40 MHz)
_LLTF = np.array([1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 0,
                  1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1], float)          # carriers -26 .. 26
_HTLTF = np.concatenate([_LLTF[:26], [1], _LLTF[27:], [-1, -1, -1, 1], [0, 0, 0], [-1, 1, 1, -1], _LLTF[:26], [1], _LLTF[27:]])   # carriers -58 .. 58
assert len(_HTLTF) == 117
_PILOTS40 = (-53, -25, -11, 11, 25, 53)
_DATA40 = [k for k in range(-58, 59) if abs(k) >= 2 and k not in _PILOTS40]
assert len(_DATA40) == 108
_LEG_PILOTS = (-21, -7, 7, 21)
_LEG_DATA = [k for k in range(-26, 27) if k != 0 and k not in _LEG_PILOTS]
# MCS 8..15 (two streams, equal modulation): (N_BPSCS, R numerator, R denominator)   Table 20-35
_MCS = {8: (1, 1, 2), 9: (2, 1, 2), 10: (2, 3, 4), 11: (4, 1, 2), 12: (4, 3, 4), 13: (6, 2, 3), 14: (6, 3, 4), 15: (6, 5, 6)}


def _spectrum40(leg53):
    """the 20 MHz legacy symbol (carriers -26..26) duplicated on both halves of the 40 MHz channel, upper half rotated by +90 degrees"""
    X = np.zeros(128, complex)
    for i, k in enumerate(range(-26, 27)):
        X[(k - 32) % 128] = leg53[i]
        X[(k + 32) % 128] = 1j * leg53[i]
    return X


_LLTF_TIME = np.fft.ifft(_spectrum40(_LLTF))                                    # one 3.2 us long training symbol at 40 MHz (128 samples)


# ---- binary convolutional code K = 7, g0 = 133, g1 = 171 (octal)   17.3.5.5
def _parity(x):
    x ^= x >> 4; x ^= x >> 2; x ^= x >> 1
    return x & 1


_NEXT = np.zeros((64, 2), np.int64); _OUT = np.zeros((64, 2, 2), np.int64)
for _s in range(64):                                                            # state = the six previous input bits, newest in bit 5
    for _b in range(2):
        _reg = (_b << 6) | _s                                                   # bit 6 = the current input, bit 5 = one step back, ...
        _OUT[_s, _b, 0] = _parity(_reg & 0o133)                                 # g0 = 133 octal = 1011011, leftmost digit = the current input: delays 0, 2, 3, 5, 6
        _OUT[_s, _b, 1] = _parity(_reg & 0o171)                                 # g1 = 171 octal = 1111001: delays 0, 1, 2, 3, 6
        _NEXT[_s, _b] = _reg >> 1
_PRED = np.zeros((64, 2), np.int64); _PBIT = np.zeros((64, 2), np.int64)
_cnt = np.zeros(64, int)
for _s in range(64):
    for _b in range(2):
        _n = _NEXT[_s, _b]; _PRED[_n, _cnt[_n]] = _s; _PBIT[_n, _cnt[_n]] = _b; _cnt[_n] += 1
_PO0 = np.stack([_OUT[_PRED[:, j], _PBIT[:, j], 0] for j in range(2)], 1).astype(float)   # expected A / B bit of the branch predecessor j -> state
_PO1 = np.stack([_OUT[_PRED[:, j], _PBIT[:, j], 1] for j in range(2)], 1).astype(float)


def viterbi(llr_a, llr_b, nbits):
    """Maximum-likelihood sequence, float metrics, ONE trace-back from the best final state after the whole field.
    llr > 0 means 'bit is 1'; a punctured position carries llr 0.  -> uint8 [nbits]"""
    n = len(llr_a)
    pm = np.full(64, -1e18); pm[0] = 0.0
    surv = np.zeros((n, 64), np.uint8)
    sa = 2.0 * _PO0 - 1.0; sb = 2.0 * _PO1 - 1.0                                 # +-1 per (state, predecessor)
    for t in range(n):
        cand = pm[_PRED] + sa * llr_a[t] + sb * llr_b[t]
        pick = cand[:, 1] > cand[:, 0]
        surv[t] = pick
        pm = np.where(pick, cand[:, 1], cand[:, 0])
    s = int(np.argmax(pm))
    out = np.zeros(n, np.uint8)
    for t in range(n - 1, -1, -1):
        j = surv[t, s]
        out[t] = _PBIT[s, j]
        s = _PRED[s, j]
    return out[:nbits]


def _depuncture(llr, num, den):
    """coded-bit LLRs in transmission order -> (A, B) per input bit; Figure 17-9 patterns: 2/3 sends A0 B0 A1, 3/4 sends A0 B0 A1 B2, 5/6 A0 B0 A1 B2 A3 B4"""
    keep = {(1, 2): [(0, 0), (0, 1)], (2, 3): [(0, 0), (0, 1), (1, 0)], (3, 4): [(0, 0), (0, 1), (1, 0), (2, 1)],
            (5, 6): [(0, 0), (0, 1), (1, 0), (2, 1), (3, 0), (4, 1)]}[(num, den)]
    per = len(keep); groups = len(llr) // per
    ab = np.zeros((groups * num, 2))
    v = np.asarray(llr[:groups * per]).reshape(groups, per)
    for j, (step, which) in enumerate(keep):
        ab[step::num, which] = v[:, j]
    return ab[:, 0], ab[:, 1]


def _interleaver_40(nbpscs, iss):
    """20.3.11.7.3: the position r (after all three permutations) of coded bit k of spatial stream iss (1-based) at 40 MHz"""
    ncol, nrow, nrot = 18, 6 * nbpscs, 29
    ncbpss = ncol * nrow
    s = max(nbpscs // 2, 1)
    k = np.arange(ncbpss)
    i = nrow * (k % ncol) + k // ncol                                                       # eq. 20-19
    j = s * (i // s) + (i + ncbpss - (ncol * i) // ncbpss) % s                              # eq. 20-20
    r = (j - (((iss - 1) * 2) % 3 + 3 * ((iss - 1) // 3)) * nrot * nbpscs) % ncbpss         # eq. 20-21
    return r


def _interleaver_legacy48():
    k = np.arange(48)
    return 3 * (k % 16) + k // 16                                                           # 17.3.5.6 with N_CBPS = 48, s = 1


def _scrambler(state7, n):
    """17.3.5.4: x^7 + x^4 + 1; state7 = [x1 .. x7]"""
    st = list(state7); out = np.zeros(n, np.uint8)
    for t in range(n):
        b = st[3] ^ st[6]
        out[t] = b
        st = [b] + st[:6]
    return out


def _crc8_htsig(bits34):
    """20.3.9.4.4: shift register c7..c0 preset to ones, generator x^8 + x^2 + x + 1, the ones' complement of the remainder is sent c7 first"""
    c = [1] * 8                                                                             # c[0] = c0 ... c[7] = c7
    for m in bits34:
        fb = int(m) ^ c[7]
        c = [fb, c[0] ^ fb, c[1] ^ fb, c[2], c[3], c[4], c[5], c[6]]
    return [1 - c[7 - i] for i in range(8)]                                                 # bit 34 + i of HT-SIG


def _llr_axis(v, d, nbits_axis):
    """max-log LLRs of one PAM axis, Gray mapping of 17.3.5.7 (first bit = sign, then the inner bits); v in units where levels are +-d, +-3d, ..."""
    x = v / d
    if nbits_axis == 1:
        return [x]
    if nbits_axis == 2:                                                                     # levels -3 -1 +1 +3 <- bits 00 01 11 10
        return [x, 2.0 - np.abs(x)]
    return [x, 4.0 - np.abs(x), 2.0 - np.abs(np.abs(x) - 4.0)]                              # 64-QAM axis: -7..+7 <- 000 001 011 010 110 111 101 100


def _soft_bits(sym, nbpscs, d):
    """equalised points of one stream and symbol (108 carriers) -> LLRs of the N_CBPSS coded bits in mapping order (I bits then Q bits per point)"""
    if nbpscs == 1:
        return np.real(sym) / d
    h = nbpscs // 2
    cols = _llr_axis(np.real(sym), d, h) + _llr_axis(np.imag(sym), d, h)
    return np.stack(cols, 1).reshape(-1)


class Frame:
    def __init__(self):
        self.start = 0; self.mcs = None; self.length = None; self.l_length = None; self.sig_ok = False; self.end = 0
        self.psdu = [b"", b""]; self.fcs_ok = [False, False]


_BACKOFF = 4                                                                     # every FFT window starts this many samples inside its guard interval: a timing estimate a few
                                                                                 # samples late then still sees one symbol only (the channel estimates take the same windows)


def _sym_fft(z, at):
    return np.fft.fft(z[:, at - _BACKOFF:at - _BACKOFF + 128], axis=1) / 128.0


def _legacy_bits(Y, Hc, qbpsk):
    """one duplicated legacy symbol: per chain and half, matched to that half's channel (from the L-LTF), combined -> 48 coded-bit LLRs"""
    acc = np.zeros(48)
    for c, k in enumerate(_LEG_DATA):
        v = 0j
        for half, rot in ((-32, 1.0), (32, -1j)):                                           # undo the +90 degrees of the upper half
            b = (k + half) % 128
            v += np.sum(np.conj(Hc[:, b]) * Y[:, b]) * rot
        acc[c] = np.imag(v) if qbpsk else np.real(v)
    out = np.zeros(48)
    out[:] = acc[_interleaver_legacy48()]                                                   # coded bit k was sent at position 3 (k mod 16) + k div 16
    return out


def receive(iq, max_frames=16, detect=0.55):
    """iq: int16 [2 chains, n, 2] raw 40 MHz capture -> list of Frame (every frame whose L-LTF is found, in time order)"""
    z = iq[..., 0].astype(float) + 1j * iq[..., 1].astype(float)
    n = z.shape[1]
    ref = _LLTF_TIME / np.sqrt(np.sum(np.abs(_LLTF_TIME) ** 2))
    frames = []
    if n < 2000:
        return frames
    # normalised cross-correlation with the long training symbol, both chains; a frame shows two peaks 128 samples apart
    num = np.zeros(n - 128)
    for c in range(2):
        num += np.abs(np.correlate(z[c], ref, mode="valid")[:n - 128]) ** 2
    csum = np.concatenate([[0.0], np.cumsum(np.sum(np.abs(z) ** 2, axis=0))])
    en = csum[128:n] - csum[:n - 128] + 1e-9
    metric = num / en
    pos = 0
    while pos < n - 128 - 1200 and len(frames) < max_frames:
        m2 = metric[pos:n - 256 - 900] * 0 + np.minimum(metric[pos:n - 256 - 900], metric[pos + 128:n - 128 - 900])
        if len(m2) == 0:
            break
        cand = np.nonzero(m2 > detect)[0]
        if len(cand) == 0:
            break
        first = int(cand[0]); win = m2[first:first + 96]
        t0 = pos + first + int(np.argmax(win))                                              # first sample of the first long symbol
        f = _decode_at(z, t0)
        if f is None:
            pos = t0 + 256
            continue
        frames.append(f)
        pos = max(f.end, t0 + 256)
    return frames


def _decode_at(z, t0):
    n = z.shape[1]
    f = Frame(); f.start = t0 - 64 - 320                                                    # GI2 and the short training field in front of it
    # ---- carrier offset from the repetition of the long symbol (one estimate over both chains), then a finer one is not needed at these offsets
    a = z[:, t0:t0 + 128]; b = z[:, t0 + 128:t0 + 256]
    w = np.angle(np.sum(b * np.conj(a))) / 128.0
    rot = np.exp(-1j * w * (np.arange(n) - t0))
    y = z * rot[None]
    # ---- channel of the two halves from the averaged long symbol; noise from the difference of its two copies
    Y1 = _sym_fft(y, t0); Y2 = _sym_fft(y, t0 + 128)
    Lsp = _spectrum40(_LLTF)
    used = np.nonzero(Lsp != 0)[0]
    Hleg = np.zeros((2, 128), complex)
    Hleg[:, used] = (Y1[:, used] + Y2[:, used]) / 2.0 / Lsp[used][None]
    noise_bin = float(np.mean(np.abs(Y1[:, used] - Y2[:, used]) ** 2) / 2.0)                # variance of one FFT bin of one symbol
    # ---- L-SIG (BPSK, rate 1/2, 24 bits), 32-sample guard interval
    t = t0 + 256
    soft = _legacy_bits(_sym_fft(y, t + 32), Hleg, False)
    bits = viterbi(soft[0::2], soft[1::2], 24)
    if int(bits[:17].sum() + bits[17]) & 1 or tuple(bits[:4]) != (1, 1, 0, 1) or bits[4]:
        return None
    f.l_length = int(sum(int(bits[5 + i]) << i for i in range(12)))
    # ---- HT-SIG (two symbols, BPSK rotated by 90 degrees, 48 bits)
    s1 = _legacy_bits(_sym_fft(y, t + 160 + 32), Hleg, True); s2 = _legacy_bits(_sym_fft(y, t + 320 + 32), Hleg, True)
    soft = np.concatenate([s1, s2])
    hb = viterbi(soft[0::2], soft[1::2], 48)
    f.end = t + 480
    if list(hb[34:42]) != _crc8_htsig(hb[:34]) or hb[42:48].any():
        return f                                                                            # a frame whose HT-SIG does not check: found, not decoded
    f.mcs = int(sum(int(hb[i]) << i for i in range(7))); cbw40 = int(hb[7]); f.length = int(sum(int(hb[8 + i]) << i for i in range(16)))
    if not cbw40 or f.mcs not in _MCS or f.length == 0:
        return f
    f.sig_ok = True
    nb, rn, rd = _MCS[f.mcs]
    ndbps = 108 * nb * rn // rd
    nsym = -(-(16 + 8 * f.length + 6) // ndbps)
    t = t + 480 + 160                                                                       # behind HT-SIG and the 4 us HT-STF
    f.end = t + 160 * (2 + nsym)
    if f.end > n:
        f.sig_ok = False
        return f
    # ---- HT-LTFs: P_HTLTF = [[1, -1], [1, 1]] (rows = streams, columns = training symbols; eq. 20-27), no cyclic shift (P2)
    A = _sym_fft(y, t + 32); B = _sym_fft(y, t + 160 + 32)
    H = np.zeros((128, 2, 2), complex)                                                      # [bin][chain][stream]
    occ = []
    for k in range(-58, 59):
        v = _HTLTF[k + 58]
        if v == 0:
            continue
        bn = k % 128; occ.append(bn)
        H[bn, :, 0] = (A[:, bn] - B[:, bn]) / (2.0 * v)
        H[bn, :, 1] = (A[:, bn] + B[:, bn]) / (2.0 * v)
    # ---- MMSE detection per carrier: x = (H^H H + s2 I)^-1 H^H y, each stream rescaled to unit gain; SINR per stream for the LLR weights
    W = np.zeros((128, 2, 2), complex); gain = np.ones((128, 2))
    for bn in occ:
        Hk = H[bn]
        G = np.linalg.inv(Hk.conj().T @ Hk + noise_bin * np.eye(2)) @ Hk.conj().T
        dg = np.real(np.diag(G @ Hk))
        W[bn] = G / dg[:, None]
        gain[bn] = dg / np.maximum(1.0 - dg, 1e-6)                                          # post-detection SINR of an MMSE stream
    data_bins = np.array([k % 128 for k in _DATA40]); pil_bins = np.array([k % 128 for k in _PILOTS40])
    X = np.zeros((nsym, 2, 108), complex)
    hp = H[pil_bins, :, 0] + H[pil_bins, :, 1]                                              # what a pilot (+1 on both streams, P3) looks like on each chain
    for d in range(nsym):
        Yd = _sym_fft(y, t + 320 + 160 * d + 32)
        theta = np.angle(np.sum(np.conj(hp) * Yd[:, pil_bins].T))                           # common phase of the symbol (residual offset, phase noise)
        Yd = Yd * np.exp(-1j * theta)
        X[d] = np.einsum("bsc,cb->sb", W[data_bins], Yd[:, data_bins])
    # ---- the constellation's scale (P4) from a moment of the points: E|Re|, E|Im| of a square lattice with half-spacing d
    mean_abs = float(np.mean(np.abs(np.concatenate([X.real.ravel(), X.imag.ravel()]) if nb > 1 else X.real.ravel())))
    d0 = mean_abs / {1: 1.0, 2: 1.0, 4: 2.0, 6: 4.0}[nb]
    wts = gain[data_bins]                                                                   # [108, 2]
    for s in range(2):
        r = _interleaver_40(nb, s + 1)                                                      # coded bit k sits at position r[k] of its symbol
        llr = np.zeros((nsym, 108 * nb))
        for d in range(nsym):
            sb = _soft_bits(X[d, s], nb, d0) * np.repeat(wts[:, s], nb)
            llr[d] = sb[r]
        la, lb = _depuncture(llr.reshape(-1), rn, rd)
        nbits = 16 + 8 * f.length + 6
        dec = viterbi(la[:nsym * ndbps], lb[:nsym * ndbps], nsym * ndbps)
        # ---- descramble: the first seven SERVICE bits are zero before scrambling, so they ARE the scrambler's first seven outputs (17.3.5.4)
        first7 = dec[:7]
        st = [int(first7[6 - i]) for i in range(7)]                                          # after seven outputs the register holds them, newest first
        seq = np.concatenate([first7, _scrambler(st, len(dec) - 7)])
        plain = dec ^ seq
        payload = np.packbits(plain[16:16 + 8 * f.length], bitorder="little").tobytes()
        f.psdu[s] = payload
        f.fcs_ok[s] = len(payload) >= 4 and zlib.crc32(payload[:-4]) == int.from_bytes(payload[-4:], "little")
    return f
