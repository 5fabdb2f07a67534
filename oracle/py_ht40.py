"""py_ht40.py -- TEST INFRASTRUCTURE.  **Parity unpinned**: the reference has no 40 MHz receive graph to be held to.

BASELINE.json configs[3] names an "802.11n 2x2 MIMO 40 MHz HT (128-pt FFT, MMSE MIMO detect, dual Viterbi)" receiver; the reference's
own 802.11n graph is 20 MHz, zero-forcing, one decoder, MCS 8-10 only (kernel/bb/Brick11/src/PHY_11n.hpp:497, channel_11n.hpp:423-433).
This module is an independent numpy MODEL of the data field of such a frame, written from IEEE 802.11n-2009 clause 20 (HT-mixed format,
40 MHz, N_SS = 2), used to generate the captures the GPU path (sora_ht40_*, sora_amd/csrc/k_ht40.hip) is tested and benchmarked on, and
to decode them in floating point for cross-checks.  What is modelled:
  * two spatial streams, each with its OWN K = 7 (133, 171) encoder ("dual Viterbi": one decoder per stream), scrambler, puncturing
    (1/2, 2/3, 3/4 with the reference's patterns, viterbi.hpp:166-187) and HT interleaver for 40 MHz (N_COL 18, N_ROW 6 N_BPSC, N_ROT 29);
  * 114 occupied carriers -58..-2, 2..58: 108 data + 6 pilots (+-11, +-25, +-53; sent as +1 on both streams -- the reference's
    TPilotTrack_11n takes the pilots' phases as they are, pilot_11n.hpp:84-141);
  * two HT-LTF symbols with P = [[1, -1], [1, 1]] (as TMimoChannelEst expects them, channel_11n.hpp:329-443), then the data symbols,
    128-point IFFT + 32-sample cyclic prefix at 40 MHz;
  * constellation levels placed where the reference's soft-demapper tables (dsp_demap.h, regenerated in dev_11n.h) expect them after
    the x128 scaling of TMimoChannelComp.
Not modelled: the legacy preamble / HT-SIG (the reference's 20 MHz front end finds and parses those), cyclic shifts, STBC, short GI.
"""
import zlib

import numpy as np

PILOTS = (-53, -25, -11, 11, 25, 53)
DATA_CARRIERS = [k for k in list(range(-58, -1)) + list(range(2, 59)) if k not in PILOTS]          # 108, in demapping order
assert len(DATA_CARRIERS) == 108
# HT-LTF for 40 MHz, carriers -58..58 (IEEE 802.11n-2009 eq. 20-24)
_L = [1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1]
_R = [1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1]
HTLTF40 = np.array(_L + [0] + _R + [-1, -1, -1, 1, 0, 0, 0, -1, 1, 1, -1] + _L + [0] + _R, np.int8)
HTLTF40[26] = 1; HTLTF40[26 + 64] = 1                                                                # the former DC positions of the two halves carry +1
assert len(HTLTF40) == 117 and HTLTF40[58] == 0
LEVEL = {1: 48.0, 2: 48.0, 4: 31.25, 6: 17.3}          # constellation level spacing / 2 after x128 scaling, per N_BPSC (see dev_11n.h kRuns)
N_ROT = 29


def ndbps(nbpsc, code_rate):
    return 108 * nbpsc * (1, 2, 3)[code_rate] // (2, 3, 4)[code_rate]


def nsym_for(lengths, nbpsc, code_rate):
    return -(-(16 + 8 * max(lengths) + 6) // ndbps(nbpsc, code_rate))


def scramble_seq(seed, n):
    s = [(seed >> i) & 1 for i in range(7)]                      # s[0] = x^1 ... register as the reference's T11aSc keeps it
    out = np.zeros(n, np.uint8)
    st = list(s)
    for i in range(n):
        b = st[3] ^ st[6]
        out[i] = b
        st = [b] + st[:6]
    return out


def encode(bits):
    """K = 7 (133, 171) as kernel/bb/Brick11/src/conv_enc.hpp:6-14: G0 = x^s4^s3^s1^s0, G1 = x^s0^s3^s4^s5, s = (s >> 1) | (x << 5)."""
    s = 0; a = np.zeros(len(bits), np.uint8); b = np.zeros(len(bits), np.uint8)
    for i, x in enumerate(bits):
        x = int(x)
        a[i] = x ^ (s >> 4 & 1) ^ (s >> 3 & 1) ^ (s >> 1 & 1) ^ (s & 1)
        b[i] = x ^ (s & 1) ^ (s >> 3 & 1) ^ (s >> 4 & 1) ^ (s >> 5 & 1)
        s = (s >> 1) | (x << 5)
    return a, b


def puncture(a, b, code_rate):
    if code_rate == 0:
        return np.stack([a, b], 1).reshape(-1)
    if code_rate == 1:                                           # 2/3: (A0 B0) (A1)
        n = len(a) // 2
        return np.stack([a[0::2][:n], b[0::2][:n], a[1::2][:n]], 1).reshape(-1)
    n = len(a) // 3                                              # 3/4: (A0 B0) (A1) (B2)
    return np.stack([a[0::3][:n], b[0::3][:n], a[1::3][:n], b[2::3][:n]], 1).reshape(-1)


def interleave_map(nbpsc, iss):
    """position of coded bit k after the HT interleaver (40 MHz), stream iss = 0, 1"""
    ncbpss = 108 * nbpsc; s = max(nbpsc // 2, 1); ncol = 18; nrow = 6 * nbpsc
    k = np.arange(ncbpss)
    i = nrow * (k % ncol) + k // ncol
    j = s * (i // s) + (i + ncbpss - (ncol * i) // ncbpss) % s
    r = (j - ((iss * 2) % 3 + 3 * (iss // 3)) * N_ROT * nbpsc) % ncbpss
    return r


def qam(bits, nbpsc):
    """Gray mapping, LSB-first per axis as the 802.11 tables; amplitude = LEVEL * odd integer (x128 domain)."""
    d = LEVEL[nbpsc]
    if nbpsc == 1:
        return (2.0 * bits - 1.0) * d + 0j
    h = nbpsc // 2
    b = bits.reshape(-1, nbpsc)
    def axis(bb):
        if h == 1: return 2.0 * bb[:, 0] - 1.0
        if h == 2: return (2.0 * bb[:, 0] - 1.0) * (3.0 - 2.0 * bb[:, 1])                      # 00 -3, 01 -1, 11 +1, 10 +3
        return (2.0 * bb[:, 0] - 1.0) * np.array([7.0, 5.0, 1.0, 3.0])[(2 * bb[:, 1] + bb[:, 2]).astype(int)]   # 000 -7, 001 -5, 011 -3, 010 -1, 110 +1, 111 +3, 101 +5, 100 +7
    return (axis(b[:, :h]) + 1j * axis(b[:, h:])) * d


def stream_bits(psdu_with_fcs, nsym, nbpsc, code_rate, seed):
    n = nsym * ndbps(nbpsc, code_rate)
    data = np.zeros(n, np.uint8)
    payload = np.unpackbits(np.frombuffer(psdu_with_fcs, np.uint8), bitorder="little")
    data[16:16 + len(payload)] = payload
    scr = data ^ scramble_seq(seed, n)
    scr[16 + len(payload):16 + len(payload) + 6] = 0             # tail
    return scr


def add_fcs(mpdu_nofcs):
    return bytes(mpdu_nofcs) + int(zlib.crc32(bytes(mpdu_nofcs))).to_bytes(4, "little")


def bin_of(k):
    return k % 128


def tx(psdus, nbpsc, code_rate, seeds=(0x5D, 0x2B), ltf_amp=1.0):
    """psdus: two byte strings WITH FCS.  -> complex128 [2, (2 + nsym) * 160]: the two TX streams, HT-LTF x2 then data; unit = the
    amplitude of an HT-LTF carrier (= 128 in the demapper's domain)."""
    nsym = nsym_for([len(p) for p in psdus], nbpsc, code_rate)
    X = np.zeros((2, 2 + nsym, 128), complex)
    for k in range(-58, 59):
        v = HTLTF40[k + 58]
        X[0, 0, bin_of(k)] = v;  X[0, 1, bin_of(k)] = -v           # P = [[1, -1], [1, 1]]
        X[1, 0, bin_of(k)] = v;  X[1, 1, bin_of(k)] = v
    for s in range(2):
        bits = stream_bits(psdus[s], nsym, nbpsc, code_rate, seeds[s])
        a, b = encode(bits)
        coded = puncture(a, b, code_rate)
        ncbpss = 108 * nbpsc
        imap = interleave_map(nbpsc, s)
        for d in range(nsym):
            blk = coded[d * ncbpss:(d + 1) * ncbpss]
            il = np.zeros(ncbpss, np.uint8); il[imap] = blk
            pts = qam(il.astype(float), nbpsc) / 128.0
            for c, k in enumerate(DATA_CARRIERS):
                X[s, 2 + d, bin_of(k)] = pts[c]
            for k in PILOTS:
                X[s, 2 + d, bin_of(k)] = 1.0 * (LEVEL[1] / 128.0) * 2.0
    x = np.fft.ifft(X, axis=2) * 128.0 * ltf_amp
    x = np.concatenate([x[:, :, -32:], x], axis=2)
    return x.reshape(2, -1), nsym


def channel(x, H, sigma, rng, scale=250.0, cfo_step=0.0, lead=0):
    """x: [2, n] TX streams -> int16 [2, n + lead, 2] RX chains: y = H x * scale (+ CFO, + noise)."""
    y = (np.asarray(H, complex) @ x) * scale
    n = y.shape[1]
    if cfo_step:
        y = y * np.exp(1j * 2 * np.pi * cfo_step / 65536.0 * np.arange(n))[None]
    y = np.concatenate([np.zeros((2, lead), complex), y], axis=1)
    y = y + (rng.normal(0, sigma, y.shape) + 1j * rng.normal(0, sigma, y.shape)) if sigma else y
    out = np.stack([y.real, y.imag], axis=2)
    return np.clip(np.rint(out), -32768, 32767).astype(np.int16)


# ---- floating-point receiver (cross-check of intermediate quantities; not bit-exact with the GPU's fixed-point path)
def rx_symbols(iq, offset, nsym):
    z = iq[:, offset:offset + (2 + nsym) * 160, 0].astype(float) + 1j * iq[:, offset:offset + (2 + nsym) * 160, 1]
    z = z.reshape(2, 2 + nsym, 160)[:, :, 32:]
    return np.fft.fft(z, axis=2) / 128.0


def mmse_weights(Y, noise_var):
    """Y: [2 chains, 2 + nsym, 128] -> W [128, 2, 2] with x = W y (x128 domain handled by the caller)."""
    W = np.zeros((128, 2, 2), complex)
    for k in range(-58, 59):
        v = HTLTF40[k + 58]
        if v == 0: continue
        b = bin_of(k)
        p = Y[:, 0, b]; q = Y[:, 1, b]
        H = np.stack([(p - q) / 2 * v, (p + q) / 2 * v], axis=1)                                    # columns = streams
        G = H.conj().T @ H + noise_var * np.eye(2)
        Wb = np.linalg.solve(G, H.conj().T)
        W[b] = Wb / np.real(np.diag(Wb @ H))[:, None]                                                # unbiased MMSE
    return W


# ------------------------------------------------------------------------------------------------ the preamble (HT-mixed format, 40 MHz)
# IEEE 802.11n-2009 20.3.9: L-STF, L-LTF, L-SIG and HT-SIG are the 20 MHz legacy waveforms sent on BOTH halves of the 40 MHz channel,
# the upper half rotated by +90 degrees; then HT-STF (4 us), the HT-LTFs and the data field at the full width.  Both TX chains send the
# same legacy part (no cyclic shifts in this model).  What a receiver gets from this: a 20 MHz front end that reads the even samples of
# y[n] = x[n] j^n (a shift by +10 MHz: the lower copy lands on DC, the upper copy aliases onto it) sees (1 + j) times the 20 MHz legacy
# waveform through the channel H_lower + j H_upper -- so the reference's own 802.11n front end (carrier sense, L-LTF, the SIG decoder and
# T11nSigParser, kernel/bb/Brick11/src/PHY_11n.hpp:400-514) finds and parses it unchanged; oracle-side that is how this part is PINNED
# (tests/test_ht40_preamble_model.py runs the restated reference receiver on it).
_STF = np.zeros(53, complex)
for _k, _v in zip(range(-24, 25, 4), [1, -1, 1, -1, -1, 1, 0, -1, -1, 1, 1, 1, 1]):
    _STF[_k + 26] = _v * (1 + 1j) * np.sqrt(13.0 / 6.0) * np.sqrt(0.5)
_LTF = np.array([1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 0,
                 1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1], float)
_LEG_DATA = [k for k in range(-26, 27) if k not in (-21, -7, 0, 7, 21)]
# two-stream MCS 8..14 -> (N_BPSC, code rate index); MCS 15 (5/6) has no decoder here
MCS2 = {8: (1, 0), 9: (2, 0), 10: (2, 2), 11: (4, 0), 12: (4, 2), 13: (6, 1), 14: (6, 2)}


def _dup40(sym20):
    """53 legacy carriers (-26..26) -> 128-bin spectrum of the 40 MHz channel: lower copy at -32 + k, upper copy at +32 + k times j"""
    X = np.zeros(128, complex)
    for i, k in enumerate(range(-26, 27)):
        X[(k - 32) % 128] = sym20[i]
        X[(k + 32) % 128] = 1j * sym20[i]
    return X


def _leg_symbol(coded48, qbpsk, amp):
    il = np.zeros(48)
    k = np.arange(48)
    il[3 * (k % 16) + k // 16] = coded48                                     # the 11a BPSK interleaver (N_CBPS 48): receiver side k <- 3 (k & 15) + (k >> 4)
    s = np.zeros(53, complex)
    for c, kk in enumerate(_LEG_DATA):
        v = (2.0 * il[c] - 1.0) * amp
        s[kk + 26] = 1j * v if qbpsk else v
    for kk, p in zip((-21, -7, 7, 21), (1, 1, 1, -1)):
        s[kk + 26] = p * amp
    return s


def ht_sig_bits(mcs, length, cbw40=1):
    b = np.zeros(48, np.uint8)
    for i in range(7): b[i] = (mcs >> i) & 1
    b[7] = cbw40
    for i in range(16): b[8 + i] = (length >> i) & 1
    b[24] = 1; b[25] = 1; b[26] = 1                                          # smoothing, not sounding, reserved
    crc = 0xFF                                                               # CRC-8 over bits 0..33 exactly as T11nSigParser recomputes it (PHY_11n.hpp:486-492)
    for i in range(34):
        crc ^= int(b[i])
        crc = (crc >> 1) ^ 0xE0 if crc & 1 else crc >> 1
    c = (~crc) & 0xFF
    for i in range(8): b[34 + i] = (c >> i) & 1
    return b


def l_sig_bits(l_length):
    b = np.zeros(24, np.uint8)
    b[0], b[1], b[2], b[3] = 1, 1, 0, 1                                      # RATE = 6 Mbps
    for i in range(12): b[5 + i] = (l_length >> i) & 1
    b[17] = int(b[:17].sum() & 1)
    return b


def tx_frame(psdus, mcs, seeds=(0x5D, 0x2B), sig_amp=1.0):
    """A whole HT-mixed 40 MHz two-stream frame: L-STF, L-LTF, L-SIG, HT-SIG, HT-STF, 2 x HT-LTF, data.  psdus: two byte strings WITH FCS
    and of EQUAL length (the HT-SIG LENGTH field is one number; each spatial stream carries its own PSDU of that length through its own
    encoder -- BASELINE.json configs[3]'s 'dual Viterbi').  -> (complex128 [2, n] @40 MHz, nsym, first sample of HT-LTF 1)."""
    assert len(psdus[0]) == len(psdus[1]) and mcs in MCS2
    nb, cr = MCS2[mcs]
    data, nsym = tx(psdus, nb, cr, seeds)
    dur_us = 36 + 4 * nsym                                                   # L-STF 8, L-LTF 8, L-SIG 4, HT-SIG 8, HT-STF 4, HT-LTFs 8 -> 40; the legacy part announces what follows L-SIG
    l_length = max(1, -(-(dur_us + 4 - 20) // 4) * 3 - 3)
    parts = []
    stf = np.fft.ifft(_dup40(_STF)) * 128.0
    parts.append(np.tile(stf, 3)[:320])                                      # 8 us: the 0.8 us pattern repeats (period 32 samples @40 MHz)
    ltf = np.fft.ifft(_dup40(_LTF)) * 128.0
    parts.append(np.concatenate([ltf[-64:], ltf, ltf]))                      # GI2 + two long symbols
    a, b = encode(np.concatenate([l_sig_bits(l_length)]))
    sym = np.fft.ifft(_dup40(_leg_symbol(np.stack([a, b], 1).reshape(-1), False, sig_amp))) * 128.0
    parts.append(np.concatenate([sym[-32:], sym]))
    a, b = encode(ht_sig_bits(mcs, len(psdus[0])))
    coded = np.stack([a, b], 1).reshape(-1)
    for h in range(2):
        sym = np.fft.ifft(_dup40(_leg_symbol(coded[48 * h:48 * h + 48], True, sig_amp))) * 128.0
        parts.append(np.concatenate([sym[-32:], sym]))
    parts.append(np.tile(stf, 2)[:160])                                      # HT-STF (the duplicated short symbol: the receiver only needs its place)
    pre = np.concatenate(parts)
    x = np.concatenate([np.stack([pre, pre]), data], axis=1)
    return x, nsym, len(pre)


def front_end_view(iq):
    """What the reused 20 MHz front end reads: the even samples of x[n] j^n = (-1)^m x[2m], as a 40 MHz-rate capture whose even samples
    carry it (the reference's TDownSample2 keeps the even samples).  iq: int16 [n, 2] -> int16 [n, 2]"""
    z = np.asarray(iq, np.int32).copy()
    z[2::4] = np.clip(-z[2::4], -32768, 32767)                                # m odd <-> n = 2m = 2 mod 4
    return z.astype(np.int16)
