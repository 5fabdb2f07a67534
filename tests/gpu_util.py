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


_RATES = (6000, 9000, 12000, 18000, 24000, 36000, 48000, 54000)


def multipath(x, rng, deep=False, rate_mhz=40):
    """A frequency-selective channel (SURVEY section 8d iv): 2-4 taps, the echoes 1-8 samples @20 MHz behind the direct path, 3-12 dB
    down, random phase; deep=True: one echo as strong as the direct path (within 1 dB) -- spectral nulls in which the truncating
    per-carrier division of T11aLTS::_channel_estimation meets divisors near (and at) zero (channel_11a.hpp:125-178).
    x: [n,2] at rate_mhz -> float64 [n,2] (not yet rounded)."""
    z = np.asarray(x, np.float64); z = z[:, 0] + 1j * z[:, 1]
    step = 2 if rate_mhz == 40 else 1
    y = z.copy()
    ntap = int(rng.integers(1, 4))
    for k in range(ntap):
        d = int(rng.integers(1, 9)) * step
        if deep and k == 0:
            g = 10 ** (-rng.uniform(0, 1) / 20)
        else:
            g = 10 ** (-rng.uniform(3, 12) / 20)
        y[d:] += g * np.exp(2j * np.pi * rng.random()) * z[:-d]
    y *= 1.0 / np.sqrt(1 + 0.3 * ntap)                                       # keep the level about where it was
    return np.stack([y.real, y.imag], 1)


def random_capture(o, rng, mhz, multipath_p=0.0):
    """multipath_p: probability of a frequency-selective channel (0 keeps the random stream the recorded fixtures were made with)"""
    kind = rng.integers(0, 10)
    parts = []
    if kind == 0:                                                        # noise only, sometimes loud enough to trip carrier sense
        n = int(rng.integers(2, 200)) * 28
        return pad_capture(np.rint(rng.normal(0, rng.choice([30, 300, 3000]), (n, 2))).astype(np.int16), mhz)
    nfr = int(rng.choice([1, 1, 1, 2, 3]))
    for _ in range(nfr):
        rate = int(rng.choice(_RATES)); L = int(rng.choice([1, 5, 20, 60, 150, 400, 900, 1500]))
        mp = rng.integers(0, 256, L).astype(np.uint8).tobytes()
        cap = o.tx_capture(mp, rate, seed=int(rng.integers(1, 128)), lead=int(rng.integers(0, 120)), tail=int(rng.choice([40, 160, 200, 400, 900])))
        parts.append(cap)
    x = np.concatenate(parts)
    if kind == 1:                                                        # truncated: the last frame runs past the capture
        x = x[:int(len(x) * rng.uniform(0.3, 0.95))]
    if rng.random() < 0.3:                                               # carrier frequency offset
        f = rng.uniform(-80e3, 80e3)
        z = (x[:, 0].astype(np.float64) + 1j * x[:, 1]) * np.exp(2j * np.pi * f * np.arange(len(x)) / 40e6)
        x = np.stack([np.rint(z.real), np.rint(z.imag)], 1)
    x = x.astype(np.float64)
    if multipath_p and rng.random() < multipath_p:                        # frequency-selective channel, one in four with a deep null
        x = multipath(x, rng, deep=rng.random() < 0.25)
    if rng.random() < 0.3:                                               # DC offset (TDCRemoveEx / TDCEstimator path)
        x += rng.uniform(-600, 600, size=(1, 2))
    if rng.random() < 0.2:                                               # gain
        x *= rng.uniform(0.25, 1.6)
    x = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    sigma = float(rng.choice([0, 0, 60, 150, 400, 900, 2000]))
    if sigma:
        x = awgn(x, sigma, int(rng.integers(1 << 30)))
    if mhz == 20:
        x = x[::2].copy()
    return pad_capture(x, mhz)


def multipath_capture(o, rng, mhz, deep=None):
    """One or two frames through a multipath channel (always), then CFO / noise as in random_capture."""
    parts = []
    for _ in range(int(rng.choice([1, 1, 2]))):
        rate = int(rng.choice(_RATES)); L = int(rng.choice([20, 60, 150, 400, 900, 1500]))
        mp = rng.integers(0, 256, L).astype(np.uint8).tobytes()
        parts.append(o.tx_capture(mp, rate, seed=int(rng.integers(1, 128)), lead=int(rng.integers(0, 120)), tail=int(rng.choice([160, 200, 400]))))
    x = multipath(np.concatenate(parts), rng, deep=(rng.random() < 0.4) if deep is None else deep)
    if rng.random() < 0.3:
        f = rng.uniform(-60e3, 60e3)
        z = (x[:, 0] + 1j * x[:, 1]) * np.exp(2j * np.pi * f * np.arange(len(x)) / 40e6)
        x = np.stack([z.real, z.imag], 1)
    x = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    sigma = float(rng.choice([0, 30, 60, 150, 400]))
    if sigma:
        x = awgn(x, sigma, int(rng.integers(1 << 30)))
    if mhz == 20:
        x = x[::2].copy()
    return pad_capture(x, mhz)


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


def source_position(end_sample20):
    """40 MHz source position at which RxThread sees a frame event: the end of the 28-sample source call that consumed
    20 MHz sample end_sample20 - 1 (memsource.hpp:87-114; error_code is tested after every source call)."""
    return -(-end_sample20 * 2 // 28) * 28


def upsample_40_to_44(x40):
    """Some 44 MHz capture: linear interpolation of a 40 MHz one, whole 28-sample source bursts (test input only)."""
    n44 = len(x40) * 11 // 10 // 28 * 28
    t = np.arange(n44) * (10 / 11.0)
    i = np.minimum(t.astype(np.int64), len(x40) - 2); f = (t - i)[:, None]
    x = x40.astype(np.float64)
    return np.rint(x[i] * (1 - f) + x[i + 1] * f).astype(np.int16)


def source_position_44(end_sample20):
    """44 MHz source position at which RxThread sees an event under CreateDemodGraph11a_44M: TDownSample44_40 hands on
    its k-th 28-sample burst during the first source call n after which it holds 28k samples (one output per input
    except input indexes = 1 mod 11; 44MTo40M.hpp:83-123, sampling.hpp:48-63)."""
    k = -(-end_sample20 * 2 // 28)
    n = 0
    while (28 * n - (28 * n + 9) // 11) // 28 < k:
        n += 1
    return 28 * n


def same_as_reference_graph(rows, ref_events, position=None):
    """rows: oracle / GPU result dicts of ONE 40 MHz capture; ref_events: ReferenceGraph.rx11a() of the same capture."""
    if len(rows) != len(ref_events):
        return False, "event count %d vs reference %d" % (len(rows), len(ref_events))
    for i, (x, y) in enumerate(zip(rows, ref_events)):
        if x["error_code"] != y["error_code"]:
            return False, "event %d: error_code %#x vs reference %#x" % (i, x["error_code"], y["error_code"])
        pos = (position or source_position)(x["end_sample"])
        if pos != y["sample_index"]:
            return False, "event %d: position %d vs reference %d" % (i, pos, y["sample_index"])
        if x["error_code"] in (0x1, 0x80000006):
            for f in ("rate_kbps", "length", "crc32", "mpdu"):
                if x[f] != y[f]:
                    return False, "event %d: %s differs from the reference" % (i, f)
    return True, ""


def random_capture_11b(graph, rng, multipath_p=0.0):
    """A random 44 MHz capture for the 802.11b receive graph: 1-3 frames of the REFERENCE's own modulator (1, 2, 5.5 or 11 Mbps,
    long preamble; graph = oracle.pyoracle.ReferenceGraph) with gaps, gain, DC offset, a small carrier offset, noise,
    sometimes truncated, sometimes noise only."""
    kind = rng.integers(0, 10)
    if kind == 0:
        n = int(rng.integers(4, 400)) * 28
        return np.rint(rng.normal(0, rng.choice([30, 300, 3000]), (n, 2))).astype(np.int16)
    parts = []
    for _ in range(int(rng.choice([1, 1, 1, 2, 3]))):
        rate = int(rng.choice([1000, 2000, 5500, 11000])); ln = int(rng.choice([1, 5, 14, 20, 60, 150, 400]))
        s8 = graph.tx11b(rng.integers(0, 256, ln).astype(np.uint8).tobytes(), rate)
        x = np.zeros((int(rng.integers(0, 3000)) + len(s8) + int(rng.choice([400, 1600, 2800, 6000])), 2), np.int16)
        lead = len(x) - len(s8) - int(rng.choice([400, 1600, 2800, 6000][:1])) if False else int(rng.integers(0, len(x) - len(s8) + 1))
        x[lead:lead + len(s8)] = s8.astype(np.int16) << 8
        parts.append(x)
    x = np.concatenate(parts).astype(np.float64)
    if kind == 1:
        x = x[:int(len(x) * rng.uniform(0.3, 0.95))]
    if multipath_p and rng.random() < multipath_p:                           # echoes up to two chips behind the direct path (8 samples @44 MHz)
        x = multipath(x, rng, deep=rng.random() < 0.25, rate_mhz=44)
    if rng.random() < 0.3:
        f = rng.uniform(-20e3, 20e3)
        z = (x[:, 0] + 1j * x[:, 1]) * np.exp(2j * np.pi * f * np.arange(len(x)) / 44e6)
        x = np.stack([z.real, z.imag], 1)
    if rng.random() < 0.3:
        x += rng.uniform(-600, 600, size=(1, 2))
    if rng.random() < 0.3:
        x *= rng.uniform(0.2, 1.5)
    sigma = float(rng.choice([0, 40, 150, 400, 1200, 3000]))
    if sigma:
        x += rng.normal(0, sigma, x.shape)
    x = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    return x[:len(x) // 28 * 28]


def same_as_reference_11b(rows, ref_events):
    """rows: oracle / GPU results of one 44 MHz capture; ref_events: ReferenceGraph.rx11b().  The FCS word of the
    reference holds three FCS bytes and one stale buffer byte (PHY_11b.hpp:725-731): its top byte is not compared."""
    if len(rows) != len(ref_events):
        return False, "event count %d vs reference %d" % (len(rows), len(ref_events))
    for i, (x, y) in enumerate(zip(rows, ref_events)):
        if x["error_code"] != y["error_code"]:
            return False, "event %d: error_code %#x vs reference %#x" % (i, x["error_code"], y["error_code"])
        if x["end_sample"] != y["sample_index"]:
            return False, "event %d: position %d vs reference %d" % (i, x["end_sample"], y["sample_index"])
        if x["error_code"] in (0x1, 0x80000006):
            if (x["rate_kbps"], x["length"], x["crc32"] & 0xFFFFFF, x["mpdu"]) != (y["rate_kbps"], y["length"], y["crc32"] & 0xFFFFFF, y["mpdu"]):
                return False, "event %d: rate/length/FCS/MPDU differ from the reference" % i
    return True, ""


def htsig_soft(rng, rate4, lsig_len, mcs, ht_len, flips=0, noise=0):
    """144 soft values (as T11nSigDemap emits them) of an L-SIG + HT-SIG with the given fields: rate-1/2 K=7 code, BPSK interleaver
    per 48, bit 1 -> 6 / bit 0 -> 1, optional noise and hard flips.  Written from IEEE 802.11n (HT-mixed format), not from the reference."""
    def crc8(bits34):
        crc = 0xFF
        for b in bits34:
            crc ^= b
            crc = (crc >> 1) ^ 0xE0 if crc & 1 else crc >> 1
        return (~crc) & 0xFF

    def enc(bits):
        r = 0; out = []
        for u in bits:
            r = ((r << 1) | u) & 127
            out += [bin(r & 0o155).count("1") & 1, bin(r & 0o117).count("1") & 1]
        return out
    ls = [(rate4 >> i) & 1 for i in range(4)] + [0] + [(lsig_len >> i) & 1 for i in range(12)]
    ls += [sum(ls) & 1] + [0] * 6
    h = [(mcs >> i) & 1 for i in range(7)] + [0] + [(ht_len >> i) & 1 for i in range(16)] + [1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
    c = crc8(h); h += [(c >> i) & 1 for i in range(8)] + [0] * 6
    coded = enc(ls) + enc(h)
    k = np.arange(48); idx = 3 * (k % 16) + k // 16                         # de-interleaved position k comes from interleaved idx[k]
    soft = np.zeros(144, np.uint8)
    for s_ in range(3):
        inter = np.zeros(48, int); inter[idx] = coded[48 * s_:48 * s_ + 48]
        soft[48 * s_:48 * s_ + 48] = np.where(inter == 1, 6, 1)
    if noise:
        soft = np.clip(soft.astype(int) + rng.integers(-noise, noise + 1, size=144), 0, 7).astype(np.uint8)
    for _ in range(flips):
        j = rng.integers(0, 144); soft[j] = 7 - soft[j]
    return soft


def htsig_cases(seed, n):
    """n soft-value bursts: mostly decodable SIG fields, plus every failure path of T11nSigParser and some pure noise."""
    rng = np.random.default_rng(seed); out = []
    for t in range(n):
        rate4 = int(rng.choice([0xB, 0xB, 0xB, 0xF, 0x3, 0x8])); llen = int(rng.integers(0, 1200)) if t % 7 else int(rng.integers(0, 4096))
        mcs = int(rng.choice([8, 9, 10, 10, 9, 8, 11, 0, 15, 7])); hlen = int(rng.integers(0, 1501)) if t % 5 else int(rng.integers(0, 65536))
        soft = htsig_soft(rng, rate4, llen, mcs, hlen, flips=int(rng.integers(0, 6)) if t % 3 == 0 else 0, noise=t % 4)
        if t % 50 == 49:
            soft = rng.integers(0, 8, size=144).astype(np.uint8)
        out.append(soft)
    return np.stack(out)


def capture_11n(rng, frames, sigma=20.0, cut=None, multipath_p=0.0):
    """Two-chain 40 MHz capture (int16 [n,2] each, n a multiple of 28) from `frames` = [(s0, s1)] transmit waveforms of the two TX
    chains: per frame a random gap, gain, per-chain phase, cross-talk and CFO; white noise on top; `cut` (0..1) truncates the last frame."""
    segs0, segs1 = [], []
    for i, (s0, s1) in enumerate(frames):
        gap = int(rng.integers(200, 1500)); x = float(rng.choice([0.0, 0.1, 0.3])); gain = float(rng.choice([0.3, 1.0, 2.0]))
        ph = np.exp(1j * rng.uniform(0, 2 * np.pi, 2)); cfo = rng.uniform(-3e-4, 3e-4)
        c0 = s0[:, 0] + 1j * s0[:, 1]; c1 = s1[:, 0] + 1j * s1[:, 1]; k = np.arange(len(c0))
        if multipath_p and rng.random() < multipath_p:                       # a 2x2 matrix of frequency-selective channels: every TX -> RX path its own echoes
            def path(c, direct):
                y = direct * c
                for _ in range(int(rng.integers(1, 4))):
                    d = int(rng.integers(1, 9)) * 2; g = 10 ** (-rng.uniform(3, 14) / 20) * abs(direct if direct else 0.3)
                    y[d:] += g * np.exp(2j * np.pi * rng.random()) * c[:-d]
                return y
            r0 = gain * (path(c0, ph[0]) + path(c1, x)) * np.exp(1j * cfo * k); r1 = gain * (path(c1, ph[1]) + path(c0, x)) * np.exp(1j * cfo * k)
        else:
            r0 = gain * (ph[0] * c0 + x * c1) * np.exp(1j * cfo * k); r1 = gain * (ph[1] * c1 + x * c0) * np.exp(1j * cfo * k)
        if cut is not None and i == len(frames) - 1:
            r0 = r0[:int(len(r0) * cut)]; r1 = r1[:len(r0)]
        segs0 += [np.zeros(gap, complex), r0]; segs1 += [np.zeros(gap, complex), r1]
    tail = 600 if cut is None else 0
    a = np.concatenate(segs0 + [np.zeros(tail, complex)]); b = np.concatenate(segs1 + [np.zeros(tail, complex)])
    n = len(a) // 28 * 28

    def q(z):
        z = z[:n] + rng.normal(0, sigma, n) + 1j * rng.normal(0, sigma, n)
        return np.stack([np.clip(np.rint(z.real), -32768, 32767), np.clip(np.rint(z.imag), -32768, 32767)], 1).astype(np.int16)
    return q(a), q(b)


def same_events_11n(got, want, position=None):
    """events of the 802.11n graph: error code always (and the 40 MHz source position, `position` = key of it in `want`); MCS, length, FCS and MPDU unless the header failed (the reference then reports
    whatever an earlier frame left in its context)"""
    if len(got) != len(want):
        return False, "event count %d != %d" % (len(got), len(want))
    for i, (x, y) in enumerate(zip(got, want)):
        if x["error_code"] != y["error_code"]:
            return False, "event %d: error %#x != %#x" % (i, x["error_code"], y["error_code"])
        if position is not None and x["end_sample"] != y[position]:
            return False, "event %d: position %d != %d" % (i, x["end_sample"], y[position])
        if x["error_code"] != 0x80000005 and (x["rate_kbps"], x["length"], x["crc32"], x["mpdu"]) != (y["rate_kbps"], y["length"], y["crc32"], y["mpdu"]):
            return False, "event %d differs" % i
    return True, ""
