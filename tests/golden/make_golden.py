#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REFERENCE (run in the build container only).

  fsample6_40mhz_i8.npz  kernel/test-data/fsample-6.dmp, de-framed (brickutil.h:20-58) and 14->16-bit
                         sign-fixed ((int16)(raw<<2)); every value is then a multiple of 256, so the
                         stream is stored as int8 (value>>8).  Golden: sha256 of the dump, MPDU sha256, FCS.
  ref_vectors.npz        input/output pairs produced by the reference's OWN SSE kernels compiled into
                         oracle/_ref/libsora_ref.so (FFT<64>/IFFT<64>/IFFT<128>, vcs mul, demap LUT walk,
                         Viterbi_sig11, TViterbiCore frame decodes at 1/2, 2/3, 3/4 under noise, Down44to40).
  refgraph_events.npz    what the reference's OWN receive graph (oracle/_ref/libsora_refgraph.so = CreateDemodGraph11a_40M
                         compiled from the reference sources) reports for 400 seeded random captures
                         (tests/gpu_util.random_capture): per event the capture index, error code, source position,
                         FCS and the first 8 bytes of the MPDU's sha256; and length + sha256 prefix of what the reference's
                         modulation graph (CreateModGraph11a_40M) emits for a list of frames.
  refgraph_11b.npz       802.11b: the reference modulator's output (COMPLEX8 @44 MHz) for six 1/2 Mbps frames and the events
                         its receive graph reports for captures made of them (tests/test_oracle_11b.channel_11b).
  refgraph_11b_cck.npz   the same for six 5.5 / 11 Mbps CCK frames (`python make_golden.py 11b_cck` writes only this file).
  ref_vectors_11n.npz    802.11n stage bricks: inputs and what the reference's own T11nDemap* / T11nDeinterleave*_S{0,1} bricks make of them.
All files travel to the GPU box; /root/reference does not.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.pyoracle import Oracle, Reference, ReferenceGraph, CR_12, CR_23, CR_34  # noqa: E402
from gpu_util import random_capture  # noqa: E402

REF = os.environ.get("SORA_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def record_11b(G, name, cases, seed):
    """What the reference's 11b modulation graph emits for `cases` = [(rate_kbps, mpdu_bytes)], and what its receive graph reports
    for the captures tests/test_oracle_11b.channel_11b makes of them."""
    from test_oracle_11b import channel_11b
    b = {"frames": len(cases)}; evs = {"count": [], "error": [], "position": [], "rate": [], "length": [], "crc": []}
    rng = np.random.default_rng(seed)
    for f, (rate, ln) in enumerate(cases):
        s8 = G.tx11b(rng.integers(0, 256, ln).astype(np.uint8).tobytes(), rate)
        b["tx_%d" % f] = s8
        for rep in range(3):
            ev = G.rx11b(channel_11b(s8, 100 * f + rep))
            evs["count"].append(len(ev))
            for e in ev:
                b["mpdu_%d" % len(evs["error"])] = np.frombuffer(e["mpdu"], np.uint8)
                evs["error"].append(e["error_code"]); evs["position"].append(e["sample_index"]); evs["rate"].append(e["rate_kbps"])
                evs["length"].append(e["length"]); evs["crc"].append(e["crc32"])
    np.savez_compressed(os.path.join(OUT, name), ev_count=np.array(evs["count"], np.int32),
                        ev_error=np.array(evs["error"], np.uint32), ev_position=np.array(evs["position"], np.uint32),
                        ev_rate=np.array(evs["rate"], np.uint32), ev_length=np.array(evs["length"], np.uint32),
                        ev_crc=np.array(evs["crc"], np.uint32), **b)


CCK_CASES = [(5500, 1), (5500, 30), (5500, 200), (11000, 2), (11000, 77), (11000, 400)]


def main():
    if sys.argv[1:] == ["11b_cck"]:
        G = ReferenceGraph(); assert G.available()
        record_11b(G, "refgraph_11b_cck.npz", CCK_CASES, 1103)
        return
    O = Oracle(); R = Reference()
    assert R.available(), "build oracle/_ref first (oracle/build_ref.sh)"
    dump = os.path.join(REF, "kernel", "test-data", "fsample-6.dmp")
    raw = open(dump, "rb").read()
    iq = O.load_dump(dump, raw14=True)
    assert (iq % 256 == 0).all()
    np.savez_compressed(os.path.join(OUT, "fsample6_40mhz_i8.npz"), iq_i8=(iq >> 8).astype(np.int8),
                        dump_sha256=hashlib.sha256(raw).hexdigest())

    rng = np.random.default_rng(20260925)
    v = {}
    amps = [32767, 20000, 8000, 500, 30]
    x64 = np.stack([rng.integers(-amps[i % 5], amps[i % 5] + 1, size=(64, 2)) for i in range(40)]).astype(np.int16)
    x64[3, 7] = (-32768, 32767); x64[4, 0] = (32767, -32768)
    v["fft64_in"] = x64
    v["fft64_out"] = np.stack([R.fft(x, 64) for x in x64])
    v["ifft64_out"] = np.stack([R.fft(x, 64, inverse=True) for x in x64])
    x128 = np.stack([rng.integers(-amps[i % 5], amps[i % 5] + 1, size=(128, 2)) for i in range(20)]).astype(np.int16)
    v["fft128_in"] = x128
    v["ifft128_out"] = np.stack([R.fft(x, 128, inverse=True) for x in x128])
    v["fft128_out"] = np.stack([R.fft(x, 128) for x in x128])
    a = rng.integers(-32768, 32768, size=(64, 4, 2)).astype(np.int16); b = rng.integers(-32768, 32768, size=(64, 4, 2)).astype(np.int16)
    a[0, 0] = (-32768, -32768); b[0, 0] = (-32768, -32768)
    v["mul_a"] = a; v["mul_b"] = b
    v["mul_q15"] = np.stack([R.vcs2("ref_vcs_mul", x, y) for x, y in zip(a, b)])
    dm = rng.integers(-2600, 2600, size=(16, 64, 2)).astype(np.int16)
    v["demap_in"] = dm
    for nb in (1, 2, 4, 6):
        v["demap_out_%d" % nb] = np.stack([R.demap(nb, x) for x in dm])
    sig = rng.integers(0, 8, size=(32, 48)).astype(np.uint8)
    v["vsig_in"] = sig
    v["vsig_out"] = np.array([R.viterbi_sig(s) for s in sig], np.uint32)
    # noisy frames through the reference TViterbiCore with the T11aViterbi schedule
    for name, cr, per in (("12", CR_12, 2), ("23", CR_23, 3), ("34", CR_34, 4)):
        flen = 300
        nsteps_in = {CR_12: 2, CR_23: 1.5, CR_34: 4 / 3}[cr]
        nsoft = int(np.ceil((flen * 8 + 16 + 6 + 40) * nsteps_in / per) * per)
        soft = rng.integers(0, 8, size=nsoft).astype(np.uint8)          # pure noise: worst case for tie-breaks/wrap
        # half of the vector: a plausible code word with moderate noise
        half = nsoft // 2
        soft[:half] = np.clip(rng.choice([0, 7], size=half) + rng.integers(-3, 4, size=half), 0, 7)
        v["vit%s_in" % name] = soft
        v["vit%s_out" % name] = R.viterbi_frame(soft, cr, flen)
        v["vit%s_len" % name] = np.array([flen])
    # the reference's Down44to40 driven as TDownSample44_40 does (drawn last: the vectors above keep their values)
    r44 = rng.integers(-32768, 32768, size=(28 * 45, 2)).astype(np.int16)
    r44[5] = (-32768, 32767); r44[16] = (32767, -32768)
    v["down44_in"] = r44
    v["down44_out"] = R.down44to40(r44)
    np.savez_compressed(os.path.join(OUT, "ref_vectors.npz"), **v)

    # events of the reference's own receive graph on seeded random captures
    G = ReferenceGraph()
    assert G.available(), "build oracle/_ref first (oracle/build_ref.sh)"
    seed, ncap = 20260926, 400
    rng = np.random.default_rng(seed)
    ev = {"capture": [], "error": [], "position": [], "crc32": [], "sha": []}
    for i in range(ncap):
        for e in G.rx11a(random_capture(O, rng, 40)):
            ev["capture"].append(i); ev["error"].append(e["error_code"]); ev["position"].append(e["sample_index"])
            ev["crc32"].append(e["crc32"]); ev["sha"].append(np.frombuffer(hashlib.sha256(e["mpdu"]).digest()[:8], np.uint8))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_vs_refgraph import TX_CASES, _tx_payload               # the reference modulator's output for the listed frames
    tx = [G.tx11a(_tx_payload(rate, ln), rate, seed=sd) for rate, ln, sd in TX_CASES]
    np.savez_compressed(os.path.join(OUT, "refgraph_events.npz"), seed=seed, captures=ncap,
                        tx_len=np.array([len(x) for x in tx], np.int32),
                        tx_sha=np.stack([np.frombuffer(hashlib.sha256(x.tobytes()).digest()[:8], np.uint8) for x in tx]),
                        ev_capture=np.array(ev["capture"], np.int32), ev_error=np.array(ev["error"], np.uint32),
                        ev_position=np.array(ev["position"], np.uint32), ev_crc32=np.array(ev["crc32"], np.uint32),
                        ev_mpdu_sha=np.stack(ev["sha"]))
    # 802.11b: what the reference's modulation graph emits for a few frames, and what its receive graph reports for the
    # captures tests/test_oracle_11b.channel_11b makes of them
    record_11b(G, "refgraph_11b.npz", [(1000, 1), (1000, 14), (1000, 40), (2000, 5), (2000, 60), (2000, 200)], 1102)
    record_11b(G, "refgraph_11b_cck.npz", CCK_CASES, 1103)
    # 802.11n stage bricks: one burst each through the reference's own T11nDemap* / T11nDeinterleave*_S{0,1}
    rng = np.random.default_rng(1111)
    n11 = {}
    sym = np.stack([rng.integers(-a, a + 1, size=(64, 2)) for a in (100, 150, 300, 3000, 32767) for _ in range(4)]).astype(np.int16)
    sym[0, 1] = (127, -128); sym[0, 2] = (128, -129); sym[1, 40] = (-32768, 32767)
    n11["demap_in"] = sym
    for nb in (1, 2, 4, 6):
        n11["demap_out_%d" % nb] = np.stack([G.demap11n(nb, x) for x in sym])
        soft = rng.integers(0, 8, size=(6, 52 * nb)).astype(np.uint8)
        n11["deint_in_%d" % nb] = soft
        for st in (0, 1):
            n11["deint_out_%d_%d" % (nb, st)] = np.stack([G.deinterleave11n(nb, st, x) for x in soft])
    # TMimoChannelEst / TMimoChannelComp: random, singular and zero channels
    l0 = np.stack([rng.integers(-a, a + 1, size=(128, 2)) for a in (30, 400, 3000, 32767) for _ in range(3)]).astype(np.int16)
    l1 = np.stack([rng.integers(-a, a + 1, size=(128, 2)) for a in (30, 400, 3000, 32767) for _ in range(3)]).astype(np.int16)
    l1[1] = l0[1]; l0[2, 5] = 0; l0[2, 69] = 0; l1[2, 5] = 0; l1[2, 69] = 0
    est = [G.mimo_est11n(a, b) for a, b in zip(l0, l1)]
    n11["mimo_ltf0"] = l0; n11["mimo_ltf1"] = l1
    n11["mimo_h"] = np.stack([e[0] for e in est]); n11["mimo_hinv"] = np.stack([e[1] for e in est])
    y0 = rng.integers(-3000, 3001, size=(len(l0), 64, 2)).astype(np.int16); y1 = rng.integers(-3000, 3001, size=(len(l0), 64, 2)).astype(np.int16)
    cmp = [G.mimo_comp11n(e[1], a, b) for e, a, b in zip(est, y0, y1)]
    n11["mimo_y0"] = y0; n11["mimo_y1"] = y1
    n11["mimo_x0"] = np.stack([c[0] for c in cmp]); n11["mimo_x1"] = np.stack([c[1] for c in cmp])
    # TFreqEstimator_11n / TFreqComp_11n / TPilotTrack_11n (dsp_math tables generated with this image's libm)
    ll0 = rng.integers(-3000, 3001, size=(8, 128, 2)).astype(np.int16); ll1 = rng.integers(-3000, 3001, size=(8, 128, 2)).astype(np.int16)
    for l in (ll0, ll1):
        for f in range(8):
            z = (l[f, :64, 0] + 1j * l[f, :64, 1]) * np.exp(1j * (f - 4) * 0.07)
            l[f, 64:, 0] = np.rint(z.real); l[f, 64:, 1] = np.rint(z.imag)
    n11["cfo_l0"] = ll0; n11["cfo_l1"] = ll1
    n11["cfo_state"] = np.stack([G.cfo_est11n(a, b) for a, b in zip(ll0, ll1)])
    fin0 = rng.integers(-20000, 20001, size=(8, 160, 2)).astype(np.int16); fin1 = rng.integers(-20000, 20001, size=(8, 160, 2)).astype(np.int16)
    st_in = n11["cfo_state"].copy(); st_in[:, 16:] = rng.integers(-4000, 4000, size=(8, 1))
    fc = [G.freq_comp11n(st, a, b) for st, a, b in zip(st_in, fin0, fin1)]
    n11["fc_state_in"] = st_in; n11["fc_in0"] = fin0; n11["fc_in1"] = fin1
    n11["fc_state_out"] = np.stack([c[0] for c in fc]); n11["fc_out0"] = np.stack([c[1] for c in fc]); n11["fc_out1"] = np.stack([c[2] for c in fc])
    px0 = rng.integers(-6000, 6001, size=(40, 64, 2)).astype(np.int16); px1 = rng.integers(-6000, 6001, size=(40, 64, 2)).astype(np.int16)
    th = np.full(8, 123, np.int16); ths = []
    for a, b in zip(px0, px1):
        th = G.pilot_track11n(th, a, b); ths.append(th.copy())
    n11["pt_x0"] = px0; n11["pt_x1"] = px1; n11["pt_theta"] = np.stack(ths)
    # the legacy part of the preamble: TSisoChannelEst, TSisoChannelComp -> TMrcCombine, T11nSigDemap
    sl0 = np.stack([rng.integers(-a, a + 1, size=(128, 2)) for a in (20, 300, 3000, 32767) for _ in range(3)]).astype(np.int16)
    sl1 = np.stack([rng.integers(-a, a + 1, size=(128, 2)) for a in (20, 300, 3000, 32767) for _ in range(3)]).astype(np.int16)
    sl0[0, 3] = 0; sl0[0, 67] = 0; sl1[4, 9] = (-32768, -32768); sl0[5, 70] = (32767, -32768)
    sch = np.stack([G.siso_est11n(a, b) for a, b in zip(sl0, sl1)])
    sy0 = rng.integers(-3000, 3001, size=(len(sl0), 64, 2)).astype(np.int16); sy1 = rng.integers(-3000, 3001, size=(len(sl0), 64, 2)).astype(np.int16)
    sc = [G.siso_comp11n(c, a, b) for c, a, b in zip(sch, sy0, sy1)]
    n11["siso_l0"] = sl0; n11["siso_l1"] = sl1; n11["siso_ch"] = sch; n11["siso_y0"] = sy0; n11["siso_y1"] = sy1
    n11["siso_x0"] = np.stack([c[0] for c in sc]); n11["siso_x1"] = np.stack([c[1] for c in sc]); n11["siso_mrc"] = np.stack([c[2] for c in sc])
    sig = rng.integers(-300, 301, size=(10, 3, 64, 2)).astype(np.int16)
    n11["sig_sym"] = sig; n11["sig_soft"] = np.stack([G.sig_demap11n(x) for x in sig])
    # T11aDeinterleaveBPSK x3 -> T11nViterbiSig -> T11nSigParser
    from gpu_util import htsig_cases
    hs = htsig_cases(2024, 120)
    dec = [G.sig_decode11n(x) for x in hs]
    n11["sigdec_soft"] = hs; n11["sigdec_ok"] = np.array([d[0] for d in dec], np.int32)
    n11["sigdec_bytes"] = np.stack([d[1] for d in dec]); n11["sigdec_fields"] = np.stack([d[2] for d in dec])
    np.savez_compressed(os.path.join(OUT, "ref_vectors_11n.npz"), **n11)
    # 802.11n 2x2 receive graph: four transmit waveforms of the reference modulator, captures built from them with gpu_util.capture_11n,
    # the events of the reference receive graph, and TCCA11n's detections on the decimated streams
    from gpu_util import capture_11n
    rng = np.random.default_rng(1144)
    g11 = {}; frames = []
    for i, (mcs, ln) in enumerate(((8, 40), (9, 90), (10, 150), (12, 30))):
        mp = rng.integers(0, 256, ln).astype(np.uint8)
        s0, s1 = G.tx11n(mp.tobytes(), mcs)
        g11["tx%d_0" % i] = s0; g11["tx%d_1" % i] = s1; g11["mpdu%d" % i] = mp; frames.append((s0, s1))
    plan = [((0,), None, 20), ((1, 2), None, 40), ((3, 0, 1), None, 10), ((2,), 0.97, 20), ((0, 1), 0.6, 30), ((1,), 0.35, 20), ((2, 2, 0), None, 150), ((0,), 0.995, 5)]
    g11["plan_frames"] = np.array([",".join(map(str, p[0])) for p in plan]); g11["plan_cut"] = np.array([-1.0 if p[1] is None else p[1] for p in plan])
    g11["plan_sigma"] = np.array([p[2] for p in plan], np.float64)
    ev = {"count": [], "err": [], "mcs": [], "length": [], "crc": [], "pos": []}; det = []
    rng = np.random.default_rng(1145)
    for fr, cut, sg in plan:
        a, b = capture_11n(rng, [frames[i] for i in fr], sigma=sg, cut=cut)
        e = G.rx11n(a, b)
        ev["count"].append(len(e))
        for x in e:
            ev["pos"].append(x["sample_index"]); ev["err"].append(x["error_code"]); ev["mcs"].append(x["rate_kbps"]); ev["length"].append(x["length"]); ev["crc"].append(x["crc32"])
        n4 = len(a) // 2 // 4 * 4
        d = G.cca11n(a[::2][:n4], b[::2][:n4], skip=0); det.append(",".join(map(str, d)))
    g11["ev_count"] = np.array(ev["count"], np.int32); g11["ev_err"] = np.array(ev["err"], np.uint32); g11["ev_mcs"] = np.array(ev["mcs"], np.uint32)
    g11["ev_pos"] = np.array(ev["pos"], np.uint32); g11["ev_length"] = np.array(ev["length"], np.uint32); g11["ev_crc"] = np.array(ev["crc"], np.uint32); g11["cca_detect"] = np.array(det)
    np.savez_compressed(os.path.join(OUT, "refgraph_11n.npz"), **g11)
    print("written", os.listdir(OUT))


if __name__ == "__main__":
    main()
