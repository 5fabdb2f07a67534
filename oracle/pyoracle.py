"""ctypes bindings for the CPU oracle (oracle/_build/libsora_oracle.so) and, when present, the compiled
reference kernels (oracle/_ref/libsora_ref.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libsora_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsora_ref.so")
REFGRAPH_SO = os.path.join(HERE, "_ref", "libsora_refgraph.so")
REFLEGACY_SO = os.path.join(HERE, "_ref", "libsora_reflegacy.so")

E_SUCCESS = 0x0
E_FRAME_OK = 0x1
E_PLCP_HEADER_FAIL = 0x80000005
E_CRC32_FAIL = 0x80000006
CR_12, CR_23, CR_34 = 0, 1, 2
RATES = (6000, 9000, 12000, 18000, 24000, 36000, 48000, 54000)


def rate_params(kbps):
    """-> (nbpsc, code_rate, ndbps)   (ieee80211a_cmn.h:65-149)"""
    return {6000: (1, CR_12, 24), 9000: (1, CR_34, 36), 12000: (2, CR_12, 48), 18000: (2, CR_34, 72),
            24000: (4, CR_12, 96), 36000: (4, CR_34, 144), 48000: (6, CR_23, 192), 54000: (6, CR_34, 216)}[kbps]


class FrameResult(ctypes.Structure):
    _fields_ = [("start_sample", ctypes.c_uint32), ("end_sample", ctypes.c_uint32), ("error_code", ctypes.c_uint32),
                ("rate_kbps", ctypes.c_uint32), ("length", ctypes.c_uint16), ("nsym", ctypes.c_uint16),
                ("crc32", ctypes.c_uint32), ("cfo_est", ctypes.c_int16), ("reserved", ctypes.c_uint16),
                ("mpdu_offset", ctypes.c_uint32)]


class RxCtx(ctypes.Structure):
    _fields_ = [("CFO_est", ctypes.c_int16), ("FreqCoeffs", ctypes.c_int16 * 128), ("ChannelCoeffs", ctypes.c_int16 * 128),
                ("CFO_comp", ctypes.c_int16), ("SFO_comp", ctypes.c_int16), ("CompCoeffs", ctypes.c_int16 * 128),
                ("CFO_tracker", ctypes.c_int16), ("SFO_tracker", ctypes.c_int16), ("symbol_count", ctypes.c_uint32)]


class Trace(ctypes.Structure):
    _fields_ = [("ctx_after_lts", ctypes.POINTER(RxCtx)), ("eq", ctypes.c_void_p), ("tracked", ctypes.c_void_p),
                ("soft", ctypes.c_void_p), ("decoded", ctypes.c_void_p), ("cap_syms", ctypes.c_uint32),
                ("cap_soft", ctypes.c_uint32), ("n_syms", ctypes.c_uint32), ("n_soft", ctypes.c_uint32)]


def build(force=False):
    """Compile the C restatement (and oracle/_ref when the reference tree is present)."""
    if force or not os.path.exists(ORACLE_SO) or any(
            os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(ORACLE_SO)
            for f in os.listdir(HERE) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-s", "-C", HERE])
    ref_root = os.environ.get("SORA_REFERENCE", "/root/reference")
    if os.path.isdir(os.path.join(ref_root, "kernel", "core", "inc")):
        srcs = [os.path.join(HERE, f) for f in ("ref_shim.cpp", "ref_graph_shim.cpp", "ref_graph_mt_shim.cpp", "ref_graph_hip_shim.cpp", "ref_legacy_shim.cpp", "ref_legacy_pre.h", "ref_legacy_rxstream.h",
                                                "ref_flatten.py", "ref_compat.h", "build_ref.sh")]
        if force or any(not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in srcs)
                        for so in (REF_SO, REFGRAPH_SO, os.path.join(os.path.dirname(REFGRAPH_SO), "libsora_refgraph_mt.so"), os.path.join(os.path.dirname(REFGRAPH_SO), "libsora_refgraph_hip.so"), REFLEGACY_SO)):
            subprocess.check_call(["bash", os.path.join(HERE, "build_ref.sh")])


_P = lambda a: a.ctypes.data_as(ctypes.c_void_p)


class Oracle:
    def __init__(self):
        build()
        L = self.L = ctypes.CDLL(ORACLE_SO)
        L.so_init()
        L.so_crc32.restype = ctypes.c_uint32
        L.so_uatan2.restype = ctypes.c_int16
        L.so_viterbi_sig.restype = ctypes.c_uint32
        L.so_desc_sink.restype = ctypes.c_uint32
        for n in ("so_usin_lut", "so_ucos_lut", "so_uatan2_lut", "so_twiddle"):
            getattr(L, n).restype = ctypes.POINTER(ctypes.c_int16)
        L.so_demap_lut.restype = ctypes.POINTER(ctypes.c_uint8)
        L.so_sts_pattern.restype = ctypes.POINTER(ctypes.c_int16)

    # ---- LUTs
    def usin_lut(self): return np.ctypeslib.as_array(self.L.so_usin_lut(), shape=(65536,)).copy()
    def ucos_lut(self): return np.ctypeslib.as_array(self.L.so_ucos_lut(), shape=(65536,)).copy()
    def uatan2_lut(self): return np.ctypeslib.as_array(self.L.so_uatan2_lut(), shape=(65536,)).copy()
    def demap_lut(self, w): return np.ctypeslib.as_array(self.L.so_demap_lut(w), shape=(256,)).copy()
    def twiddle(self, n, k): return np.ctypeslib.as_array(self.L.so_twiddle(n, k), shape=(n // 4, 2)).copy()
    def sts_pattern(self): return np.ctypeslib.as_array(self.L.so_sts_pattern(), shape=(16, 16, 2)).copy()

    # ---- primitives
    def fft(self, x, n=64, inverse=False):
        x = np.ascontiguousarray(x, np.int16).reshape(n, 2); o = np.zeros_like(x)
        getattr(self.L, "so_%sfft%d" % ("i" if inverse else "", n))(_P(x), _P(o))
        return o

    def crc32(self, b):
        a = np.frombuffer(bytes(b), np.uint8)
        return self.L.so_crc32(_P(a), len(a))

    # ---- stages
    def new_ctx(self):
        c = RxCtx(); self.L.so_rx11a_ctx_reset(ctypes.byref(c)); return c

    def lts(self, ctx, in144):
        a = np.ascontiguousarray(in144, np.int16).reshape(144, 2)
        self.L.so_lts(ctypes.byref(ctx), _P(a))

    def sym_front(self, ctx, in80):
        a = np.ascontiguousarray(in80, np.int16).reshape(80, 2); o = np.zeros((64, 2), np.int16)
        self.L.so_sym_front(ctypes.byref(ctx), _P(a), _P(o)); return o

    def sym_track(self, ctx, eq):
        a = np.ascontiguousarray(eq, np.int16).reshape(64, 2); o = np.zeros((64, 2), np.int16)
        self.L.so_sym_track(ctypes.byref(ctx), _P(a), _P(o)); return o

    def demap(self, nbpsc, x):
        a = np.ascontiguousarray(x, np.int16).reshape(64, 2); o = np.zeros(48 * nbpsc, np.uint8)
        self.L.so_demap(nbpsc, _P(a), _P(o)); return o

    def deinterleave(self, nbpsc, s):
        a = np.ascontiguousarray(s, np.uint8); o = np.zeros(48 * nbpsc, np.uint8)
        self.L.so_deinterleave(nbpsc, _P(a), _P(o)); return o

    def viterbi_sig(self, soft48):
        a = np.ascontiguousarray(soft48, np.uint8); return self.L.so_viterbi_sig(_P(a))

    def viterbi_frame(self, soft, code_rate, frame_length):
        a = np.ascontiguousarray(soft, np.uint8); o = np.zeros(frame_length + 64, np.uint8)
        n = self.L.so_viterbi_frame(_P(a), len(a), code_rate, frame_length, _P(o)); return o[:n]

    def viterbi_frame_ex(self, soft, code_rate, frame_length, depth, lookahead):
        """T11aViterbi with another window schedule (the 802.11n graph's 192 / 36)."""
        a = np.ascontiguousarray(soft, np.uint8); o = np.zeros(frame_length + 64, np.uint8)
        n = self.L.so_viterbi_frame_ex(_P(a), len(a), code_rate, frame_length, _P(o), depth, lookahead); return o[:n]

    def desc_sink(self, dec, frame_length):
        a = np.ascontiguousarray(dec, np.uint8); o = np.zeros(frame_length, np.uint8); crc = ctypes.c_uint32(0)
        e = self.L.so_desc_sink(_P(a), frame_length, _P(o), ctypes.byref(crc)); return e, o, crc.value

    # ---- whole capture
    def rx_capture(self, iq, sample_rate_mhz=40, max_frames=64, trace=False):
        """iq: int16 [n,2].  Returns (list of dict, optional trace dict)."""
        iq = np.ascontiguousarray(iq, np.int16).reshape(-1, 2)
        res = (FrameResult * max_frames)(); mp = np.zeros(max_frames * 2504, np.uint8)
        tr = None; keep = None
        if trace:
            cap = 1400
            ctx = RxCtx(); eq = np.zeros((cap, 64, 2), np.int16); trk = np.zeros((cap, 64, 2), np.int16)
            soft = np.zeros(cap * 288, np.uint8); dec = np.zeros(2600, np.uint8)
            tr = Trace(ctypes.pointer(ctx), eq.ctypes.data, trk.ctypes.data, soft.ctypes.data, dec.ctypes.data, cap, soft.size, 0, 0)
            keep = (ctx, eq, trk, soft, dec)
        n = self.L.so_rx11a_capture(_P(iq), len(iq), sample_rate_mhz, res, max_frames, _P(mp), mp.size,
                                    ctypes.byref(tr) if tr is not None else None)
        out = []
        for i in range(n):
            r = res[i]
            d = {f: getattr(r, f) for f, _ in FrameResult._fields_}
            d["mpdu"] = mp[r.mpdu_offset:r.mpdu_offset + r.length].tobytes() if r.error_code in (E_FRAME_OK, E_CRC32_FAIL) else b""
            out.append(d)
        if trace:
            ctx, eq, trk, soft, dec = keep
            return out, {"ctx": ctx, "eq": eq[:tr.n_syms], "tracked": trk[:tr.n_syms], "soft": soft[:tr.n_soft], "decoded": dec}
        return out

    def demap11n(self, nbpsc, sym64):
        a = np.ascontiguousarray(sym64, np.int16).reshape(64, 2); o = np.zeros(52 * nbpsc, np.uint8)
        assert self.L.so_demap11n(nbpsc, _P(a), _P(o)) == 52 * nbpsc; return o

    def deinterleave11n(self, nbpsc, stream, soft):
        a = np.ascontiguousarray(soft, np.uint8); o = np.zeros(52 * nbpsc, np.uint8)
        assert len(a) == 52 * nbpsc and self.L.so_deinterleave11n(nbpsc, stream, _P(a), _P(o)) == 52 * nbpsc; return o

    def mimo_est11n(self, ltf0, ltf1):
        a = np.ascontiguousarray(ltf0, np.int16).reshape(128, 2); b = np.ascontiguousarray(ltf1, np.int16).reshape(128, 2)
        h = np.zeros((2, 128, 2), np.int16); hi = np.zeros((2, 128, 2), np.int16)
        self.L.so_mimo_est11n(_P(a), _P(b), _P(h), _P(hi)); return h, hi

    def mimo_comp11n(self, hinv, y0, y1):
        hi = np.ascontiguousarray(hinv, np.int16).reshape(2, 128, 2)
        a = np.ascontiguousarray(y0, np.int16).reshape(64, 2); b = np.ascontiguousarray(y1, np.int16).reshape(64, 2)
        x0 = np.zeros((64, 2), np.int16); x1 = np.zeros((64, 2), np.int16)
        self.L.so_mimo_comp11n(_P(hi), _P(a), _P(b), _P(x0), _P(x1)); return x0, x1

    def cfo_est11n(self, l0, l1):
        a = np.ascontiguousarray(l0, np.int16).reshape(128, 2); b = np.ascontiguousarray(l1, np.int16).reshape(128, 2)
        st = np.zeros(24, np.int16); self.L.so_cfo_est11n(_P(a), _P(b), _P(st)); return st

    def freq_comp11n(self, state, in0, in1):
        a = np.ascontiguousarray(in0, np.int16).reshape(-1, 2); b = np.ascontiguousarray(in1, np.int16).reshape(-1, 2)
        st = np.array(state, np.int16).copy(); o0 = np.zeros_like(a); o1 = np.zeros_like(b)
        self.L.so_freq_comp11n(_P(st), _P(a), _P(b), _P(o0), _P(o1), len(a) // 8); return st, o0, o1

    def pilot_track11n(self, theta8, x0, x1):
        th = np.array(theta8, np.int16).copy()
        a = np.ascontiguousarray(x0, np.int16).reshape(64, 2); b = np.ascontiguousarray(x1, np.int16).reshape(64, 2)
        self.L.so_pilot_track11n(_P(th), _P(a), _P(b)); return th

    def siso_est11n(self, l0, l1):
        a = np.ascontiguousarray(l0, np.int16).reshape(128, 2); b = np.ascontiguousarray(l1, np.int16).reshape(128, 2)
        ch = np.zeros((2, 64, 2), np.int16); self.L.so_siso_est11n(_P(a), _P(b), _P(ch)); return ch

    def siso_comp11n(self, ch, y0, y1):
        """TSisoChannelComp then TMrcCombine -> (x0, x1, mrc)"""
        c = np.ascontiguousarray(ch, np.int16).reshape(2, 64, 2)
        a = np.ascontiguousarray(y0, np.int16).reshape(64, 2); b = np.ascontiguousarray(y1, np.int16).reshape(64, 2)
        x0 = np.zeros((64, 2), np.int16); x1 = np.zeros((64, 2), np.int16); m = np.zeros((64, 2), np.int16)
        self.L.so_siso_comp11n(_P(c), _P(a), _P(b), _P(x0), _P(x1)); self.L.so_mrc11n(_P(x0), _P(x1), _P(m)); return x0, x1, m

    def sig_demap11n(self, sym3):
        a = np.ascontiguousarray(sym3, np.int16).reshape(192, 2); o = np.zeros(144, np.uint8)
        self.L.so_sig_demap11n(_P(a), _P(o)); return o

    def rx11n_capture(self, iq0, iq1, max_frames=16):
        """802.11n 2x2 receive graph over two int16 [n,2] captures @40 MHz -> list of dict (rate_kbps = MCS index)."""
        a = np.ascontiguousarray(iq0, np.int16).reshape(-1, 2); b = np.ascontiguousarray(iq1, np.int16).reshape(-1, 2)
        assert len(a) == len(b)
        res = (FrameResult * max_frames)(); mp = np.zeros(max_frames * 4096, np.uint8)
        n = self.L.so_rx11n_capture(_P(a), _P(b), len(a), res, max_frames, _P(mp), mp.size)
        out = []
        for r in res[:n]:
            d = {f: getattr(r, f) for f, _ in FrameResult._fields_}
            d["mpdu"] = mp[r.mpdu_offset:r.mpdu_offset + r.length].tobytes() if r.error_code in (E_FRAME_OK, E_CRC32_FAIL) else b""
            out.append(d)
        return out

    def cca11n(self, iq0, iq1, skip=0, max_detect=64):
        """TCCA11n over two int16 [4n,2] streams @20 MHz -> indices of the 4-sample bursts in which power was detected."""
        a = np.ascontiguousarray(iq0, np.int16).reshape(-1, 2); b = np.ascontiguousarray(iq1, np.int16).reshape(-1, 2)
        d = np.zeros(max_detect, np.uint32); n = self.L.so_cca11n(_P(a), _P(b), len(a) // 4, skip, _P(d), max_detect); return d[:min(n, max_detect)].tolist()

    def sig_decode11n(self, soft144):
        """T11aDeinterleaveBPSK x3 -> T11nViterbiSig -> T11nSigParser -> (ok, out9 bytes, fields uint32[9])"""
        a = np.ascontiguousarray(soft144, np.uint8); o9 = np.zeros(9, np.uint8); f = np.zeros(9, np.uint32)
        assert a.size == 144
        ok = self.L.so_sig_decode11n(_P(a), _P(o9), _P(f)); return ok, o9, f

    def rx11b_capture(self, iq44, max_frames=16):
        """802.11b receive graph over int16 [n,2] @44 MHz -> list of dict (end_sample = 44 MHz source position)."""
        iq = np.ascontiguousarray(iq44, np.int16).reshape(-1, 2)
        res = (FrameResult * max_frames)(); mp = np.zeros(max_frames * 4096, np.uint8)
        n = self.L.so_rx11b_capture(_P(iq), len(iq), res, max_frames, _P(mp), mp.size)
        out = []
        for r in res[:n]:
            d = {f: getattr(r, f) for f, _ in FrameResult._fields_}
            d["mpdu"] = mp[r.mpdu_offset:r.mpdu_offset + r.length].tobytes() if r.error_code in (E_FRAME_OK, E_CRC32_FAIL) else b""
            out.append(d)
        return out

    def load_dump(self, path_or_bytes, raw14=False):
        raw = np.fromfile(path_or_bytes, np.uint8) if isinstance(path_or_bytes, str) else np.frombuffer(path_or_bytes, np.uint8)
        cap = (len(raw) // 128 + 1) * 28; iq = np.zeros((cap, 2), np.int16)
        n = self.L.so_load_dump(_P(raw), len(raw), _P(iq), cap, 1 if raw14 else 0); return iq[:n]

    def down44to40(self, iq):
        """TDownSample44_40 over whole 28-sample blocks (sampling.hpp:35-66, 44MTo40M.hpp:62-123)"""
        a = np.ascontiguousarray(iq, np.int16).reshape(-1, 2); o = np.zeros((len(a) + 28, 2), np.int16)
        n = self.L.so_down44to40(_P(a), len(a), _P(o), len(o)); return o[:n]

    def downsample2(self, iq):
        a = np.ascontiguousarray(iq, np.int16).reshape(-1, 2); o = np.zeros((len(a) // 2 + 4, 2), np.int16)
        n = self.L.so_downsample2(_P(a), len(a), _P(o), len(o)); return o[:n]

    # ---- transmitter
    def tx(self, mpdu_nofcs, rate_kbps, seed=0xFF):
        """-> int8 [n,2] COMPLEX8 @40 MHz (what `demod11 -m` writes)."""
        a = np.frombuffer(bytes(mpdu_nofcs), np.uint8); cap = 640 + 160 * 1400
        o = np.zeros((cap, 2), np.int8)
        n = self.L.so_tx11a(_P(a), len(a), rate_kbps, seed, _P(o), cap)
        if n < 0: raise ValueError("so_tx11a failed %d" % n)
        return o[:n]

    def tx_capture(self, mpdu_nofcs, rate_kbps, seed=0xFF, lead=0, tail=160, rate_mhz=40):
        """TX -> `demod11 -c` expansion (<<8) -> int16 capture with `lead`/`tail` zero samples (@40 MHz)."""
        s8 = self.tx(mpdu_nofcs, rate_kbps, seed)
        x = np.zeros((lead + len(s8) + tail, 2), np.int16)
        x[lead:lead + len(s8)] = s8.astype(np.int16) << 8
        return x if rate_mhz == 40 else x[::2].copy()


class Reference:
    """The reference's own SSE kernels (oracle/_ref).  available() is False where the .so is absent."""
    def __init__(self):
        self.L = ctypes.CDLL(REF_SO) if os.path.exists(REF_SO) else None
        if self.L is not None:
            L = self.L
            L.ref_crc32.restype = ctypes.c_uint32
            L.ref_uatan2.restype = ctypes.c_int16; L.ref_usin.restype = ctypes.c_int16; L.ref_ucos.restype = ctypes.c_int16
            L.ref_viterbi_sig.restype = ctypes.c_uint32
            L.ref_vit_new.restype = ctypes.c_void_p
            for n in ("ref_usin_lut", "ref_ucos_lut", "ref_uatan2_lut"):
                getattr(L, n).restype = ctypes.POINTER(ctypes.c_int16)

    def available(self): return self.L is not None

    def fft(self, x, n=64, inverse=False):
        x = np.ascontiguousarray(x, np.int16).reshape(n, 2); o = np.zeros_like(x)
        getattr(self.L, "ref_%sfft%d" % ("i" if inverse else "", n))(_P(x), _P(o)); return o

    def lut(self, name): return np.ctypeslib.as_array(getattr(self.L, "ref_%s_lut" % name)(), shape=(65536,)).copy()

    def vcs2(self, fn, a, b, *extra):
        a = np.ascontiguousarray(a, np.int16).reshape(4, 2); b = np.ascontiguousarray(b, np.int16).reshape(4, 2)
        o = np.zeros((4, 2), np.int16); getattr(self.L, fn)(_P(a), _P(b), *extra, _P(o)); return o

    def brick64(self, which, x, coeffs):
        """one symbol through the reference's own primitives in the order of TFreqCompensation ("freq_comp"), TChannelEqualization ("channel_equalize") or
        TPhaseCompensate ("phase_comp") -- oracle/ref_shim.cpp"""
        a = np.ascontiguousarray(x, np.int16).reshape(64, 2); c = np.ascontiguousarray(coeffs, np.int16).reshape(64, 2); o = np.zeros_like(a)
        getattr(self.L, "ref_%s64" % which)(_P(a), _P(c), _P(o)); return o

    def demap(self, nbpsc, x):
        a = np.ascontiguousarray(x, np.int16).reshape(64, 2); lim = np.zeros_like(a); o = np.zeros(48 * nbpsc, np.uint8)
        self.L.ref_demap_limit64(_P(a), _P(lim)); self.L.ref_demap11a(_P(lim), nbpsc, _P(o)); return o

    def viterbi_sig(self, soft48):
        a = np.ascontiguousarray(soft48, np.uint8); return self.L.ref_viterbi_sig(_P(a))

    def down44to40(self, iq):
        """the reference's Down44to40 driven block by block as TDownSample44_40 does"""
        a = np.ascontiguousarray(iq, np.int16).reshape(-1, 2); o = np.zeros((len(a) + 28, 2), np.int16)
        n = self.L.ref_down44to40(_P(a), len(a) // 28, _P(o), len(o)); return o[:n]

    def viterbi_frame(self, soft, code_rate, frame_length):
        a = np.ascontiguousarray(soft, np.uint8); o = np.zeros(frame_length + 64, np.uint8)
        h = ctypes.c_void_p(self.L.ref_vit_new())
        n = self.L.ref_vit_decode_frame(h, _P(a), len(a), code_rate, frame_length, _P(o))
        self.L.ref_vit_free(h); return o[:n]

    def crc32(self, b):
        a = np.frombuffer(bytes(b), np.uint8); return self.L.ref_crc32(_P(a), len(a))


class RefFrame(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("error_code", "sample_index", "rate_kbps", "length", "crc32", "mpdu_offset")]


class LegacyEvent(ctypes.Structure):
    _fields_ = [("hr", ctypes.c_int32), ("rate_code", ctypes.c_uint32), ("length", ctypes.c_uint32), ("crc_ok", ctypes.c_uint32), ("block_pos", ctypes.c_uint64)]


LEGACY_RATE_KBPS = {0xB: 6000, 0xF: 9000, 0xA: 12000, 0xE: 18000, 0x9: 24000, 0xD: 36000, 0x8: 48000, 0xC: 54000}     # bba.h: DOT11A_RATE_*
BB11A_OK_FRAME, BB11A_E_CRC32 = 0x202, 0x80006004 - (1 << 32)


class ReferenceLegacy:
    """The reference's LEGACY 802.11a receiver (kernel/bb/dot11a: BB11ARxCarrierSense / BB11ARxFrameDemod + the Viterbi worker thread), compiled from its
    sources into oracle/_ref/libsora_reflegacy.so: the second cross-check oracle of SURVEY section 8 f4.  One instance per process (static context)."""
    def __init__(self):
        if not os.path.exists(REFLEGACY_SO):
            try:
                build()
            except Exception:
                pass
        self.L = ctypes.CDLL(REFLEGACY_SO) if os.path.exists(REFLEGACY_SO) else None

    def available(self): return self.L is not None

    def rx11a(self, iq40, max_frames=64):
        """CsFrameDemod (demod11a.cpp:52-185) over int16 [n,2] @40 MHz -> list of dict: hr (BB11A_OK_FRAME 0x202 / error), rate_kbps, length (incl. FCS),
        crc_ok, block (RX blocks of 28 samples consumed when the frame ended), mpdu (length bytes, for OK and CRC32 results)."""
        a = np.ascontiguousarray(iq40, np.int16).reshape(-1, 2)
        ev = (LegacyEvent * max_frames)(); fb = np.zeros(max_frames * 4096, np.uint8)
        n = self.L.ref_legacy_rx11a(_P(a), len(a) // 28 * 28, ev, max_frames, _P(fb), fb.size)
        out = []; used = 0
        for e in ev[:n]:
            d = {"hr": e.hr & 0xFFFFFFFF, "rate_kbps": LEGACY_RATE_KBPS.get(e.rate_code & 0xF, 0), "length": e.length, "crc_ok": bool(e.crc_ok), "block": e.block_pos, "mpdu": b""}
            if d["hr"] in (0x202, 0x80006004) and e.length >= 4 and used + e.length <= fb.size:
                d["mpdu"] = fb[used:used + e.length].tobytes(); used += e.length
            out.append(d)
        return out


class ReferenceGraph:
    """The reference's own BRICK graphs compiled from its sources (oracle/_ref/libsora_refgraph.so, build_ref.sh).
    NOT thread-safe and one instance per process: the reference keeps its graph context in globals."""
    def __init__(self):
        if not os.path.exists(REFGRAPH_SO):
            try:
                build()                                                  # builds oracle/_ref where the reference tree is present; a no-op elsewhere
            except Exception:
                pass
        self.L = ctypes.CDLL(REFGRAPH_SO) if os.path.exists(REFGRAPH_SO) else None

    def available(self): return self.L is not None

    def rx11b_bench(self, iq44_caps, reps=1):
        """iq44_caps: int16 [ncap, n, 2]; runs every capture `reps` times inside the library -> FRAME_OK count."""
        a = np.ascontiguousarray(iq44_caps, np.int16)
        self.L.ref_rx11b_bench.restype = ctypes.c_uint32
        return self.L.ref_rx11b_bench(_P(a), a.shape[0], a.shape[1], reps)

    def rx11a_bench(self, iq40_caps, reps=1):
        """iq40_caps: int16 [ncap, n, 2]; runs every capture `reps` times inside the library -> FRAME_OK count."""
        a = np.ascontiguousarray(iq40_caps, np.int16)
        self.L.ref_rx11a_bench.restype = ctypes.c_uint32
        return self.L.ref_rx11a_bench(_P(a), a.shape[0], a.shape[1], reps)

    def tx11a(self, mpdu_nofcs, rate_kbps, seed=0xFF):
        """The reference's preamble + modulation graphs (Test11A_FB_Mod) -> int8 [n,2] COMPLEX8 @40 MHz."""
        a = np.frombuffer(bytes(mpdu_nofcs), np.uint8); cap = 640 + 160 * 1400
        o = np.zeros((cap, 2), np.int8)
        n = self.L.ref_tx11a(_P(a), len(a), rate_kbps, seed, _P(o), cap)
        if n < 0: raise ValueError("ref_tx11a failed")
        return o[:n]

    def tx11b(self, mpdu_nofcs, rate_kbps):
        """The reference's 802.11b modulation graph (Test11B_FB_Mod) -> int8 [n,2] COMPLEX8 @44 MHz."""
        a = np.frombuffer(bytes(mpdu_nofcs), np.uint8); cap = 1 << 20
        o = np.zeros((cap, 2), np.int8)
        n = self.L.ref_tx11b(_P(a), len(a), rate_kbps, _P(o), cap)
        if n < 0: raise ValueError("ref_tx11b failed")
        return o[:n]

    def rx11b(self, iq44, max_frames=16):
        """The reference's 802.11b receive graph (Test11B_FB_Demod / MAC11b_Receive) over int16 [n,2] @44 MHz."""
        return self._events(self.L.ref_rx11b_capture, iq44, max_frames)

    def demap11n(self, nbpsc, sym64):
        """T11nDemap{BPSK,QPSK,QAM16,QAM64}: one burst through the reference's own brick."""
        a = np.ascontiguousarray(sym64, np.int16).reshape(64, 2); o = np.zeros(52 * nbpsc, np.uint8)
        assert self.L.ref_11n_demap(nbpsc, _P(a), _P(o)) == 52 * nbpsc; return o

    def deinterleave11n(self, nbpsc, stream, soft):
        """T11nDeinterleave*_S{0,1}: one burst through the reference's own brick."""
        a = np.ascontiguousarray(soft, np.uint8); o = np.zeros(52 * nbpsc, np.uint8)
        assert self.L.ref_11n_deinterleave(nbpsc, stream, _P(a), _P(o)) == 52 * nbpsc; return o

    def mimo_est11n(self, ltf0, ltf1):
        """TMimoChannelEst: one burst through the reference's own brick -> (h, hinv), each int16 [2,128,2]."""
        a = np.ascontiguousarray(ltf0, np.int16).reshape(128, 2); b = np.ascontiguousarray(ltf1, np.int16).reshape(128, 2)
        h = np.zeros((2, 128, 2), np.int16); hi = np.zeros((2, 128, 2), np.int16)
        self.L.ref_11n_mimo_est(_P(a), _P(b), _P(h), _P(hi)); return h, hi

    def mimo_comp11n(self, hinv, y0, y1):
        """TMimoChannelComp: one burst through the reference's own brick -> (x0, x1)."""
        hi = np.ascontiguousarray(hinv, np.int16).reshape(2, 128, 2)
        a = np.ascontiguousarray(y0, np.int16).reshape(64, 2); b = np.ascontiguousarray(y1, np.int16).reshape(64, 2)
        x0 = np.zeros((64, 2), np.int16); x1 = np.zeros((64, 2), np.int16)
        self.L.ref_11n_mimo_comp(_P(hi), _P(a), _P(b), _P(x0), _P(x1)); return x0, x1

    def cfo_est11n(self, l0, l1):
        """TFreqEstimator_11n through the reference's own brick -> state int16 [24] (vfo_delta_i | vfo_step_i | vfo_theta_i)."""
        a = np.ascontiguousarray(l0, np.int16).reshape(128, 2); b = np.ascontiguousarray(l1, np.int16).reshape(128, 2)
        st = np.zeros(24, np.int16); self.L.ref_11n_cfo_est(_P(a), _P(b), _P(st)); return st

    def freq_comp11n(self, state, in0, in1):
        a = np.ascontiguousarray(in0, np.int16).reshape(-1, 2); b = np.ascontiguousarray(in1, np.int16).reshape(-1, 2)
        st = np.array(state, np.int16).copy(); o0 = np.zeros_like(a); o1 = np.zeros_like(b)
        self.L.ref_11n_freq_comp(_P(st), _P(a), _P(b), _P(o0), _P(o1), len(a) // 8); return st, o0, o1

    def pilot_track11n(self, theta8, x0, x1):
        th = np.array(theta8, np.int16).copy()
        a = np.ascontiguousarray(x0, np.int16).reshape(64, 2); b = np.ascontiguousarray(x1, np.int16).reshape(64, 2)
        self.L.ref_11n_pilot_track(_P(th), _P(a), _P(b)); return th

    def siso_est11n(self, l0, l1):
        """TSisoChannelEst through the reference's own brick -> ch int16 [2,64,2] (bins 28..35, which the brick never writes, zeroed)."""
        a = np.ascontiguousarray(l0, np.int16).reshape(128, 2); b = np.ascontiguousarray(l1, np.int16).reshape(128, 2)
        ch = np.zeros((2, 64, 2), np.int16); self.L.ref_11n_siso_est(_P(a), _P(b), _P(ch)); return ch

    def siso_comp11n(self, ch, y0, y1):
        """TSisoChannelComp -> TMrcCombine through the reference's own bricks -> (x0, x1, mrc)."""
        c = np.ascontiguousarray(ch, np.int16).reshape(2, 64, 2)
        a = np.ascontiguousarray(y0, np.int16).reshape(64, 2); b = np.ascontiguousarray(y1, np.int16).reshape(64, 2)
        x0 = np.zeros((64, 2), np.int16); x1 = np.zeros((64, 2), np.int16); m = np.zeros((64, 2), np.int16)
        self.L.ref_11n_siso_comp_mrc(_P(c), _P(a), _P(b), _P(x0), _P(x1), _P(m)); return x0, x1, m

    def sig_demap11n(self, sym3):
        a = np.ascontiguousarray(sym3, np.int16).reshape(192, 2); o = np.zeros(144, np.uint8)
        self.L.ref_11n_sig_demap(_P(a), _P(o)); return o

    def sig_decode11n(self, soft144):
        """the same three bricks of the reference, one burst; the parser's context fields are zeroed first"""
        a = np.ascontiguousarray(soft144, np.uint8); o9 = np.zeros(9, np.uint8); f = np.zeros(9, np.uint32)
        assert a.size == 144
        ok = self.L.ref_11n_sig_decode(_P(a), _P(o9), _P(f)); return ok, o9, f

    def cca11n(self, iq0, iq1, skip=0, max_detect=64):
        """a fresh TCCA11n brick over two int16 [4n,2] streams @20 MHz; `skip` bursts withheld after each detection, then Reset"""
        a = np.ascontiguousarray(iq0, np.int16).reshape(-1, 2); b = np.ascontiguousarray(iq1, np.int16).reshape(-1, 2)
        d = np.zeros(max_detect, np.uint32); n = self.L.ref_11n_cca(_P(a), _P(b), len(a) // 4, skip, _P(d), max_detect); return d[:min(n, max_detect)].tolist()

    def tx11n(self, mpdu_nofcs, mcs):
        """The reference's 802.11n 2x2 modulation graphs (Test11N_FB_Mod) -> two int16 [n,2] COMPLEX16 streams @40 MHz."""
        a = np.frombuffer(bytes(mpdu_nofcs), np.uint8); cap = 1 << 20
        o0 = np.zeros((cap, 2), np.int16); o1 = np.zeros((cap, 2), np.int16)
        n = self.L.ref_tx11n(_P(a), len(a), mcs, _P(o0), _P(o1), cap)
        if n < 0: raise ValueError("ref_tx11n failed")
        return o0[:n], o1[:n]

    def rx11n(self, iq0, iq1, max_frames=16):
        """The reference's 802.11n 2x2 receive graph (Test11N_FB_Demod / RxThread) over two int16 [n,2] captures @40 MHz.
        rate_kbps carries the MCS index."""
        a = np.ascontiguousarray(iq0, np.int16).reshape(-1, 2); b = np.ascontiguousarray(iq1, np.int16).reshape(-1, 2)
        assert len(a) == len(b)
        return self._events(lambda p, n, res, mf, mp, cap: self.L.ref_rx11n_capture(p, _P(b), n, res, mf, mp, cap), a, max_frames)

    def rx11n_bench(self, iq0_caps, iq1_caps, reps=1):
        """iq*_caps: int16 [ncap, n, 2] -> decoded-OK frame count; the RxThread loop of the 11n graph runs inside the library."""
        a = np.ascontiguousarray(iq0_caps, np.int16); b = np.ascontiguousarray(iq1_caps, np.int16)
        return self.L.ref_rx11n_bench(_P(a), _P(b), a.shape[0], a.shape[1], reps)

    def rx11a_44(self, iq44, max_frames=64):
        """CreateDemodGraph11a_44M (TDownSample44_40 in front) over int16 [n,2] @44 MHz; sample_index in 44 MHz samples."""
        return self._events(self.L.ref_rx11a_capture44, iq44, max_frames)

    def rx11a_two_threads(self, iq40, max_frames=64):
        """The same graph WITH the reference's thread boundary (TThreadSeparator + a joined ViterbiThread), from
        oracle/_ref/libsora_refgraph_mt.so (a separate library: the reference keeps its graph context in globals).  None if absent."""
        p = os.path.join(os.path.dirname(REFGRAPH_SO), "libsora_refgraph_mt.so")
        if not os.path.exists(p):
            return None
        if not hasattr(self, "_mt"):
            self._mt = ctypes.CDLL(p)
        return self._events(self._mt.ref_rx11a_capture_mt, iq40, max_frames)

    def rx11a(self, iq40, max_frames=64):
        """iq40: int16 [n,2] at 40 MHz.  -> list of dict(error_code, sample_index (40 MHz source position when
        RxThread sees the event), rate_kbps, length, crc32, mpdu)."""
        return self._events(self.L.ref_rx11a_capture, iq40, max_frames)

    def _events(self, fn, iq, max_frames):
        iq = np.ascontiguousarray(iq, np.int16).reshape(-1, 2)
        res = (RefFrame * max_frames)(); mp = np.zeros(max_frames * 4096, np.uint8)
        n = fn(_P(iq), len(iq), res, max_frames, _P(mp), mp.size)
        out = []
        for r in res[:n]:
            d = {f: getattr(r, f) for f, _ in RefFrame._fields_}
            d["mpdu"] = mp[r.mpdu_offset:r.mpdu_offset + r.length].tobytes() if r.error_code in (E_FRAME_OK, E_CRC32_FAIL) else b""
            out.append(d)
        return out
