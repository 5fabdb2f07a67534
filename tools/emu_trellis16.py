#!/usr/bin/env python3
"""emu_trellis16.py -- lane-level model of the 16-lanes-per-frame-pair trellis kernel (k_viterbi16, sora_amd/csrc/k_vit16.hip).

The model executes what one 16-lane ROW of a wave executes -- four packed registers per lane, DPP exchanges as lane
permutations, the marks / guard arithmetic, banking into the survivor ring, normalisation, the window schedule and the
trace-back walk -- with the kernel's own index formulas, and compares the decoded bytes with the oracle's restatement of
T11aViterbi (oracle/so_rx11a.c).  It exists so that every piece of index algebra of the kernel is checked on a CPU before
the HIP version is written against it (there is no GPU in the development container); tests/test_trellis16_model.py runs it.

Layout.  The 64 states of a frame pair live in 16 lanes x 4 registers.  W = {0, 21, 42, 63} is a subgroup of (Z_2)^6 that the
in-place butterfly (state -> rol6(state)) maps onto itself and whose members are orthogonal to every rotation of both
generator polynomials: the four states s ^ w, w in W, therefore have IDENTICAL branch metrics at every step, and a lane
holds exactly such a coset in its four registers (register i <-> w_i).  Lane l holds the coset of v(l), v a linear map onto
span(e0..e3); the butterfly partner of state s at step t is s ^ e_j, j = 5 - t mod 6, and
    e0, e2, e4 = e0 ^ e2 ^ 21     ->  lane ^ 1, lane ^ 2, lane ^ 3 (and register ^ 1)      all quad_perm
    e1, e3, e5 = e1 ^ e3 ^ 42     ->  lane ^ 8, lane ^ 7, lane ^ 15 (and register ^ 2)     row_ror:8, row_half_mirror, row_mirror
so every exchange is ONE row-local DPP move and a register renaming: no v_permlane swaps, no in-register special case,
and the two branch-metric operands of a step (P and K - P + mark) serve all four registers.  (The model starts from the pair's soft values
as operand fields; how the kernel gets them -- 16-bit loads from the frames' packed three-bit streams, an operand table in LDS -- is the
format checked by tests/test_soft3_format.py.)
"""
import numpy as np

KFLD = (1 << 9) | (1 << 25)
KONE = 0x00020001
M32 = 0xFFFFFFFF
W = (0, 21, 42, 63)


def rol6(v, r):
    r %= 6
    return ((v << r) | (v >> (6 - r))) & 63


def rev6(x):
    return int('{:06b}'.format(x)[::-1], 2)


def parity(x):
    return bin(x).count('1') & 1


def lane_of_coords(c0, c1, c2, c3):
    return c0 ^ (c2 << 1) ^ (c1 << 3) ^ (7 if c3 else 0)


def v_of_lane(l):
    b0, b1, b2, b3 = l & 1, (l >> 1) & 1, (l >> 2) & 1, (l >> 3) & 1
    c3 = b2; c1 = b3; c0 = b0 ^ c3; c2 = b1 ^ c3
    return c0 | (c1 << 1) | (c2 << 2) | (c3 << 3)


# partner of lane l at phase ph = t mod 6 (pair bit j = 5 - ph) and the register renaming of that phase
LANE_XOR = (15, 3, 7, 2, 8, 1)       # e5, e4, e3, e2, e1, e0
REG_XOR = (2, 1, 0, 0, 0, 0)


def slot_state0(l, i):
    return v_of_lane(l) ^ W[i]


def pk_min16(a, b):
    lo = np.minimum(a & 0xFFFF, b & 0xFFFF)
    hi = np.minimum(a >> 16, b >> 16)
    return (hi << 16) | lo


def which_of(cr, ph):
    return 0 if cr == 0 else (ph & 1) if cr == 1 else ph % 3


class Row:
    """One 16-lane row = one frame pair."""

    def __init__(self, cr, win, look):
        self.cr, self.win, self.look = cr, win, look
        lanes = np.arange(16)
        self.v = np.array([v_of_lane(int(l)) for l in lanes])
        # ---- per-lane masks, per t mod 24 (MX) / t mod 6 (MY): exactly what the kernel builds at start-up
        self.MX = np.zeros((24, 16), np.uint64); self.MY = np.zeros((6, 16), np.uint64)
        self.wb = np.zeros((6, 4), int)          # role bit of register i at phase ph (wave-uniform): bit j of w_i
        for ph in range(6):
            j = 5 - ph
            for i in range(4):
                self.wb[ph][i] = (W[i] >> j) & 1
        for t in range(24):
            ph, k = t % 6, t % 8
            j = 5 - ph
            for l in range(16):
                n = rol6(int(self.v[l]), ph + 1)                              # register 0's state after the step (all four agree on the masks)
                vb = (int(self.v[l]) >> j) & 1
                ma = 7 * KFLD if parity(n & 0o155) else 0
                mb = 7 * KFLD if parity(n & 0o117) else 0
                mx = mb if which_of(cr, ph) == 2 else ma
                self.MX[t][l] = ((mx ^ (7 * KFLD)) | (KONE << k)) if vb else mx
                if t < 6:
                    self.MY[t][l] = (mb ^ (7 * KFLD)) if vb else mb
        # ---- metrics: ALL_INIT0 for state 0, ALL_INIT for the others
        self.U = np.zeros((4, 16), np.uint64)
        for i in range(4):
            for l in range(16):
                self.U[i][l] = 0 if slot_state0(l, i) == 0 else 0x18 * KFLD
        self.ring = {}                           # block -> [64] 16-bit entries indexed by rev6(state at the block's end)
        self.sidx = np.zeros((3, 4, 16), int)    # ring index of the state slot (l, i) holds at the end of block j of a row
        for jb, r in enumerate((2, 4, 0)):       # (8 jb + 8) mod 6
            for i in range(4):
                for l in range(16):
                    self.sidx[jb][i][l] = rev6(rol6(slot_state0(l, i), r))

    def dpp(self, x, ph):
        return x[np.arange(16) ^ LANE_XOR[ph]]

    def acs_step(self, which, t24, a, b, block):
        ph, k = t24 % 6, t24 % 8
        Kp = ((14 if which == 0 else 7) * KFLD + (KONE << k)) & M32
        if which == 0:
            bm = ((a ^ self.MX[t24]) + (b ^ self.MY[ph])) & M32
        elif which == 1:
            bm = a ^ self.MX[t24]
        else:
            bm = b ^ self.MX[t24]
        bo = (Kp - bm) & M32
        new = np.zeros_like(self.U)
        for i in range(4):
            X = self.U[i]
            Y = self.dpp(self.U[i ^ REG_XOR[ph]], ph)
            if self.wb[ph][i] == 0:
                new[i] = pk_min16((X + bm) & M32, (Y + bo) & M32)
            else:
                new[i] = pk_min16((X + bo) & M32, (Y + bm) & M32)
        self.U = new
        if k == 7:
            ent = np.zeros(64, np.uint64)
            for i in range(4):
                w = (self.U[i] & 0xFF) | ((self.U[i] >> 9) & 0xFF00)
                ent[self.sidx[(t24 // 8)][i]] = w
            self.ring[block] = ent
            self.U = self.U & 0xFE00FE00

    def normalize(self):
        m = self.U[0]
        for i in range(1, 4):
            m = pk_min16(m, self.U[i])
        lo = int((m & 0xFFFF).min()); hi = int((m >> 16).min())
        self.U = (self.U - ((hi << 16) | lo)) & M32


def decode_pair(softA, softB, cr, lenA, lenB, win=256, look=24):
    """Two frames of one code rate in lockstep, as the kernel runs them.  Returns (bytesA, bytesB)."""
    GB = 2 if cr == 0 else 4 if cr == 2 else 3
    GS = 1 if cr == 0 else 3 if cr == 2 else 2
    hasB = softB is not None
    n = max(len(softA), len(softB) if hasB else 0)
    sa = np.zeros(n + 64, np.uint64); sa[:len(softA)] = softA
    sb = np.zeros(n + 64, np.uint64)
    if hasB:
        sb[:len(softB)] = softB
    ops = (sa << 9) | (sb << 25)                                             # the pair stream
    R = Row(cr, win, look)
    nstepsA = len(softA) // GB * GS
    nstepsB = len(softB) // GB * GS if hasB else 0
    nsteps = max(nstepsA, nstepsB)
    tr_end = [lenA * 8 + 16 + 6, (lenB * 8 + 16 + 6) if hasB else 0]
    done = [False, not hasB]
    out = [bytearray(lenA + 2 + 64), bytearray((lenB if hasB else 0) + 2 + 64)]
    tr, ob = 0, 0

    def next_event():
        t = ob + win + look + 6
        for f in range(2):
            if not done[f]:
                t = min(t, tr_end[f])
        return t

    def trace(cnt, t24_last):
        # arg-min start state per frame with the reference's tie-break metric << 8 | state << 2, metric = 2u + last decision
        k = t24_last % 8
        j = (tr - 1) >> 3
        nn = tr - 8 * j
        for f in range(2):
            if cnt[f] == 0:
                continue
            best = None
            for i in range(4):
                for l in range(16):
                    U = int(R.U[i][l])
                    st = rol6(slot_state0(l, i), tr)
                    if k == 7:
                        w = int(R.ring[j][rev6(st)]); last = (w >> (7 + 8 * f)) & 1
                    else:
                        last = (U >> (k + 17 * f)) & 1
                    u = ((U >> (16 * f)) & 0xFFFF) >> 9
                    key = (((u << 1) | last) << 8) | (st << 2)
                    if best is None or key < best[0]:
                        best = (key, st, (U >> (17 * f)) & 0xFF)
            st = best[1]
            if nn == 8:
                H = (int(R.ring[j][rev6(st)]) >> (8 * f)) & 0xFF
            else:
                H = best[2] & ((1 << nn) - 1)
            q = rev6(((st >> nn) | rev6(H & 0x3F)) & 0x3F)
            m_lo = ob >> 3
            path = {j: H}
            for b in range(j - 1, m_lo - 1, -1):
                w = (int(R.ring[b][q]) >> (8 * f)) & 0xFF
                path[b] = w
                q = w & 63
            for m in range(m_lo, m_lo + cnt[f] // 8):
                out[f][m] = (path[m] >> 6) | ((path[m + 1] & 0x3F) << 2)

    next_thr = next_event()
    pos = 0
    while tr < nsteps and not (done[0] and done[1]):
        t24 = tr % 24
        a = ops[pos]; b = ops[pos + 1]
        R.acs_step(0, t24, a, b, tr >> 3)
        if cr != 0:
            R.acs_step(1, t24 + 1, ops[pos + 2], 0, (tr + 1) >> 3)
        if cr == 2:
            R.acs_step(2, t24 + 2, 0, ops[pos + 3], (tr + 2) >> 3)
        pos += GB
        if (t24 + GS) % 8 == 0:
            R.normalize()
        tr += GS
        if tr >= next_thr:
            partial = tr >= ob + win + look + 6
            cnt = [0, 0]
            for f in range(2):
                if not done[f]:
                    if tr >= tr_end[f]:
                        cnt[f] = tr_end[f] - ob - 6; done[f] = True
                    elif partial:
                        cnt[f] = win
            if cnt[0] | cnt[1]:
                trace(cnt, (tr - 1) % 24)
            if partial:
                ob += win
            next_thr = next_event()
    return bytes(out[0][:lenA + 2]), (bytes(out[1][:lenB + 2]) if hasB else None)


def self_check():
    """Layout algebra: the slot map is a bijection, every partner is the claimed DPP move, roles are as claimed."""
    seen = set()
    for l in range(16):
        for i in range(4):
            seen.add(slot_state0(l, i))
    assert len(seen) == 64
    slot = {slot_state0(l, i): (l, i) for l in range(16) for i in range(4)}
    for ph in range(6):
        j = 5 - ph
        for l in range(16):
            for i in range(4):
                s0 = slot_state0(l, i)
                pl, pi = slot[s0 ^ (1 << j)]
                assert pl == l ^ LANE_XOR[ph] and pi == i ^ REG_XOR[ph], (ph, l, i, pl, pi)
    for w in W:
        for r in range(6):
            assert rol6(w, r) in W
            assert parity(rol6(w, r) & 0o55) == 0 and parity(rol6(w, r) & 0o17) == 0
    return True


if __name__ == "__main__":
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle.pyoracle import Oracle
    self_check()
    o = Oracle()
    rng = np.random.default_rng(5)
    for cr in (0, 1, 2):
        per = {0: 2, 1: 3, 2: 4}[cr]
        for (LA, LB) in ((33, 100), (1, 4), (260, 257), (64, None), (300, 300)):
            def mk(L):
                steps = L * 8 + 16 + 6 + 40
                nsoft = int(np.ceil(steps * per / {0: 1, 1: 2, 2: 3}[cr] / 48.0)) * 48
                nsoft = (nsoft + per * 4 - 1) // (per * 4) * (per * 4)
                s = rng.integers(0, 8, size=nsoft).astype(np.uint8)
                h = nsoft // 2
                s[:h] = np.clip(rng.choice([0, 7], size=h) + rng.integers(-3, 4, size=h), 0, 7)
                return s
            sA = mk(LA); sB = mk(LB) if LB is not None else None
            gA, gB = decode_pair(sA, sB, cr, LA, LB if LB is not None else 0)
            wA = bytes(o.viterbi_frame(sA, cr, LA))
            assert gA == wA, ("A", cr, LA, LB)
            if LB is not None:
                assert gB == bytes(o.viterbi_frame(sB, cr, LB)), ("B", cr, LA, LB)
            print("ok", cr, LA, LB)
