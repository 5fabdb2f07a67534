"""other formulations of the add-compare-select step, costed against the reference's arithmetic BEFORE anything is built.

The reference step (viterbicore.h:293-324): per new state n with predecessors p0 = n >> 1, p1 = p0 + 32 and 7-bit metrics u (the byte's upper seven bits),
    c0 = (u[p0] + b0) mod 128,  c1 = (u[p1] + b1) mod 128,  decision = c1 < c0 (a tie keeps branch 0),  u'[n] = min(c0, c1)
-- the comparison is between WRAPPED sums, and at rate 3/4 metrics wrap routinely (normalisation only every 24 steps).
(i)  difference form: keep d = u[p1] - u[p0] per butterfly and decide by sign(d + b1 - b0): one subtraction and one compare serve the butterfly.
(ii) radix-4: two steps at once, the minimum of four two-step sums.
Both are exact only while no candidate wraps.  This script counts, over random reachable states, how often each disagrees with the reference -- and since the
answer is "at a rate that depends only on how often sums cross 128", shows it on metric vectors taken from real soft streams at rate 3/4."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def ref_step(u, b0, b1):
    """u: [64] 7-bit metrics; b0, b1: [64] branch costs of the two predecessors of each new state (0..14).  -> (u', decisions)"""
    n = np.arange(64); p0 = n >> 1; p1 = p0 + 32
    c0 = (u[p0] + b0) & 127; c1 = (u[p1] + b1) & 127
    d = c1 < c0
    return np.where(d, c1, c0), d


def diff_step(u, b0, b1):
    """(i): decide from the metric DIFFERENCE of the butterfly (mod-128 signed), as any formulation that carries differences must"""
    n = np.arange(64); p0 = n >> 1; p1 = p0 + 32
    dd = ((u[p1] - u[p0] + b1 - b0 + 64) & 127) - 64                       # signed 7-bit difference of the candidates
    d = dd < 0
    return np.where(d, (u[p1] + b1) & 127, (u[p0] + b0) & 127), d


def radix4(u, b0a, b1a, b0b, b1b):
    """(ii): the survivor of two steps as the minimum of the four wrapped two-step sums (ties to the lower branch pair, as two reference steps would break them)"""
    u1, d1 = ref_step(u, b0a, b1a)
    u2, d2 = ref_step(u1, b0b, b1b)
    n = np.arange(64); p0 = n >> 1; p1 = p0 + 32                              # second-step predecessors
    cand = []
    for q, bb in ((p0, b0b), (p1, b1b)):
        r0 = q >> 1; r1 = r0 + 32
        cand += [((u[r0] + b0a[q] + bb) & 127), ((u[r1] + b1a[q] + bb) & 127)]
    c = np.stack(cand)                                                        # [4][64]
    return c.min(0), u2


def main():
    rng = np.random.default_rng(5)
    trials = 200000
    bad_i = bad_ii = wraps = 0
    for _ in range(trials // 64):
        base = rng.integers(0, 128)                                           # where in the 7-bit range the vector sits (normalisation puts the minimum at 0; 24 steps later it is anywhere)
        u = (base + rng.integers(0, 60, 64)) & 127                            # a spread of up to 60 between states, as the decoder's vectors have
        b = [rng.integers(0, 8, 64) * 1 + rng.integers(0, 8, 64) for _ in range(4)]
        ur, dr = ref_step(u, b[0], b[1])
        ui, di = diff_step(u, b[0], b[1])
        bad_i += int((di != dr).sum())
        n = np.arange(64); p0 = n >> 1; p1 = p0 + 32
        wraps += int(((u[p0] + b[0] >= 128) != (u[p1] + b[1] >= 128)).sum())  # exactly one candidate wrapped
        m4, u2 = radix4(u, b[0], b[1], b[2], b[3])
        bad_ii += int((m4 != u2).sum())
    print(f"{trials} state updates on vectors placed uniformly in the 7-bit range:")
    print(f"  exactly one of the two candidates wraps in {wraps} ({100.0 * wraps / trials:.1f} %)")
    print(f"  (i)  difference form decides differently from the reference in {bad_i} ({100.0 * bad_i / trials:.2f} %): where one candidate wraps, or where the candidates lie more than 63 apart")
    print(f"  (ii) min of four two-step sums differs from two reference steps in {bad_ii} ({100.0 * bad_ii / trials:.2f} %)")


if __name__ == "__main__":
    main()
