"""Shared by tools/gen_bch_golden.py and tools/fuzz_bch_vs_ref.py: GF(2^m) tables and received words CRAFTED to reach
the two places where the reference's BCH decoder throws (lib/bch.cc:359-367 -> lib/gf.h:110, and lib/bch.cc:443-444).

Both constructions prescribe the syndromes and solve a linear system over GF(2) for a bit pattern with exactly those
syndromes (the syndrome map bits -> (S_1, S_3, ..., S_2t-1) is GF(2)-linear and the all-zero word is a codeword):
  * "quadratic": S_j = power sums of the two roots of an IRREDUCIBLE x^2 + s1 x + s2 over GF(2^m) (S_1 = s1, S_2 = s1^2,
    S_j = s1 S_j-1 + s2 S_j-2): Berlekamp returns the degree-2 polynomial 1 + s1 x + s2 x^2, which has no root in the
    field, the quadratic LUT yields 0 and galois_field::inverse(0) throws;
  * "beyond_n": S_j = alpha^(j e) with n <= e < 2^m - 1, the syndromes of a single bit error at a position the
    shortened code does not have: Berlekamp returns 1 + alpha^e x and the location e >= n throws.
"""
import numpy as np


class GF:
    def __init__(self, m, prim):
        self.m, self.P = m, (1 << m) - 1
        self.antilog = np.zeros(self.P, np.int64); self.log = np.zeros(self.P + 1, np.int64)
        low, x = prim ^ (1 << m), 1
        for i in range(self.P):
            self.antilog[i] = x; self.log[x] = i
            x = ((x << 1) & self.P) ^ ((x >> (m - 1)) * low)

    def mul(self, a, b):
        return 0 if a == 0 or b == 0 else int(self.antilog[(self.log[a] + self.log[b]) % self.P])

    def trace(self, a):
        t, x = 0, a
        for _ in range(self.m):
            t ^= x; x = self.mul(x, x)
        return t  # 0 or 1


def _solve_bits(gf, n, t, S_odd, rng, ncols=None):
    """positions e in [0, n) (polynomial exponents) whose single-bit syndromes add up to S_odd[u] = S_(2u+1)."""
    m = gf.m
    rows = m * t
    ncols = ncols or rows + 64
    for _ in range(20):
        pos = rng.choice(n, ncols, replace=False)
        A = np.zeros((rows, ncols + 1), np.uint8)
        for c, e in enumerate(pos):
            for u in range(t):
                v = int(gf.antilog[((2 * u + 1) * int(e)) % gf.P])
                for b in range(m):
                    A[u * m + b, c] = (v >> b) & 1
        for u in range(t):
            for b in range(m):
                A[u * m + b, ncols] = (S_odd[u] >> b) & 1
        # Gauss-Jordan over GF(2)
        piv, r = [], 0
        for c in range(ncols):
            rr = np.nonzero(A[r:, c])[0]
            if len(rr) == 0:
                continue
            p = r + rr[0]
            if p != r:
                A[[r, p]] = A[[p, r]]
            hit = np.nonzero(A[:, c])[0]
            hit = hit[hit != r]
            A[hit] ^= A[r]
            piv.append(c); r += 1
            if r == rows:
                break
        if np.any(A[r:, ncols]):
            continue  # inconsistent with this choice of columns
        x = np.zeros(ncols, np.uint8)
        for i, c in enumerate(piv):
            x[c] = A[i, ncols]
        return sorted(int(e) for e, b in zip(pos, x) if b)
    raise RuntimeError("no solution found")


def word_from_exponents(n, exps):
    """n/8 bytes, network bit order (first bit = x^(n-1), lib/bch.cc:436-449) with the given polynomial exponents set."""
    w = np.zeros(n // 8, np.uint8)
    for e in exps:
        net = n - 1 - e
        w[net >> 3] ^= np.uint8(1 << (7 - (net & 7)))
    return w


def craft_quadratic(gf, n, t, rng):
    while True:
        s1 = int(rng.integers(1, gf.P + 1)); s2 = int(rng.integers(1, gf.P + 1))
        inv_s1sq = int(gf.antilog[(gf.P - (2 * gf.log[s1]) % gf.P) % gf.P])
        if gf.trace(gf.mul(s2, inv_s1sq)) == 1:  # x^2 + s1 x + s2 irreducible
            break
    S = [0, s1, gf.mul(s1, s1)]
    for j in range(3, 2 * t + 1):
        S.append(gf.mul(s1, S[j - 1]) ^ gf.mul(s2, S[j - 2]))
    return _solve_bits(gf, n, t, [S[2 * u + 1] for u in range(t)], rng)


def craft_beyond_n(gf, n, t, rng):
    assert n < gf.P
    e = int(rng.integers(n, gf.P))
    return _solve_bits(gf, n, t, [int(gf.antilog[((2 * u + 1) * e) % gf.P]) for u in range(t)], rng)
