#!/usr/bin/env python3
"""Derive the constants used by vamp_amd/csrc/voigt_math.hpp.

J-fraction (even contraction of the Laplace continued fraction of the Faddeeva function):

    w(z) = (i z / sqrt(pi)) * K(zeta),  zeta = z^2,
    K(zeta) = 1/(zeta - b0 - a1/(zeta - b1 - a2/(zeta - b2 - ...))),
    b_j = (4j+1)/2,  a_j = j(2j-1)/2.

Truncating after m levels gives K_m = P_{m-1}(zeta)/Q_m(zeta); the coefficients below are exact
rationals converted to double.  Also prints the Gaussian weights e^{-j^2 h^2} of the midpoint
trapezoid rule (h = 1/2) used near the real axis.
"""
from fractions import Fraction
import math


def jfrac_polys(m):
    def mul_lin(p, b):
        r = [Fraction(0)] * (len(p) + 1)
        for i, c in enumerate(p):
            r[i + 1] += c
            r[i] -= b * c
        return r

    def comb(p, q, a):
        n = max(len(p), len(q))
        r = [Fraction(0)] * n
        for i, c in enumerate(p):
            r[i] += c
        for i, c in enumerate(q):
            r[i] += a * c
        return r

    Pm2, Pm1 = [Fraction(1)], [Fraction(0)]
    Qm2, Qm1 = [Fraction(0)], [Fraction(1)]
    for j in range(1, m + 1):
        b = Fraction(4 * (j - 1) + 1, 2)
        a = Fraction(1) if j == 1 else -Fraction((j - 1) * (2 * (j - 1) - 1), 2)
        P = comb(mul_lin(Pm1, b), Pm2, a)
        Q = comb(mul_lin(Qm1, b), Qm2, a)
        Pm2, Pm1 = Pm1, P
        Qm2, Qm1 = Qm1, Q
    while Pm1 and Pm1[-1] == 0:
        Pm1 = Pm1[:-1]
    return Pm1, Qm1


if __name__ == "__main__":
    for m in (1, 2, 3, 4, 6, 8):
        P, Q = jfrac_polys(m)
        print(f"// m = {m}: P (deg {len(P)-1}) low->high, Q (deg {len(Q)-1}) low->high")
        print("P%d = {" % m + ", ".join(repr(float(c)) for c in P) + "}")
        print("Q%d = {" % m + ", ".join(repr(float(c)) for c in Q) + "}")
    print("// c_j = exp(-j^2/4), j = 1..13")
    print(", ".join("%.17e" % math.exp(-0.25 * j * j) for j in range(1, 14)))


def gh_partial_fractions(m, dps=60):
    """K_m(zeta) = sum_j c_j / (zeta - zeta_j): poles = roots of Q_m (squares of the positive
    Gauss-Hermite nodes of order 2m), residues c_j = P_{m-1}(zeta_j) / Q_m'(zeta_j) > 0."""
    import mpmath as mp
    mp.mp.dps = dps
    P, Q = jfrac_polys(m)
    Pm = [mp.mpf(c.numerator) / mp.mpf(c.denominator) for c in P]
    Qm = [mp.mpf(c.numerator) / mp.mpf(c.denominator) for c in Q]
    roots = mp.polyroots(Qm[::-1], maxsteps=500, extraprec=200)
    roots = sorted(mp.re(r) for r in roots)
    dQ = [k * Qm[k] for k in range(1, len(Qm))]
    ev = lambda c, z: sum(ck * z ** k for k, ck in enumerate(c))
    return [(z, ev(Pm, z) / ev(dQ, z)) for z in roots]


if __name__ == "__main__":
    print("// partial fractions of K_m: zeta_j, c_j  (c_j > 0, sum c_j = 1)")
    for m in (2, 3, 4, 6):
        pf = gh_partial_fractions(m)
        print("// m =", m, " sum c =", float(sum(c for _, c in pf)))
        print("Z%d = {" % m + ", ".join("%.17e" % float(z) for z, _ in pf) + "}")
        print("C%d = {" % m + ", ".join("%.17e" % float(c) for _, c in pf) + "}")


def economisation_matrix(n_keep, n_all, h=Fraction(1, 4)):
    """Chebyshev economisation of a power series sum_{n < n_all} c_n d^n on |d| <= h down to degree n_keep - 1:
    the powers d^(n_all-1) ... d^(n_keep) are replaced, highest first, by their best lower-degree stand-ins
    (d^n = h^n [T_n(d/h) / 2^(n-1) - lower powers of d/h], T_n dropped).  Returns E with
    c'_j = c_j + sum_n E[n - n_keep][j] c_n (exact rationals; entries of the other parity are 0)."""
    def cheb(n):                    # coefficients of T_n(s), low -> high
        a, b = [Fraction(1)], [Fraction(0), Fraction(1)]
        if n == 0:
            return a
        for _ in range(n - 1):
            c = [Fraction(0)] + [2 * v for v in b]
            for i, v in enumerate(a):
                c[i] -= v
            a, b = b, c
        return b
    # M[n] = the series coefficients (over d^0 .. d^(n_all-1)) that c_n = 1 stands for
    rows = []
    for n0 in range(n_keep, n_all):
        v = [Fraction(0)] * n_all
        v[n0] = Fraction(1)
        for n in range(n_all - 1, n_keep - 1, -1):
            if v[n] == 0:
                continue
            t = cheb(n)
            lead = t[n]                                   # 2^(n-1)
            cn, v[n] = v[n], Fraction(0)
            for k in range(n):
                if t[k] != 0:
                    v[k] -= cn * h ** (n - k) * t[k] / lead
        rows.append(v[:n_keep])
    return rows


if __name__ == "__main__":
    print("// TAB_ECON (fp64 tables: 18 -> 14 coefficients), by parity")
    for n, row in zip(range(14, 18), economisation_matrix(14, 18)):
        print("    {" + ", ".join("%.17e" % float(row[(n & 1) + 2 * m]) for m in range(7)) + "},")
    print("// TAB32_ECON (fp32 tables: 12 -> 8 coefficients), by parity")
    for n, row in zip(range(8, 12), economisation_matrix(8, 12)):
        print("    {" + ", ".join("%.17e" % float(row[(n & 1) + 2 * m]) for m in range(4)) + "},")
