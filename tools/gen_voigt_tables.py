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
