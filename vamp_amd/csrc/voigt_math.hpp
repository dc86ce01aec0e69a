// voigt_math.hpp -- in-register evaluation of the Voigt function H(y, x) = Re w(x + i y), y >= 0.
//
// The reference evaluates its Voigt tau-profile through astropy's Voigt1D (vpfits.py:75-76),
// i.e. Re of the Faddeeva function (scipy.special.wofz in the commented form, vpfits.py:72-73).
// This file is a from-scratch fp64 evaluator shaped for a 64-wide wavefront that walks
// consecutive pixels: |z| varies smoothly along the wave, so the region tests below are nearly
// wave-uniform and each wave normally executes one branch.
//
//   |z|^2 >= 1e8          1-level J-fraction, overflow-safe closed form
//   |z|^2 >= 1e4          2-level J-fraction      w(z) = (i z / sqrt(pi)) K_m(zeta), zeta = z^2: even
//   |z|^2 >= 625          3-level                 contraction of Laplace's continued fraction = m-point
//   |z|^2 >= 196          4-level                 Gauss-Hermite quadrature of the Cauchy integral, so
//   |z|^2 >= 64           6-level                 Re w keeps its factor y.  Evaluated through the partial
//                                                 fractions of K_m: m positive real terms over one
//                                                 denominator (see GHFrac below)
//   otherwise             midpoint trapezoid rule, step h = 1/2, nodes centred on x:
//        H = A(y) e^{-x^2} cos(2xy) + (h y/pi) sum_n e^{-(x-u_n)^2} / (u_n^2 + y^2),
//        u_n = (n+1/2) h, A(y) = 2 e^{y^2} / (1 + e^{2 pi y/h})        (pole correction)
//     Centring the grid on x makes the Lorentzian denominators depend on y only, i.e. on the
//     line, not on the pixel: they are tabulated once per (walker, component) in LDS.  Both
//     terms are positive and individually accurate, so the RELATIVE error stays ~1e-14 down to
//     y -> 0 where H -> e^{-x^2}.
//   For y < 1e-9 the truncated fractions miss the (then dominant) e^{-x^2} term; it is added.
//
// Measured against 60+-digit mpmath over |z| in [1e-3, 300], y in [1e-300, 300]: max relative
// error 3e-15 in the fraction branches, 6e-15 overall (scipy.special.wofz itself: 2e-14).  Constants: tools/gen_voigt_tables.py.
//
// fp32 variant: Humlicek's W4 (JQSRT 27 (1982) 437), the four-region rational approximation.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define VAMP_DEV __device__ __forceinline__
#else
#define VAMP_DEV static inline   // host build of the same arithmetic, used by tests/ only
#endif

#ifndef VAMP_CC_FRONTLOAD
#define VAMP_CC_FRONTLOAD 1
#endif

namespace vamp {

constexpr double INV_SQRT_PI = 0.56418958354775628695;
constexpr double SQRT_PI = 1.77245385090551602730;
constexpr double PI = 3.14159265358979323846;
constexpr int CORE_J = 13;           // half-width of the node window
constexpr int DTAB_OFF = CORE_J;     // dtab[i] holds node n = i - DTAB_OFF  (n = -13 .. 30)
constexpr int DTAB_N = 44;           // entries of the per-line table 1/(u_n^2 + y^2)
constexpr double CORE_H = 0.5;
constexpr double R2_CORE = 64.0;     // below: trapezoid
constexpr double R2_M4 = 196.0;
constexpr double R2_M3 = 625.0;
constexpr double R2_M2 = 1.0e4;
constexpr double R2_M1 = 1.0e8;
constexpr double Y_POLE_MAX = 4.5;   // A(y) e^{-x^2} negligible beyond
constexpr double Y_TINY = 1.0e-9;
constexpr double X_FAR = 1.0e4;      // x >= X_FAR implies |z|^2 >= R2_M1

// All evaluators below return Hs = sqrt(pi) * H: the 1/sqrt(pi) of w(z) is folded into the
// per-line tau scale (tau_k = A y Hs), one multiply less per pixel.

// 1/d: v_rcp_f64 seed (measured relative error 4.5e-8) + two Newton steps.  (A v_rcp_f32 seed is
// ~6 cycles cheaper in isolation but needs |Q|^2 < 3e38 and gained nothing in the full kernel.)
VAMP_DEV double rcp_nr(double d) {
#if defined(__HIPCC__)
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(e, r, r);
    e = fma(-d, r, 1.0);
    r = fma(e, r, r);
    return r;
#else
    return 1.0 / d;
#endif
}

// 1/k!.  Every series below (e^{-d^2}, cosh d, sinh d, cos, sin, the exp kernel) is a Taylor
// polynomial written on these SAME literals: fp64 literals live in SGPR pairs on gfx950 (VOP3 takes
// no 64-bit immediate), the near-axis branch alone would otherwise need > 100 SGPRs of them, and
// the compiler spills what does not fit to VGPR lanes inside the pixel loop.
constexpr double F2 = 0.5, F3 = 1.6666666666666666e-01, F4 = 4.1666666666666664e-02, F5 = 8.3333333333333332e-03,
                 F6 = 1.3888888888888889e-03, F7 = 1.9841269841269841e-04, F8 = 2.4801587301587302e-05,
                 F9 = 2.7557319223985893e-06, F10 = 2.7557319223985888e-07, F11 = 2.5052108385441720e-08,
                 F12 = 2.0876756987868100e-09, F13 = 1.6059043836821613e-10, F14 = 1.1470745597729725e-11,
                 F15 = 7.6471637318198164e-13, F16 = 4.7794773323873853e-14, F17 = 2.8114572543455206e-15;

// e^a for a <= 0 (any finite a works): n = rint(a log2 e), r = a - n ln2 in two parts, degree-13
// Taylor kernel on |r| <= 0.347 (truncation 4e-18), v_ldexp.  ~1 ulp; underflows to 0 like exp.
VAMP_DEV double exp_taylor(double a) {
    a = (a < -800.0) ? -800.0 : a;                             // e^-800 = 0 in fp64; keeps -inf finite, NaN stays NaN
    const double n = rint(a * 1.4426950408889634074);
    double r = fma(-n, 6.93147180369123816490e-01, a);         // ln2, leading bits (n*hi exact)
    r = fma(-n, 1.90821492927058770002e-10, r);                // ln2 - hi
    double p = F13;
    p = fma(p, r, F12);
    p = fma(p, r, F11);
    p = fma(p, r, F10);
    p = fma(p, r, F9);
    p = fma(p, r, F8);
    p = fma(p, r, F7);
    p = fma(p, r, F6);
    p = fma(p, r, F5);
    p = fma(p, r, F4);
    p = fma(p, r, F3);
    p = fma(p, r, F2);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);                                   // NaN in: n = NaN -> p = NaN out
}

// The same with its 15 constants read from a table (EXP_TAB below, copied to LDS by the tile kernels):
// fp64 literals occupy SGPR (or VGPR) pairs for the whole pixel loop, these are read where they are used
// and gone again.  Same operations on the same constants: bit-identical to exp_taylor.
constexpr int EXP_TAB_N = 16;
#define VAMP_EXP_TAB_INIT {1.4426950408889634074, 6.93147180369123816490e-01, 1.90821492927058770002e-10, vamp::F13, vamp::F12, \
                           vamp::F11, vamp::F10, vamp::F9, vamp::F8, vamp::F7, vamp::F6, vamp::F5, vamp::F4, vamp::F3, vamp::F2, 0.0}
// DROP: leading terms left out (|r| <= ln2 / 2: r^12 / 12! = 6e-15, r^13 / 13! = 2e-16 of the value): DROP = 2 is the degree-11
// kernel, 6e-15 relative -- for the model flux of the chi^2 sweep, whose far field is good to ~1e-13 anyway
template <int DROP = 0>
VAMP_DEV double exp_taylor_tab(double a, const double* c) {
    a = (a < -800.0) ? -800.0 : a;
    const double n = rint(a * c[0]);
    double r = fma(-n, c[1], a);
    r = fma(-n, c[2], r);
    double p = c[3 + DROP];
#pragma unroll
    for (int k = 4 + DROP; k <= 14; ++k) p = fma(p, r, c[k]);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}
VAMP_DEV double exp_neg_sq_tab(double x, const double* c) {
    double s = x * x;
    double e = fma(x, x, -s);
    return exp_taylor_tab(-s, c) * (1.0 - e);
}

// exp(-x^2) with the rounding error of x*x folded back in (x up to ~27 before underflow matters)
VAMP_DEV double exp_neg_sq(double x) {
    double s = x * x;
    double e = fma(x, x, -s);
    return exp_taylor(-s) * (1.0 - e);
}

// cos(a) for |a| < ~1e3 (the near-axis rule needs |a| = 2xy <= 72): two-part pi/2 reduction and
// the minimax-free Taylor kernels on [-pi/4, pi/4]; absolute error < 2e-16.  OCML's cos costs
// ~44 DFMA issue slots (measured), this ~20.
VAMP_DEV double cos_small(double a) {
    const double k = rint(a * 0.63661977236758138243);          // 2/pi
    double r = fma(-k, 1.57079632673412561417e+00, a);          // pi/2, leading 33 bits (k*hi exact)
    r = fma(-k, 6.07710050650619224932e-11, r);                 // pi/2 - hi
    const double m = -(r * r);
    double c = F16;
    c = fma(c, m, F14);
    c = fma(c, m, F12);
    c = fma(c, m, F10);
    c = fma(c, m, F8);
    c = fma(c, m, F6);
    c = fma(c, m, F4);
    c = fma(c, m, F2);
    c = fma(c, m, 1.0);                                         // cos r
    double sn = F17;
    sn = fma(sn, m, F15);
    sn = fma(sn, m, F13);
    sn = fma(sn, m, F11);
    sn = fma(sn, m, F9);
    sn = fma(sn, m, F7);
    sn = fma(sn, m, F5);
    sn = fma(sn, m, F3);
    sn = fma(sn * m, r, r);                                     // sin r
    const int q = (int)k & 3;                                   // cos(r + q pi/2)
    const double v = (q & 1) ? sn : c;
    return (q == 1 || q == 2) ? -v : v;
}

// Per-line constants of the near-axis rule, computed once per (walker, component).
// Table entry i <-> node n = i - DTAB_OFF, u_n = (n + 1/2) h; d is even in u, so negative n need
// no special casing and a pixel reads dtab[n0 + DTAB_OFF +- j] with compile-time offsets.
// (reciprocals through rcp_nr -- v_rcp_f64 + two Newton steps, within an ulp -- and exponentials
// through exp_taylor: an IEEE fp64 divide is ~35 instructions on gfx950, and these run once per
// (walker, line, node), which is most of the work of a 40-pixel region)
VAMP_DEV double core_dtab_entry(int i, double y) {
    double u = ((i - DTAB_OFF) + 0.5) * CORE_H;
    return rcp_nr(fma(u, u, y * y));
}
VAMP_DEV double core_pole_factor(double y) {      // sqrt(pi) * A(y)
    if (!(y < Y_POLE_MAX)) return 0.0;            // (NaN: no pole term)
    const double a = (-2.0 * PI / CORE_H) * y;    // <= 0
    const double t = exp_taylor(a);
    return (SQRT_PI * 2.0) * exp_taylor(fma(y, y, a)) * rcp_nr(1.0 + t);     // y^2 - 4 pi y < 0 for y < 4.5
}
VAMP_DEV double core_hy(double y) { return (CORE_H * INV_SQRT_PI) * y; }

// ---- J-fraction tiers -------------------------------------------------------------------
// Working form of the fractions: K_m has simple real poles zeta_j > 0 (the squares of the positive
// Gauss-Hermite nodes of order 2m) with positive residues c_j, and for z = x + i y
//     -Im[ z / (z^2 - zeta_j) ] = y (|z|^2 + zeta_j) / |z^2 - zeta_j|^2 ,
// so   sqrt(pi) Re w = y sum_j c_j (r2 + zeta_j) / D_j ,  D_j = r2^2 - 2 zeta_j Re(z^2) + zeta_j^2 :
// m real Lorentzian-like terms put over one denominator (3 instructions per extra term), every
// term positive -- no complex arithmetic, no cancellation, one reciprocal.  For |z|^2 >= 64 the
// D_j are far from their zeros (zeta_j <= 15.2).  (Constants: tools/gen_voigt_tables.py.)
template <int M> struct GHFrac;
template <> struct GHFrac<2> {
    static constexpr double Z[2] = {2.75255128608410948e-01, 2.72474487139158894e+00};
    static constexpr double C[2] = {9.08248290463863017e-01, 9.17517095361369828e-02};
    static constexpr double CZ[2] = {C[0] * Z[0], C[1] * Z[1]};
};
template <> struct GHFrac<3> {
    static constexpr double Z[3] = {1.90163509193488123e-01, 1.78449274854325157e+00, 5.52534374226326008e+00};
    static constexpr double C[3] = {8.17656939112058501e-01, 1.77231492083829045e-01, 5.11156880411249310e-03};
    static constexpr double CZ[3] = {C[0] * Z[0], C[1] * Z[1], C[2] * Z[2]};
};
template <> struct GHFrac<4> {
    static constexpr double Z[4] = {1.45303521503317101e-01, 1.33909728812636142e+00, 3.92696350135828709e+00,
                                    8.58863568901203500e+00};
    static constexpr double C[4] = {7.46024515358154727e-01, 2.34479815323518026e-01, 1.92704402415765329e-02,
                                    2.25229076750735536e-04};
};
template <> struct GHFrac<6> {
    static constexpr double Z[6] = {9.87470140684811870e-02, 8.98302834569617681e-01, 2.55258980266817126e+00,
                                    5.19615253005446576e+00, 9.12424803753117963e+00, 1.51299597811080861e+01};
    static constexpr double C[6] = {6.43328723025660021e-01, 2.93934096090659958e-01, 5.82333758247283034e-02,
                                    4.40676137506639757e-03, 9.67436984518125592e-05, 2.99985433527433577e-07};
};

// An opaque zero: the constants of a branch are read through `table + opaque_zero()`, which keeps
// their scalar loads INSIDE the branch that uses them.  Left to itself the compiler hoists every
// fp64 literal of every branch in front of the pixel loop, runs out of SGPRs (VOP3 takes no 64-bit
// immediate on gfx950) and parks the overflow in VGPRs: 179 VGPRs, 2 waves per SIMD.
VAMP_DEV int opaque_zero() {
#if defined(__HIPCC__)
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return z;
#else
    return 0;
#endif
}

// num/den = sqrt(pi) Re w (num carries the factor y).  r2 = x^2 + y^2, y2 = y^2.
//   D_j = (Re z^2 - zeta_j)^2 + (Im z^2)^2,  N_j = c_j r2 + c_j zeta_j
template <int M>
VAMP_DEV void voigt_jfrac_nd(double x, double y, double r2, double& num, double& den) {
    using G = GHFrac<M>;
    // the two hot branches (2 and 3 levels: 80 % of the evaluations of a long region) keep their
    // ten constants in registers; the deeper ones reload theirs on entry
    const double* Zt = G::Z + (M >= 4 ? opaque_zero() : 0);
    const double* Ct = G::C + (M >= 4 ? opaque_zero() : 0);
    const double y2 = y * y;
    const double zr = fma(x, x, -y2);        // Re z^2
    const double zi2 = (4.0 * y2) * (zr + y2);   // (Im z^2)^2 = 4 x^2 y^2
    // N_j: one FMA with a third literal per term in the hot branches, add + multiply in the deep ones
    auto Nterm = [&](int j) {
        if constexpr (M < 4) return fma(G::C[j], r2, G::CZ[j]);
        else return Ct[j] * (r2 + Zt[j]);
    };
    double t = zr - Zt[0];
    double N = Nterm(0);
    double D = fma(t, t, zi2);
#pragma unroll
    for (int j = 1; j < M; ++j) {
        t = zr - Zt[j];
        const double Nj = Nterm(j);
        const double Dj = fma(t, t, zi2);
        N = fma(N, Dj, Nj * D);
        D = D * Dj;
    }
    num = y * N;
    den = D;
}

template <int M>
VAMP_DEV double voigt_jfrac(double x, double y, double r2) {
    double num, den;
    voigt_jfrac_nd<M>(x, y, r2, num, den);
    return num * rcp_nr(den);
}

// Two evaluations share one reciprocal: 1/(d0 d1), then 1/d0 = r d1, 1/d1 = r d0 (one v_rcp_f64
// and one Newton pair instead of two; den <= 1e96 for every lane a fraction may see, so the
// product stays far inside the fp64 range).
template <int M>
VAMP_DEV void voigt_jfrac_x2(double x0, double x1, double y, double r20, double r21, double& h0, double& h1) {
    double n0, d0, n1, d1;
    voigt_jfrac_nd<M>(x0, y, r20, n0, d0);
    voigt_jfrac_nd<M>(x1, y, r21, n1, d1);
    const double r = rcp_nr(d0 * d1);
    h0 = n0 * (r * d1);
    h1 = n1 * (r * d0);
}

// |z|^2 >= 1e8: one level, K = 1/(zeta - 1/2); written so that huge |x| cannot overflow.
VAMP_DEV double voigt_far(double x, double y, double r2) {
    const double inv = rcp_nr(r2);                 // r2 may be +inf -> 0
    const double eps = ((x * inv) * x - (y * inv) * y - 0.25 * inv) * inv;
    return y * inv * (1.0 + 0.5 * inv) * (1.0 + eps + eps * eps);
}

// ---- near-axis rule ---------------------------------------------------------------------
// dtab: core_dtab_entry table of the line, pole = core_pole_factor(y), hy = core_hy(y); x < 8.
VAMP_DEV double voigt_core(double x, double y, const double* dtab, double pole, double hy) {
    const int n0 = (int)(x * 2.0);                       // floor(x/h)
    const double d = x - (n0 + 0.5) * CORE_H;            // |d| <= h/2
    const double d2 = d * d;
    // e^{-d^2}, e^{+d}, e^{-d} by short series (|d| <= 0.25)
    const double md2 = -d2;
    double g0 = F9;
    g0 = fma(g0, md2, F8);
    g0 = fma(g0, md2, F7);
    g0 = fma(g0, md2, F6);
    g0 = fma(g0, md2, F5);
    g0 = fma(g0, md2, F4);
    g0 = fma(g0, md2, F3);
    g0 = fma(g0, md2, F2);
    g0 = fma(g0, md2, 1.0);
    g0 = fma(g0, md2, 1.0);                              // exp(-d^2)
    double ch = F14;
    ch = fma(ch, d2, F12);
    ch = fma(ch, d2, F10);
    ch = fma(ch, d2, F8);
    ch = fma(ch, d2, F6);
    ch = fma(ch, d2, F4);
    ch = fma(ch, d2, F2);
    ch = fma(ch, d2, 1.0);                               // cosh d
    double sh = F15;
    sh = fma(sh, d2, F13);
    sh = fma(sh, d2, F11);
    sh = fma(sh, d2, F9);
    sh = fma(sh, d2, F7);
    sh = fma(sh, d2, F5);
    sh = fma(sh, d2, F3);
    sh = fma(sh, d2, 1.0) * d;                           // sinh d
    const double q = ch + sh, qi = ch - sh;              // e^{d}, e^{-d}   (h = 1/2: e^{2 d h})
    const double* p = dtab + n0 + DTAB_OFF;              // centre node; +-j are immediate offsets
    // S = p0 + sum_j c_j (q^j p_j + q^-j p_-j), c_j = e^{-j^2/4} = prod_{i<=j} rho_i, rho_i = e^{-(2i-1)/4}:
    //   S+ = t_1 (p_1 + t_2 (p_2 + ... + t_13 p_13)),  t_j = q rho_j,  t_{j-1} = t_j e^{1/2}
    // two Horner chains (q and 1/q), all terms positive, 4 instructions per node pair and only
    // two literals (rho_13, e^{1/2}) instead of thirteen.
    static_assert(CORE_J == 13, "rho_13 below is e^{-(2*13-1)/4}");
    constexpr double RHO_TOP = 1.93045413622770930e-03;  // e^{-25/4}
    constexpr double SQRT_E = 1.64872127070012819e+00;   // e^{1/2}
    double tp = q * RHO_TOP, tm = qi * RHO_TOP;
    double sp = p[CORE_J], sm = p[-CORE_J];
#pragma unroll
    for (int j = CORE_J; j >= 2; --j) {
        sp = fma(sp, tp, p[j - 1]);
        sm = fma(sm, tm, p[-(j - 1)]);
        tp *= SQRT_E;
        tm *= SQRT_E;
    }
    const double S = fma(sp, tp, fma(sm, tm, p[0]));
    double H = hy * (g0 * S);
    if (pole != 0.0) H = fma(pole * exp_neg_sq(x), cos_small(2.0 * x * y), H);
    return H;
}

// ---- per-line Taylor tables of the near-axis zone ---------------------------------------------
// For one line (fixed y) sqrt(pi) Re w(x + i y) on 0 <= x < 8 is an entire function of x.  The zone
// is cut into TAB_NI intervals of width 1/2 (their centres x_i = (i + 1/2)/2 are nodes of the
// near-axis rule, so the rule needs no exponentials there: the weights are e^{-j^2/4}), and on
// each the function is its Taylor polynomial of degree TAB_NC - 1 about x_i, economised to degree TAB_NT - 1:
//     Ws = sqrt(pi) w,  Ws' = -2 z Ws + 2i,  c_0 = Ws(z_i),  c_1 = -2 z_i c_0 + 2i,
//     c_{n+1} = -2 (z_i c_n + c_{n-1}) / (n + 1),      sqrt(pi) H(x_i + d, y) = sum_n Re(c_n) d^n
// (d real, |d| <= 1/4).  Measured against 40-digit references for y from 1e-12 to 8: absolute
// error <= 2e-16, relative <= 2e-14 (tests/test_oracle.py::test_taylor_tables_host_build).  One evaluation is
// TAB_NT - 1 fused multiply-adds on coefficients read from LDS, against ~180 issue slots for the
// rule itself; a table costs one rule evaluation (real and imaginary part) per interval.
constexpr int TAB_NI = 16;            // intervals: [i/2, (i+1)/2)
constexpr int TAB_NC = 18;            // Taylor coefficients computed per interval
#ifndef VAMP_TAB_NT
#define VAMP_TAB_NT 14
#endif
constexpr int TAB_NT = VAMP_TAB_NT;   // coefficients kept per interval (112 B: 16-byte aligned rows)
constexpr int TAB_LINE = TAB_NI * TAB_NT;   // doubles per line
// Chebyshev economisation on |d| <= 1/4: the terms d^14 .. d^17 are replaced by their best lower-degree
// stand-ins (d^n = h^n [T_n(d/h) / 2^(n-1) - lower powers], T_n dropped, highest first), which folds
// c_14 .. c_17 into the coefficients of the same parity: c'_j = c_j + sum_n E[n - 14][j / 2] c_n, exact
// dyadic factors.  The degree-13 polynomial that remains differs from the degree-17 one by
// sum_n |c_n| 4^-n 2^(1-n) < 1e-17: four multiply-adds and two 16-byte reads less per evaluation.
static_assert(TAB_NT == 14 || TAB_NT == 18, "economised (14) or plain (18) Taylor rows");
constexpr double TAB_ECON[4][7] = {
    {4.54747350886464119e-13, -7.13043846189975739e-10, 1.82539224624633789e-07, -1.75237655639648438e-05,
     8.01086425781250000e-04, -1.87988281250000000e-02, 2.18750000000000000e-01},
    {3.41060513164848089e-12, -2.03726813197135925e-09, 3.52039933204650879e-07, -2.68220901489257812e-05,
     1.04904174804687500e-03, -2.19726562500000000e-02, 2.34375000000000000e-01},
    {1.06581410364015028e-13, -1.63709046319127083e-10, 4.07453626394271851e-08, -3.75509262084960938e-06,
     1.60932540893554688e-04, -3.35693359375000000e-03, 2.92968750000000000e-02},
    {8.45545855554519221e-13, -4.94765117764472961e-10, 8.31205397844314575e-08, -6.07967376708984375e-06,
     2.22921371459960938e-04, -4.15039062500000000e-03, 3.32031250000000000e-02}};

// sin and cos together, same reduction and kernels as cos_small
VAMP_DEV void sincos_small(double a, double& sn_out, double& cs_out) {
    const double k = rint(a * 0.63661977236758138243);
    double r = fma(-k, 1.57079632673412561417e+00, a);
    r = fma(-k, 6.07710050650619224932e-11, r);
    const double m = -(r * r);
    double c = F16;
    c = fma(c, m, F14);
    c = fma(c, m, F12);
    c = fma(c, m, F10);
    c = fma(c, m, F8);
    c = fma(c, m, F6);
    c = fma(c, m, F4);
    c = fma(c, m, F2);
    c = fma(c, m, 1.0);
    double sn = F17;
    sn = fma(sn, m, F15);
    sn = fma(sn, m, F13);
    sn = fma(sn, m, F11);
    sn = fma(sn, m, F9);
    sn = fma(sn, m, F7);
    sn = fma(sn, m, F5);
    sn = fma(sn, m, F3);
    sn = fma(sn * m, r, r);
    const int q = (int)k & 3;                    // angle = r + q pi/2
    const double cq = (q & 1) ? sn : c, sq = (q & 1) ? c : sn;
    cs_out = (q == 1 || q == 2) ? -cq : cq;
    sn_out = (q >= 2) ? -sq : sq;
}

// sqrt(pi) w(x_i + i y) at the centre of interval i from the line's dtab: the near-axis rule with
// d = 0, real part as in voigt_core, imaginary part from the same nodes:
//     sqrt(pi) Im w = (h / sqrt(pi)) sum_n e^{-(x - u_n)^2} u_n / (u_n^2 + y^2) - sqrt(pi) A(y) e^{-x^2} sin(2 x y)
VAMP_DEV void core_centre(int i, double y, const double* dtab, double pole, double hy, double& re, double& im) {
    // weights c_j = e^{-j^2/4}; with u_{i+-j} = x_i +- j h the imaginary sum splits into the real one
    // and an antisymmetric one:  sum_j c_j u_{i+j} p_{i+j} = x_i S0 + h S1,
    //   S0 = p_0 + sum_{j>0} c_j (p_j + p_-j),   S1 = sum_{j>0} j c_j (p_j - p_-j)
    // -- two multiply-adds per node pair, no per-node abscissae
    constexpr double CJ[CORE_J + 1] = {1.0, 0.77880078307140486825, 0.3678794411714423216, 0.10539922456186433678,
                                       0.018315638888734180294, 0.0019304541362277092422, 0.0001234098040866795495,
                                       4.7851173921290090896e-6, 1.1253517471925911451e-7, 1.6052280551856116087e-9,
                                       1.3887943864964020595e-11, 7.2877240958196924193e-14, 2.3195228302435693883e-16,
                                       4.477732441718301199e-19};
    constexpr double JCJ[CORE_J + 1] = {0.0, 7.78800783071404878e-01, 7.35758882342884668e-01, 3.16197673685592984e-01,
                                        7.32625555549367147e-02, 9.65227068113854586e-03, 7.40458824520077367e-04,
                                        3.34958217449030638e-05, 9.00281397754072932e-07, 1.44470524966705042e-08,
                                        1.38879438649640215e-10, 8.01649650540166128e-13, 2.78342739629228356e-15,
                                        5.82105217423379233e-18};
    const double* p = dtab + i + DTAB_OFF;
    // all 27 table reads first, arithmetic after: scheduled one by one next to their use (the
    // compiler's choice) each read is an exposed LDS round trip for the one or two wavefronts a SIMD
    // holds while tables are built
    double pv[2 * CORE_J + 1];
#pragma unroll
    for (int j = 0; j <= 2 * CORE_J; ++j) pv[j] = p[j - CORE_J];
#if defined(__HIPCC__) && VAMP_CC_FRONTLOAD
    __builtin_amdgcn_sched_barrier(0);
#endif
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int j = CORE_J; j >= 1; --j) {          // small terms first
        const double a = pv[CORE_J + j], b = pv[CORE_J - j];
        s0 = fma(CJ[j], a + b, s0);
        s1 = fma(JCJ[j], a - b, s1);
    }
    s0 += pv[CORE_J];
    const double x = (i + 0.5) * CORE_H;
    re = hy * s0;
    im = (CORE_H * INV_SQRT_PI) * fma(x, s0, CORE_H * s1);
    if (pole != 0.0) {
        double sn, cs;
        sincos_small(2.0 * x * y, sn, cs);
        const double e = pole * exp_taylor(-(x * x));      // x = (2 i + 1)/4: x^2 is exact
        re = fma(e, cs, re);
        im = fma(-e, sn, im);
    }
}

// the TAB_NT coefficients of interval i from sqrt(pi) w at its centre, written to out[0 .. TAB_NT)
VAMP_DEV void taylor_table_row_from_centre(int i, double y, double c0r, double c0i, double* out) {
    const double zr = (i + 0.5) * CORE_H, zi = y;
    // c_1 = -2 z c_0 + 2i
    double c1r = -2.0 * (zr * c0r - zi * c0i);
    double c1i = fma(-2.0, fma(zr, c0i, zi * c0r), 2.0);
    double cr[TAB_NC];
    cr[0] = c0r;
    cr[1] = c1r;
    // (unrolled, so that -2/(n+1) is a literal: left as a loop this was a 35-instruction fp64 divide per
    // coefficient -- more than the rest of the table row together)
#ifdef VAMP_TT_NOREC         // timing-only builds (tools/variants.py)
    for (int n = 0; n < TAB_NT; ++n) out[n] = c1i;
    return;
#endif
#pragma unroll
    for (int n = 1; n + 1 < TAB_NC; ++n) {
        const double f = -2.0 / (double)(n + 1);
        const double nr = f * (fma(zr, c1r, -zi * c1i) + c0r);
        const double ni = f * (fma(zr, c1i, zi * c1r) + c0i);
        c0r = c1r; c0i = c1i;
        c1r = nr; c1i = ni;
        cr[n + 1] = nr;
    }
#pragma unroll
    for (int n = TAB_NT; n < TAB_NC; ++n) {
#pragma unroll
        for (int m = 0; m < 7; ++m) cr[(n & 1) + 2 * m] = fma(TAB_ECON[n - 14][m], cr[n], cr[(n & 1) + 2 * m]);
    }
#ifdef VAMP_TT_NOSTORE       // timing-only builds: all the arithmetic, one store per row
    double acc = 0.0;
#pragma unroll
    for (int n = 0; n < TAB_NT; ++n) acc += cr[n];
    out[2] = acc;
#else
#pragma unroll
    for (int n = 0; n < TAB_NT; ++n) out[n] = cr[n];
#endif
}

// ---- fp32 contexts: the same tables in single precision -------------------------------------------
// Humlicek's W4 costs ~25 instructions per evaluation in its regions I / II but ~45 (III) and ~120 (IV: a
// 7/7 complex rational plus exp and cos) in the line cores, where every near (line, tile) pair of the
// headline lies: 44 % of the fp32 kernel.  The cores therefore take the fp64 path's route -- per-line
// Taylor rows on |x| < 8, built once per walker from the same centre values (core_centre) by the same
// recurrence -- truncated to what single precision can hold: TAB32_NC = 12 Taylor coefficients
// (|c_12| (1/4)^12 < 3e-9, Cauchy estimate on a circle of radius 2), economised on |d| <= 1/4 to TAB32_NT = 8
// (error < 1e-7 of the line centre), scaled by 1/sqrt(pi) so that a row returns H like W4 does, stored as
// floats: 32 B per row, one evaluation = index, two 16-byte LDS reads, 7 fp32 multiply-adds.  Against
// scipy's wofz the rows are ~1000 x closer than W4 (tests/test_oracle.py::test_taylor_tables32_host_build);
// W4 remains the evaluator of the wings (regions I / II), of short regions and of vamp_wofz_re.
constexpr int TAB32_NC = 12;
constexpr int TAB32_NT = 8;
constexpr int TAB32_LINE = TAB_NI * TAB32_NT;   // floats per line
constexpr double TAB32_ECON[4][4] = {            // tools/gen_voigt_tables.py: economisation_matrix(8, 12)
    {-1.19209289550781250e-07, 6.10351562500000000e-05, -4.88281250000000000e-03, 1.25000000000000000e-01},
    {-5.36441802978515625e-07, 1.14440917968750000e-04, -6.59179687500000000e-03, 1.40625000000000000e-01},
    {-1.67638063430786133e-08, 8.04662704467773438e-06, -5.72204589843750000e-04, 1.09863281250000000e-02},
    {-8.19563865661621094e-08, 1.63912773132324219e-05, -8.39233398437500000e-04, 1.34277343750000000e-02}};
VAMP_DEV void taylor_table_row32_from_centre(int i, double y, double c0r, double c0i, float* out) {
    const double zr = (i + 0.5) * CORE_H, zi = y;
    double c1r = -2.0 * (zr * c0r - zi * c0i);
    double c1i = fma(-2.0, fma(zr, c0i, zi * c0r), 2.0);
    double cr[TAB32_NC];
    cr[0] = c0r;
    cr[1] = c1r;
#pragma unroll
    for (int n = 1; n + 1 < TAB32_NC; ++n) {
        const double f = -2.0 / (double)(n + 1);
        const double nr = f * (fma(zr, c1r, -zi * c1i) + c0r);
        const double ni = f * (fma(zr, c1i, zi * c1r) + c0i);
        c0r = c1r; c0i = c1i;
        c1r = nr; c1i = ni;
        cr[n + 1] = nr;
    }
#pragma unroll
    for (int n = TAB32_NT; n < TAB32_NC; ++n) {
#pragma unroll
        for (int m = 0; m < 4; ++m) cr[(n & 1) + 2 * m] = fma(TAB32_ECON[n - TAB32_NT][m], cr[n], cr[(n & 1) + 2 * m]);
    }
#pragma unroll
    for (int n = 0; n < TAB32_NT; ++n) out[n] = (float)(INV_SQRT_PI * cr[n]);
}
VAMP_DEV void taylor_table_row32(int i, double y, const double* dtab, double pole, double hy, float* out) {
    double c0r, c0i;
    core_centre(i, y, dtab, pole, hy, c0r, c0i);
    taylor_table_row32_from_centre(i, y, c0r, c0i, out);
}
// H(x, y) for 0 <= x < 8 from the line's fp32 table
VAMP_DEV float taylor_table32_eval(const float* tab, float x) {
    int i = (int)(x * 2.0f);
    i = i < TAB_NI - 1 ? i : TAB_NI - 1;
    const float d = fmaf((float)i, -(float)CORE_H, x) - 0.5f * (float)CORE_H;
    const float* a = tab + i * TAB32_NT;
    float r = a[TAB32_NT - 1];
#pragma unroll
    for (int n = TAB32_NT - 2; n >= 0; --n) r = fmaf(r, d, a[n]);
    return r;
}

// the same from the line's near-axis table
VAMP_DEV void taylor_table_row(int i, double y, const double* dtab, double pole, double hy, double* out) {
    double c0r, c0i;
#ifdef VAMP_TT_NOCENTRE     // timing-only builds (tools/variants.py)
    c0r = hy + dtab[i]; c0i = pole;
#else
    core_centre(i, y, dtab, pole, hy, c0r, c0i);
#endif
    taylor_table_row_from_centre(i, y, c0r, c0i, out);
}

// sqrt(pi) H(x, y) for 0 <= x < 8 from the line's table
VAMP_DEV double taylor_table_eval(const double* tab, double x) {
    int i = (int)(x * 2.0);
    i = i < TAB_NI - 1 ? i : TAB_NI - 1;
    const double d = fma((double)i, -CORE_H, x) - 0.5 * CORE_H;
    const double* a = tab + i * TAB_NT;
    double r = a[TAB_NT - 1];
#pragma unroll
    for (int n = TAB_NT - 2; n >= 0; --n) r = fma(r, d, a[n]);
    return r;
}

// Per-point evaluator (k_model, k_wofz, host tests).  x >= 0, y >= 0; returns sqrt(pi) H.
VAMP_DEV double voigt_Hs(double x, double y, const double* dtab, double pole, double hy) {
    const double r2 = fma(x, x, y * y);
#ifdef VAMP_FORCE_TIER   // timing-only builds (tools/tier_cost.py): every evaluation takes one branch
    {
        double xx = x < 7.9 ? x : 7.9;
        if (VAMP_FORCE_TIER == 0) return voigt_core(xx, y, dtab, pole, hy);
        if (VAMP_FORCE_TIER == 1) return voigt_jfrac<6>(x, y, r2);
        if (VAMP_FORCE_TIER == 2) return voigt_jfrac<4>(x, y, r2);
        if (VAMP_FORCE_TIER == 3) return voigt_jfrac<3>(x, y, r2);
        if (VAMP_FORCE_TIER == 4) return voigt_jfrac<2>(x, y, r2);
        if (VAMP_FORCE_TIER == 5) return voigt_far(x, y, r2);
        return x * y;             // 6: no Voigt arithmetic at all (loop + staging overhead)
    }
#endif
    if (r2 < R2_CORE) return voigt_core(x, y, dtab, pole, hy);     // (NaN compares false: no table lookup)
    double H;
    if (r2 >= R2_M3) {
        if (r2 >= R2_M1) H = voigt_far(x, y, r2);
        else if (r2 >= R2_M2) H = voigt_jfrac<2>(x, y, r2);
        else H = voigt_jfrac<3>(x, y, r2);
    } else {
        if (r2 >= R2_M4) H = voigt_jfrac<4>(x, y, r2);
        else H = voigt_jfrac<6>(x, y, r2);
    }
    if (y < Y_TINY) H += SQRT_PI * exp_neg_sq(x);
    return H;
}

// unscaled H = Re w(x + i y)
VAMP_DEV double voigt_H(double x, double y, const double* dtab, double pole, double hy) {
    return INV_SQRT_PI * voigt_Hs(x, y, dtab, pole, hy);
}

// ---- fp32: Humlicek W4 --------------------------------------------------------------------
// Re w(x + i y), t = y - i x, s = |x| + y (JQSRT 27 (1982) 437).  Complex arithmetic spelled out in
// floats; quotients through v_rcp_f32 (1 ulp, far inside the method's 1e-4).
//   region I   s >= 15                       t 0.5641896 / (0.5 + t^2)
//   region II  5.5 <= s < 15                 valid (and more accurate) for every s >= 5.5
//   region III s < 5.5, y >= 0.195|x| - 0.176
//   region IV  otherwise
struct cf32 { float re, im; };
VAMP_DEV cf32 cmul(cf32 a, cf32 b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
VAMP_DEV cf32 cadd(cf32 a, float c) { return {a.re + c, a.im}; }
VAMP_DEV cf32 cscale(cf32 a, float c) { return {a.re * c, a.im * c}; }
VAMP_DEV float rcp32(float d) {
#if defined(__HIPCC__)
    return __builtin_amdgcn_rcpf(d);
#else
    return 1.0f / d;
#endif
}
VAMP_DEV float cdiv_re(cf32 n, cf32 d) {   // Re(n/d)
    return (n.re * d.re + n.im * d.im) * rcp32(d.re * d.re + d.im * d.im);
}
constexpr float W4_XMAX = 1.0e9f;          // beyond, x^4 leaves the fp32 range; H < 1e-18 y there

VAMP_DEV float w4_region1(float x, float y) {
    x = fminf(x, W4_XMAX);
    const cf32 t = {y, -x};
    const cf32 u = cmul(t, t);
    return cdiv_re(cscale(t, 0.5641896f), cadd(u, 0.5f));
}
VAMP_DEV float w4_region2(float x, float y) {
    // |d|^2 ~ |t|^8 must stay inside the fp32 range: beyond |x| + y = 1e3 region I is exact to 1e-12 of the value and
    // takes over.  (Clamping x instead -- as this function did until round 3 -- is wrong for a heavily damped line:
    // with y ~ 1e4 the profile y / (x^2 + y^2) is still falling at x = 3e4.  Callers choose the region for a whole
    // wavefront, and in the far-field node pass the lanes of a wavefront hold DIFFERENT lines, so a lane can arrive
    // here with any s >= 5.5: tests/soak_long_regions.py f32, test_fp32_heavily_damped_lines_in_the_far_field.)
    if (x + y >= 1.0e3f) return w4_region1(x, y);
    const cf32 t = {y, -x};
    const cf32 u = cmul(t, t);
    const cf32 n = cmul(t, cadd(cscale(u, 0.5641896f), 1.410474f));
    const cf32 d = cadd(cmul(u, cadd(u, 3.0f)), 0.75f);
    return cdiv_re(n, d);
}
VAMP_DEV float w4_region3(float x, float y) {
    const cf32 t = {y, -x};
    cf32 n = cadd(cscale(t, 0.5642236f), 3.778987f);
    n = cadd(cmul(n, t), 11.96482f);
    n = cadd(cmul(n, t), 20.20933f);
    n = cadd(cmul(n, t), 16.4955f);
    cf32 d = cadd(t, 6.699398f);
    d = cadd(cmul(d, t), 21.69274f);
    d = cadd(cmul(d, t), 39.27121f);
    d = cadd(cmul(d, t), 38.82363f);
    d = cadd(cmul(d, t), 16.4955f);
    return cdiv_re(n, d);
}
VAMP_DEV float w4_region4(float x, float y) {
    const cf32 t = {y, -x};
    const cf32 u = cmul(t, t);
    cf32 n = cadd(cscale(u, -0.56419f), 1.320522f);     // 1.320522 - u*0.56419
    n = cadd(cscale(cmul(u, n), -1.0f), 35.76683f);
    n = cadd(cscale(cmul(u, n), -1.0f), 219.0313f);
    n = cadd(cscale(cmul(u, n), -1.0f), 1540.787f);
    n = cadd(cscale(cmul(u, n), -1.0f), 3321.9905f);
    n = cadd(cscale(cmul(u, n), -1.0f), 36183.31f);
    n = cmul(t, n);
    cf32 d = cadd(cscale(u, -1.0f), 1.841439f);
    d = cadd(cscale(cmul(u, d), -1.0f), 61.57037f);
    d = cadd(cscale(cmul(u, d), -1.0f), 364.2191f);
    d = cadd(cscale(cmul(u, d), -1.0f), 2186.181f);
    d = cadd(cscale(cmul(u, d), -1.0f), 9022.228f);
    d = cadd(cscale(cmul(u, d), -1.0f), 24322.84f);
    d = cadd(cscale(cmul(u, d), -1.0f), 32066.6f);
    // Re exp(t^2).  u.re = y^2 - x^2 in [-31, 0.8], |u.im| = 2 x y < 10 in this region: the hardware exponential and
    // cosine (v_exp_f32, v_cos_f32: ~1e-6 absolute on this range, a 40th of the library calls' instructions) are far
    // inside the method's 1e-4
#if defined(__HIPCC__)
    const float eu = __expf(u.re) * __cosf(u.im);
#else
    const float eu = expf(u.re) * cosf(u.im);
#endif
    return eu - cdiv_re(n, d);
}

VAMP_DEV float humlicek_w4_re(float x, float y) {
    const float s = fabsf(x) + y;
    if (s >= 15.0f) return w4_region1(x, y);
    if (s >= 5.5f) return w4_region2(x, y);
    if (y >= 0.195f * fabsf(x) - 0.176f) return w4_region3(x, y);
    return w4_region4(x, y);
}

}  // namespace vamp
