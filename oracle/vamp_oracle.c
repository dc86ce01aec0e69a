/* vamp_oracle.c -- plain-C restatement of the VAMP MCMC hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (vamp_amd) never does.  It restates the same reference arithmetic as
 * oracle/vamp_oracle.py (citations: /root/reference/vamp_1.0):
 *   Gaussian / Voigt tau-profiles   vpfits.py:43-54, 57-76
 *   flux = exp(-sum tau_k)          physics.py:98-105, vpfits.py:334-336
 *   chi^2 / Normal likelihood       vpfits.py:109-131, 39, 341
 *   priors                          vpfits.py:239-252, 283-297, 320, 326
 *   (N,b,z) maps                    physics.py:3-27, 116-134
 * and the stretch move of SURVEY Appendix B (Goodman & Weare 2010, emcee red/blue semantics) with
 * the same Philox4x32-10 counter-based draws as the Python oracle.
 *
 * Re w(z) is computed here by an independent straightforward scheme (Laplace continued fraction
 * evaluated backwards with a |z|-dependent depth; near the real axis the midpoint trapezoid rule
 * with explicit exp() per node) and is pinned to scipy.special.wofz through
 * tests/golden/wofz_grid.npz (tests/test_oracle.py).  It exists so that the CPU baseline of
 * bench.py runs multi-threaded native code on all host cores (OpenMP) instead of numpy.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VO_PI 3.14159265358979323846
#define C_LIGHT 2.98e8 /* physics.py:3 */
#define SIGMA0 0.0263  /* physics.py:4 */

static double exp_neg_sq(double x) {
    double s = x * x;
    double e = fma(x, x, -s);
    return exp(-s) * (1.0 - e);
}

/* Re w(x+iy), x >= 0, y >= 0 */
static double wofz_re1(double x, double y) {
    const double r2 = x * x + y * y;
    if (r2 >= 64.0) {
        if (r2 != r2) return NAN;
        if (!(r2 < 1e300)) { /* overflow-safe far wing: y/(sqrt(pi) r^2) */
            double m = x > y ? x : y, xs = x / m, ys = y / m;
            return (ys / (xs * xs + ys * ys)) / m / sqrt(VO_PI);
        }
        /* depth: generous version of the classic nu(|z|) rule */
        int nu = (int)(6.0 + 1100.0 / (26.0 + r2)) + 3;
        if (r2 < 400.0) nu += 12;
        double complex z = x + I * y, w = 0.0;
        for (int k = nu; k >= 1; --k) w = (0.5 * k) / (z - w);
        double complex res = I / sqrt(VO_PI) / (z - w);
        double h = creal(res);
        if (y < 1e-9) h += exp_neg_sq(x);
        return h;
    }
    /* midpoint trapezoid, h = 1/2, nodes u_n = (n + 1/2)/2, n in Z, window of 27 around x */
    const double h = 0.5;
    int n0 = (int)floor(x / h);
    double S = 0.0;
    for (int n = n0 - 13; n <= n0 + 13; ++n) {
        double u = (n + 0.5) * h;
        S += exp(-(x - u) * (x - u)) / (u * u + y * y);
    }
    double H = h * y / VO_PI * S;
    if (y < 4.5) {
        double t = exp(-2.0 * VO_PI * y / h);
        double A = 2.0 * exp(y * y - 2.0 * VO_PI * y / h) / (1.0 + t);
        H += A * exp_neg_sq(x) * cos(2.0 * x * y);
    }
    return H;
}

void vo_wofz_re(int64_t n, const double* x, const double* y, double* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = wofz_re1(fabs(x[i]), y[i]);
}

typedef struct {
    int64_t P;
    const double *x, *flux, *noise;
    int K, mode, sample_sd, include_norm;
    double c_lo, c_hi, sigma_max, fwhm_max;
    double l_fixed, line, x_origin, x_scale;
} vo_region;

static double xexp_logp(double v) {
    if (!(v >= 0.0) || !isfinite(v)) return -INFINITY;
    return log(v * exp(-v)); /* vpfits.py:244, literally */
}
static double unif_logp(double v, double lo, double hi) { return (v >= lo && v <= hi) ? -log(hi - lo) : -INFINITY; }

/* log-posterior of one parameter vector; *chi2 = chi^2 (or unweighted SSR with sample_sd) */
static double lnprob1(const vo_region* R, const double* th, double* chi2, double* tau_buf) {
    const int q = (R->mode == 1) ? 4 : 3;
    const int K = R->K;
    double a[16], c[16], L[16], G[16], s[16];
    double lp = 0.0;
    for (int k = 0; k < K; ++k) {
        const double* t = th + q * k;
        if (R->mode == 0) {
            a[k] = t[0]; c[k] = t[1]; s[k] = t[2];
            lp += xexp_logp(a[k]) + unif_logp(c[k], R->c_lo, R->c_hi) + unif_logp(s[k], 0.0, R->sigma_max);
        } else if (R->mode == 1) {
            a[k] = t[0]; c[k] = t[1]; L[k] = t[2]; G[k] = t[3];
            lp += xexp_logp(a[k]) + unif_logp(c[k], R->c_lo, R->c_hi) + unif_logp(L[k], 0.0, R->fwhm_max) +
                  unif_logp(G[k], 0.0, R->fwhm_max);
        } else {
            double sig = t[1] * 1.0e3 * sqrt(2.0) / (2.355 * (R->line * 1.0e-10));
            a[k] = t[0] * SIGMA0 / (sig * sqrt(2.0 * VO_PI));
            c[k] = (C_LIGHT / (R->line * (1.0 + t[2]) * 1.0e-10) - R->x_origin) / R->x_scale;
            G[k] = (sig / R->x_scale) * (2.0 * sqrt(2.0 * log(2.0)));
            L[k] = R->l_fixed;
            lp += xexp_logp(a[k]) + unif_logp(c[k], R->c_lo, R->c_hi) + unif_logp(G[k], 0.0, R->fwhm_max);
        }
    }
    double sd = 1.0;
    if (R->sample_sd) {
        sd = th[q * K];
        lp += unif_logp(sd, 0.0, 1.0);
    }
    if (!(lp > -INFINITY)) {
        if (chi2) *chi2 = NAN;
        return -INFINITY;
    }
    const double sl2 = sqrt(log(2.0));
    for (int64_t i = 0; i < R->P; ++i) tau_buf[i] = 0.0;
    for (int k = 0; k < K; ++k) {
        if (R->mode == 0) {
            for (int64_t i = 0; i < R->P; ++i) {
                double u = (R->x[i] - c[k]) / s[k];
                tau_buf[i] += a[k] * exp(-0.5 * u * u);
            }
        } else {
            const double amp = a[k] * L[k] * sqrt(VO_PI) * sl2 / G[k];
            const double yy = L[k] * sl2 / G[k];
            for (int64_t i = 0; i < R->P; ++i) {
                double xx = fabs(2.0 * (R->x[i] - c[k]) * sl2 / G[k]);
                tau_buf[i] += amp * wofz_re1(xx, yy);
            }
        }
    }
    double ssum = 0.0;
    for (int64_t i = 0; i < R->P; ++i) {
        double m = exp(-tau_buf[i]);
        double r = R->sample_sd ? (R->flux[i] - m) : (R->flux[i] - m) / R->noise[i];
        ssum += r * r;
    }
    if (chi2) *chi2 = ssum;
    double ll;
    if (R->sample_sd) {
        double t = 1.0 / (sd * sd);
        ll = (double)R->P * 0.5 * log(t / (2.0 * VO_PI)) - 0.5 * t * ssum;
    } else {
        ll = -0.5 * ssum;
        if (R->include_norm)
            for (int64_t i = 0; i < R->P; ++i) ll -= 0.5 * log(2.0 * VO_PI * R->noise[i] * R->noise[i]);
    }
    double v = lp + ll;
    if (v != v) v = -INFINITY;
    return v;
}

static void fill_region(vo_region* R, int64_t P, const double* x, const double* flux, const double* noise, int K, int mode,
                        int sample_sd, int include_norm, const double* bounds, const double* nbz) {
    R->P = P; R->x = x; R->flux = flux; R->noise = noise; R->K = K; R->mode = mode;
    R->sample_sd = sample_sd; R->include_norm = include_norm;
    if (bounds) {
        R->c_lo = bounds[0]; R->c_hi = bounds[1]; R->sigma_max = bounds[2]; R->fwhm_max = bounds[3];
    } else {
        R->c_lo = x[0]; R->c_hi = x[P - 1];
        R->sigma_max = (x[P - 1] - x[0]) / 2.0;
        R->fwhm_max = R->sigma_max * 2 * sqrt(2 * log(2.0));
    }
    if (nbz) { R->l_fixed = nbz[0]; R->line = nbz[1]; R->x_origin = nbz[2]; R->x_scale = nbz[3]; }
    else { R->l_fixed = 0; R->line = 1215.67; R->x_origin = 0; R->x_scale = 1; }
}

/* batch log-posterior, OpenMP over walkers */
int vo_lnprob(int64_t P, const double* x, const double* flux, const double* noise, int K, int mode, int sample_sd,
              int include_norm, const double* bounds, const double* nbz, int64_t W, const double* theta, double* lnprob,
              double* chi2, int nthreads) {
    vo_region R;
    fill_region(&R, P, x, flux, noise, K, mode, sample_sd, include_norm, bounds, nbz);
    const int D = ((mode == 1) ? 4 : 3) * K + (sample_sd ? 1 : 0);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        double* tau = (double*)malloc(sizeof(double) * (size_t)P);
#pragma omp for schedule(dynamic, 4)
        for (int64_t w = 0; w < W; ++w) {
            double c2;
            lnprob[w] = lnprob1(&R, theta + w * D, &c2, tau);
            if (chi2) chi2[w] = c2;
        }
        free(tau);
    }
    return 0;
}

/* ---- Philox4x32-10 and the split permutation: same arithmetic as oracle/vamp_oracle.py ---- */
static void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
static double u53(uint32_t hi, uint32_t lo) { return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0); }

static uint32_t split_perm(uint64_t seed, uint32_t step, uint32_t chunk, uint32_t region, uint32_t slot, uint32_t block) {
    uint32_t r[4] = {chunk, step, 2u, region};
    philox(r, (uint32_t)seed, (uint32_t)(seed >> 32));
    int bits = 0;
    for (uint32_t t = block - 1; t; t >>= 1) ++bits;
    if (bits < 1) bits = 1;
    const uint64_t mask = (1ull << bits) - 1ull;
    int sh = bits / 2; if (sh < 1) sh = 1;
    const uint64_t m0 = ((uint64_t)r[0] << 1) | 1ull, m2 = ((uint64_t)r[2] << 1) | 1ull;
    uint64_t v = slot;
    for (;;) {
        v = (v * m0 + r[1]) & mask; v ^= v >> sh;
        v = (v * m2 + r[3]) & mask; v ^= v >> sh;
        v = (v * 0x9E3779B1ull + (r[0] ^ r[3])) & mask; v ^= v >> sh;
        if (v < block) return (uint32_t)v;
    }
}

/* n_steps of the stretch move on one region; X[W,D] and lnp[W] updated in place. */
int vo_sampler_run(int64_t P, const double* x, const double* flux, const double* noise, int K, int mode, int sample_sd,
                   int include_norm, const double* bounds, const double* nbz, int64_t W, double* X, double* lnp,
                   int64_t* n_accept, int64_t n_steps, int64_t step0, uint64_t seed, double a, int32_t block,
                   int nthreads) {
    vo_region R;
    fill_region(&R, P, x, flux, noise, K, mode, sample_sd, include_norm, bounds, nbz);
    const int D = ((mode == 1) ? 4 : 3) * K + (sample_sd ? 1 : 0);
    const int64_t halfW = W / 2;
    const uint32_t hb = (uint32_t)block / 2;
    double* Q = (double*)malloc(sizeof(double) * (size_t)halfW * D);
    double* lq = (double*)malloc(sizeof(double) * (size_t)halfW);
    int64_t* ws = (int64_t*)malloc(sizeof(int64_t) * (size_t)halfW);
    double* zz = (double*)malloc(sizeof(double) * (size_t)halfW);
    double* lu = (double*)malloc(sizeof(double) * (size_t)halfW);
    if (!Q || !lq || !ws || !zz || !lu) return -4;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    for (int64_t it = 0; it < n_steps; ++it) {
        const uint32_t step = (uint32_t)(step0 + it);
        for (int half = 0; half < 2; ++half) {
#pragma omp parallel
            {
                double* tau = (double*)malloc(sizeof(double) * (size_t)P);
#pragma omp for schedule(dynamic, 4)
                for (int64_t sl = 0; sl < halfW; ++sl) {
                    const uint32_t chunk = (uint32_t)(sl / hb), pos = (uint32_t)(sl % hb);
                    const int64_t w = (int64_t)chunk * block + split_perm(seed, step, chunk, 0, pos + (half ? hb : 0), (uint32_t)block);
                    uint32_t r[4] = {(uint32_t)w, step, ((uint32_t)half << 8) | 0u, (uint32_t)((uint64_t)w >> 32)};
                    philox(r, (uint32_t)seed, (uint32_t)(seed >> 32));
                    const double t = (a - 1.0) * u53(r[0], r[1]) + 1.0;
                    const double z = t * t / a;
                    const uint64_t j = (uint64_t)(((unsigned __int128)((((uint64_t)r[2]) << 32) | r[3]) * (uint64_t)halfW) >> 64);
                    const uint32_t cch = (uint32_t)(j / hb), cpos = (uint32_t)(j % hb);
                    const int64_t wc = (int64_t)cch * block + split_perm(seed, step, cch, 0, cpos + (half ? 0 : hb), (uint32_t)block);
                    uint32_t r2[4] = {(uint32_t)w, step, ((uint32_t)half << 8) | 1u, (uint32_t)((uint64_t)w >> 32)};
                    philox(r2, (uint32_t)seed, (uint32_t)(seed >> 32));
                    const double u2 = u53(r2[0], r2[1]);
                    ws[sl] = w; zz[sl] = z; lu[sl] = u2 > 0 ? log(u2) : -INFINITY;
                    double* q = Q + sl * D;
                    for (int d = 0; d < D; ++d) {
                        const double c = X[wc * D + d];
                        q[d] = c - (c - X[w * D + d]) * z;
                    }
                    lq[sl] = lnprob1(&R, q, NULL, tau);
                }
                free(tau);
            }
            /* accept after every proposal of the half has been formed against the frozen complement */
            for (int64_t sl = 0; sl < halfW; ++sl) {
                const int64_t w = ws[sl];
                const double diff = (double)(D - 1) * log(zz[sl]) + lq[sl] - lnp[w];
                if (lu[sl] < diff) {
                    memcpy(X + w * D, Q + sl * D, sizeof(double) * D);
                    lnp[w] = lq[sl];
                    if (n_accept) n_accept[w] += 1;
                }
            }
        }
    }
    free(Q); free(lq); free(ws); free(zz); free(lu);
    return 0;
}

int vo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
