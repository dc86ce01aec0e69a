/* A plain-C client of include/vamp_hip.h (no Python, no torch): one Gaussian absorption line on a
 * 300-pixel region, log-posterior of a few parameter vectors checked against the closed form
 * computed here, a short stretch-move run, and the error path.  Built with gcc by
 * tests/test_abi.py; running it needs a GPU.  Test infrastructure only. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/vamp_hip.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != VAMP_OK) {                                                        \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, vamp_last_error());        \
            return 1;                                                                \
        }                                                                            \
    } while (0)

enum { P = 300, W = 32, D = 3 };

static double lnprob_host(const double* x, const double* flux, const double* noise, const double* th) {
    /* priors of vpfits.py:239-252 with the bounds of :250,320, known-noise chi^2 likelihood */
    const double a = th[0], c = th[1], s = th[2];
    const double c_lo = x[0], c_hi = x[P - 1], s_max = (x[P - 1] - x[0]) / 2.0;
    if (!(a >= 0.0) || c < c_lo || c > c_hi || s < 0.0 || s > s_max) return -INFINITY;
    double lp = log(a * exp(-a)) - log(c_hi - c_lo) - log(s_max), chi = 0.0;
    for (int i = 0; i < P; ++i) {
        const double u = (x[i] - c) / s;
        const double m = exp(-a * exp(-0.5 * u * u));
        const double r = (flux[i] - m) / noise[i];
        chi += r * r;
    }
    return lp - 0.5 * chi;
}

int main(void) {
    int ndev = 0;
    if (vamp_version() != VAMP_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    CHECK(vamp_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no device\n"); return 1; }
    static double x[P], flux[P], noise[P], theta[W * D], lnp[W], chi2[W];
    unsigned s = 12345u;
    for (int i = 0; i < P; ++i) {
        x[i] = i - 0.5 * (P - 1);
        const double u = (x[i] - 10.0) / 12.0;
        s = s * 1664525u + 1013904223u;
        flux[i] = exp(-1.3 * exp(-0.5 * u * u)) + 0.02 * ((s >> 8) / 16777216.0 - 0.5);
        noise[i] = 0.02;
    }
    for (int w = 0; w < W; ++w) {
        theta[w * D + 0] = 1.3 * (1.0 + 0.01 * (w % 7 - 3));
        theta[w * D + 1] = 10.0 + 0.1 * (w % 5 - 2);
        theta[w * D + 2] = 12.0 * (1.0 + 0.01 * (w % 3 - 1));
    }
    theta[5 * D + 0] = -1.0;               /* outside the prior: -inf, not an error */
    vamp_ctx* ctx = NULL;
    CHECK(vamp_ctx_create(&ctx, 0, VAMP_F64, VAMP_WOFZ_ACCURATE));
    const int64_t pix_off[2] = {0, P};
    const int32_t ncomp[1] = {1};
    CHECK(vamp_set_regions(ctx, 1, pix_off, x, flux, noise, ncomp, VAMP_GAUSS3, 0, 0, NULL, NULL));
    int nd = 0;
    CHECK(vamp_region_ndim(ctx, 0, &nd));
    if (nd != D) { fprintf(stderr, "ndim %d\n", nd); return 1; }
    CHECK(vamp_lnprob(ctx, 0, W, theta, lnp, chi2));
    double worst = 0.0;
    for (int w = 0; w < W; ++w) {
        const double want = lnprob_host(x, flux, noise, theta + w * D);
        if (isinf(want)) {
            if (!(isinf(lnp[w]) && lnp[w] < 0)) { fprintf(stderr, "walker %d: expected -inf\n", w); return 1; }
            continue;
        }
        const double e = fabs(lnp[w] - want) / fmax(1.0, fabs(want));
        if (e > worst) worst = e;
    }
    if (worst > 1e-10) { fprintf(stderr, "lnprob mismatch %.3e\n", worst); return 1; }
    /* sampler: 20 steps, all finite, something accepted */
    theta[5 * D + 0] = 1.3;
    static double chain[20 * W * D];
    static int64_t nacc[W];
    double seconds = 0.0;
    CHECK(vamp_sampler_init(ctx, W, theta, 42u, 2.0, W));
    CHECK(vamp_sampler_run(ctx, 20, 1, chain, NULL, nacc, &seconds));
    long total = 0;
    for (int w = 0; w < W; ++w) total += (long)nacc[w];
    for (int i = 0; i < 20 * W * D; ++i)
        if (!isfinite(chain[i])) { fprintf(stderr, "non-finite chain entry\n"); return 1; }
    if (total <= 0) { fprintf(stderr, "nothing accepted\n"); return 1; }
    /* error path: status code + message, no abort */
    if (vamp_lnprob(ctx, 3, W, theta, lnp, NULL) != VAMP_ERR_ARG || strlen(vamp_last_error()) == 0) {
        fprintf(stderr, "error path broken\n");
        return 1;
    }
    CHECK(vamp_ctx_destroy(ctx));
    printf("abi_client ok: lnprob max rel err %.2e, %ld accepted moves in 20 steps\n", worst, total);
    return 0;
}
