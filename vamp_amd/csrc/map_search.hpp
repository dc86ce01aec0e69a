// map_search.hpp -- Nelder-Mead searches for the maximum of the log-posterior of every (active)
// region at once; host code shared by libvamp_hip.so (candidates evaluated by k_lnprob, one launch
// per iteration) and by the host implementation of the same C ABI (oracle/vamp_cpu.cpp).
//
// The reference runs one PyMC MAP per fit (vpfits.py:352-358, 426: scipy's fmin on -logp); the
// simplex rules, coefficients (reflection 1, expansion 2, contraction 1/2, shrink 1/2), initial
// simplex (5 % per coordinate, 0.00025 for a zero coordinate) and the stopping test are scipy's
// `fmin`, so a region follows exactly the path fmin would take on its own -- but one iteration of
// ALL regions is one evaluation call: the four candidate points of every simplex (reflection,
// expansion, outside and inside contraction) are evaluated together, the rarely needed shrink in a
// second call.  The objective is f = -lnprob (1e300 where lnprob is not finite).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#if defined(__HIPCC__)
#define VAMP_NM_HD __host__ __device__ inline
#else
#define VAMP_NM_HD inline
#endif

namespace vamp {

// ---- the arithmetic of one simplex update, shared by the host search below and the device search
// (k_map_search in vamp_hip.hip: one workgroup per region, the whole search in ONE launch) ---------------
// scipy forms these points with numpy, i.e. every product and sum rounded on its own; fused multiply-adds
// are switched off here so that host and device follow fmin's path to the last bit of the objective.
// which: 0 reflection, 1 expansion, 2 outside contraction, 3 inside contraction (rho 1, chi 2, psi 1/2)
VAMP_NM_HD double nm_candidate(int which, double xbar, double worst) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    switch (which) {
        case 0: return (1.0 + 1.0) * xbar - 1.0 * worst;
        case 1: return (1.0 + 1.0 * 2.0) * xbar - 1.0 * 2.0 * worst;
        case 2: return (1.0 + 0.5 * 1.0) * xbar - 0.5 * 1.0 * worst;
        default: return (1.0 - 0.5) * xbar + 0.5 * worst;
    }
}
// shrink towards the best vertex (sigma 1/2)
VAMP_NM_HD double nm_shrink(double best, double v) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    return best + 0.5 * (v - best);
}
// coordinate k of vertex k + 1 of the start simplex (5 %, or 0.00025 for a zero coordinate)
VAMP_NM_HD double nm_start_coordinate(double y) { return (y != 0.0) ? (1.0 + 0.05) * y : 0.00025; }
// f = -lnprob, 1e300 where lnprob is not finite
VAMP_NM_HD double nm_objective(double lnprob) { return (lnprob - lnprob == 0.0) ? -lnprob : 1e300; }

// dims[r] / offs[r]: dimension of region r and the offset of its vector inside theta0 / theta_best.
// eval_all(W, theta, lnprob): log-posteriors of W points per region -- theta holds the regions'
// [W, D_r] blocks one after the other (block r at W * offs[r]), lnprob is [R, W]; returns 0 or an
// error code, which is passed through.  Candidate points are formed exactly as numpy forms them
// (no fused multiply-add), so the path is fmin's to the last bit of the objective.
template <class EvalAll>
int nelder_mead_all(int R, const int* dims, const long long* offs, const double* theta0, const uint8_t* active,
                    int64_t maxiter, int64_t maxfun, double xtol, double ftol, double* theta_best, int64_t* iterations,
                    EvalAll&& eval_all) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    struct Simplex {
        int N = 0;
        long long off = 0;             // d_before
        std::vector<double> sim, f;    // (N+1) x N vertices, N+1 values, kept sorted by value
        std::vector<double> cand;      // 4 x N: reflection, expansion, outside, inside contraction
        double f0 = 0.0;               // value at the start point
        long long it = 0, calls = 0;
        bool live = false, shrink = false;
    };
    std::vector<Simplex> S(R);
    int Nmax = 0;
    long long dsum = 0;
    for (int r = 0; r < R; ++r) {
        Simplex& s = S[r];
        s.N = dims[r];
        s.off = offs[r];
        s.live = !active || active[r];
        Nmax = std::max(Nmax, s.N);
        dsum = s.off + s.N;
        s.sim.assign((size_t)(s.N + 1) * s.N, 0.0);
        s.f.assign(s.N + 1, 0.0);
        s.cand.assign((size_t)4 * s.N, 0.0);
        const double* x0 = theta0 + s.off;
        for (int k = 0; k <= s.N; ++k)
            for (int d = 0; d < s.N; ++d) s.sim[(size_t)k * s.N + d] = x0[d];
        for (int k = 0; k < s.N; ++k) {
            double& y = s.sim[(size_t)(k + 1) * s.N + k];
            y = nm_start_coordinate(y);
        }
    }
    std::vector<double> th, lp;
    auto objective = [](double v) { return nm_objective(v); };
    // evaluate W rows per region; row(r, w) supplies the point (pad rows repeat the best vertex)
    auto evaluate = [&](int W, auto&& row) -> int {
        th.resize((size_t)W * dsum);
        lp.resize((size_t)W * R);
        for (int r = 0; r < R; ++r) {
            const Simplex& s = S[r];
            double* dst = th.data() + (size_t)W * s.off;
            for (int w = 0; w < W; ++w) {
                const double* src = row(r, w);
                for (int d = 0; d < s.N; ++d) dst[(size_t)w * s.N + d] = src[d];
            }
        }
        return eval_all(W, th.data(), lp.data());
    };
    auto sort_simplex = [](Simplex& s) {           // insertion sort, stable: ties keep their order
        for (int i = 1; i <= s.N; ++i) {
            int j = i;
            while (j > 0 && s.f[j] < s.f[j - 1]) {
                std::swap(s.f[j], s.f[j - 1]);
                for (int d = 0; d < s.N; ++d) std::swap(s.sim[(size_t)j * s.N + d], s.sim[(size_t)(j - 1) * s.N + d]);
                --j;
            }
        }
    };
    // initial simplex
    int rc = evaluate(Nmax + 1, [&](int r, int w) { return &S[r].sim[(size_t)std::min(w, S[r].N) * S[r].N]; });
    if (rc) return rc;
    for (int r = 0; r < R; ++r) {
        Simplex& s = S[r];
        for (int k = 0; k <= s.N; ++k) s.f[k] = objective(lp[(size_t)r * (Nmax + 1) + k]);
        s.f0 = s.f[0];
        s.calls = s.N + 1;
        sort_simplex(s);
    }
    for (;;) {
        bool any = false;
        for (int r = 0; r < R; ++r) {
            Simplex& s = S[r];
            if (!s.live) continue;
            // maxfun == 0: scipy's default of 200 evaluations per dimension (what PyMC's MAP.fit leaves it at)
            const long long fun_cap = maxfun > 0 ? (long long)maxfun : 200ll * s.N;
            if (s.calls >= fun_cap || s.it + 1 >= maxiter) { s.live = false; continue; }   // fmin counts from 1
            double dx = 0.0, df = 0.0;
            for (int k = 1; k <= s.N; ++k) {
                df = std::max(df, std::fabs(s.f[0] - s.f[k]));
                for (int d = 0; d < s.N; ++d) dx = std::max(dx, std::fabs(s.sim[(size_t)k * s.N + d] - s.sim[d]));
            }
            if (dx <= xtol && df <= ftol) { s.live = false; continue; }
            any = true;
            const double* worst = &s.sim[(size_t)s.N * s.N];
            for (int d = 0; d < s.N; ++d) {
                double xbar = 0.0;
                for (int k = 0; k < s.N; ++k) xbar += s.sim[(size_t)k * s.N + d];
                xbar /= s.N;
                for (int w = 0; w < 4; ++w) s.cand[w * s.N + d] = nm_candidate(w, xbar, worst[d]);   // reflection, expansion, outside / inside contraction
            }
        }
        if (!any) break;
        rc = evaluate(4, [&](int r, int w) { return S[r].live ? &S[r].cand[(size_t)w * S[r].N] : &S[r].sim[0]; });
        if (rc) return rc;
        bool any_shrink = false;
        for (int r = 0; r < R; ++r) {
            Simplex& s = S[r];
            if (!s.live) continue;
            const double fr = objective(lp[(size_t)r * 4 + 0]), fe = objective(lp[(size_t)r * 4 + 1]);
            const double foc = objective(lp[(size_t)r * 4 + 2]), fic = objective(lp[(size_t)r * 4 + 3]);
            double* worst = &s.sim[(size_t)s.N * s.N];
            auto take = [&](int which, double fv) {
                for (int d = 0; d < s.N; ++d) worst[d] = s.cand[(size_t)which * s.N + d];
                s.f[s.N] = fv;
            };
            s.shrink = false;
            s.calls += 1;
            if (fr < s.f[0]) {
                s.calls += 1;
                if (fe < fr) take(1, fe); else take(0, fr);
            } else if (fr < s.f[s.N - 1]) {
                take(0, fr);
            } else if (fr < s.f[s.N]) {
                s.calls += 1;
                if (foc <= fr) take(2, foc); else s.shrink = true;
            } else {
                s.calls += 1;
                if (fic < s.f[s.N]) take(3, fic); else s.shrink = true;
            }
            if (s.shrink) {
                any_shrink = true;
                for (int k = 1; k <= s.N; ++k)
                    for (int d = 0; d < s.N; ++d)
                        s.sim[(size_t)k * s.N + d] = nm_shrink(s.sim[d], s.sim[(size_t)k * s.N + d]);
            }
        }
        if (any_shrink) {
            rc = evaluate(Nmax, [&](int r, int w) {
                const Simplex& s = S[r];
                return (s.live && s.shrink && w < s.N) ? &s.sim[(size_t)(w + 1) * s.N] : &s.sim[0];
            });
            if (rc) return rc;
            for (int r = 0; r < R; ++r) {
                Simplex& s = S[r];
                if (!s.live || !s.shrink) continue;
                for (int k = 1; k <= s.N; ++k) s.f[k] = objective(lp[(size_t)r * Nmax + (k - 1)]);
                s.calls += s.N;
            }
        }
        for (int r = 0; r < R; ++r) {
            Simplex& s = S[r];
            if (!s.live) continue;
            s.it += 1;
            sort_simplex(s);
        }
    }
    // results: the best vertex, unless it is worse than the start point; lnprob (and chi2) there
    for (int r = 0; r < R; ++r) {
        const Simplex& s = S[r];
        const bool keep_start = !(!active || active[r]) || s.f[0] > s.f0;
        const double* best = keep_start ? theta0 + s.off : &s.sim[0];
        for (int d = 0; d < s.N; ++d) theta_best[s.off + d] = best[d];
        if (iterations) iterations[r] = s.it;
    }
    return 0;
}

}  // namespace vamp
