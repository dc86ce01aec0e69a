// vamp_cpu.cpp -- HOST implementation of the C ABI of include/vamp_hip.h  ->  oracle/libvamp_cpu.so
//
// TEST INFRASTRUCTURE AND CPU BASELINE ONLY ("same ABI implemented by libvamp_cpu.so for the
// host baseline"; BASELINE.md Baseline B).  It lives under oracle/ on purpose: only tests/, bench.py's
// cpu_baseline leg and __graft_entry__ may load it, and only when they name it explicitly
// (vamp_amd.HipContext(lib=...)); vamp_amd itself loads libvamp_hip.so or fails.  There is no fallback.
//
// What it is for
//   * the boundary without a GPU: every entry point of the header exists here with the same argument
//     checks, error codes, call-order rules and sharding arithmetic, so the ctypes layer, the VPfit
//     facade and the walker-sharded driver can be exercised in the CPU test suite;
//   * a multi-threaded C++ baseline: OpenMP over walkers, the Voigt evaluators of
//     vamp_amd/csrc/voigt_math.hpp in their host build (per pixel: J-fractions / near-axis rule; no
//     far-field interpolant, no Taylor tables), the same counter-based draws as the HIP sampler.
// It restates the same reference code as the HIP library: profiles vpfits.py:43-76, Tau2flux
// physics.py:98-105, likelihood vpfits.py:39,341, priors vpfits.py:239-252,283-297, (N,b,z) maps
// physics.py:6-27,116-134, MAP search vpfits.py:352-358 (csrc/map_search.hpp), and the stretch move of
// SURVEY Appendix B.  The independent checker of both libraries is oracle/vamp_oracle.py (scipy.wofz).
#include <omp.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "../include/vamp_hip.h"
#include "../vamp_amd/csrc/host_plan.hpp"
#include "../vamp_amd/csrc/map_search.hpp"
#include "../vamp_amd/csrc/voigt_math.hpp"

namespace {

constexpr int KMAX = VAMP_MAX_COMPONENTS;
constexpr double C_LIGHT = 2.98e8;     // physics.py:3 (the reference's value)
constexpr double SIGMA0 = 0.0263;      // physics.py:4
constexpr double SQRT_LN2 = 0.83255461115769775635;
constexpr double FWHM_PER_SIGMA = 2.35482004503094938202;
const double NEG_INF = -std::numeric_limits<double>::infinity();
const double POS_INF = std::numeric_limits<double>::infinity();

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

struct Region {
    long long pix_off = 0, theta_off = 0, walker_off = 0, d_before = 0;
    int P = 0, K = 0, mode = 0, D = 0, q = 0, sample_sd = 0, rng_id = 0;
    double c_lo = 0, c_hi = 0, w_max = 0, lp_c = 0, lp_w = 0;
    double l_fixed = 0, line = 0, x_origin = 0, x_scale = 1, norm_const = 0;
};

struct Line { double c, s, y, amp, pole, hy; };

// ---- counter-based RNG (Philox4x32-10) and the keyed red/blue split: as in the HIP sampler ------
struct U4 { uint32_t c0, c1, c2, c3; };
U4 philox(U4 c, uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = 0xD2511F53ull * c.c0, p1 = 0xCD9E8D57ull * c.c2;
        c = U4{(uint32_t)(p1 >> 32) ^ c.c1 ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.c3 ^ k1, (uint32_t)p0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
double u53(uint32_t hi, uint32_t lo) { return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0); }
constexpr uint32_t STREAM_MOVE = 0, STREAM_ACCEPT = 1, STREAM_SPLIT = 2;

uint32_t split_perm(uint64_t seed, uint32_t step, uint32_t chunk, uint32_t region, uint32_t slot, uint32_t block) {
    const U4 r = philox({chunk, step, STREAM_SPLIT, region}, (uint32_t)seed, (uint32_t)(seed >> 32));
    int bits = 32 - __builtin_clz((block - 1) | 1u);
    if (bits < 1) bits = 1;
    const uint64_t mask = (1ull << bits) - 1ull;
    int sh = bits / 2;
    if (sh < 1) sh = 1;
    const uint64_t m0 = ((uint64_t)r.c0 << 1) | 1ull, m2 = ((uint64_t)r.c2 << 1) | 1ull;
    uint64_t v = slot;
    for (;;) {
        v = (v * m0 + r.c1) & mask;  v ^= v >> sh;
        v = (v * m2 + r.c3) & mask;  v ^= v >> sh;
        v = (v * 0x9E3779B1ull + (r.c0 ^ r.c3)) & mask;  v ^= v >> sh;
        if (v < block) return (uint32_t)v;
    }
}

}  // namespace

struct vamp_ctx {
    int dtype = VAMP_F64;
    bool f32 = false;
    int n_regions = 0, mode = VAMP_VOIGT4;
    std::vector<Region> R;
    std::vector<double> x, f, wt;
    int threads = 1;
    // sampler
    bool ready = false, X_ext = false;
    long long W = 0, total_theta = 0, total_walkers = 0, step = 0;
    int split_block = 0;
    double a = 2.0;
    uint64_t seed = 0;
    std::vector<double> X_own, lnp_own;
    double* X = nullptr;
    double* lnp = nullptr;
    std::vector<long long> nacc;
    long long slot_begin = 0, slot_end = 0, part_slots = 0, part_stride = 0;
    int shard_rank = 0, shard_world = 1, shard_parts = 1;
    bool comm = false;
    std::vector<double> send, recv;        // [parts][part_slots][D+1], [parts][world*part_slots][D+1]
    std::vector<unsigned> part_step;
    std::vector<int> part_half;
    bool timing = false;
    double timing_ms = 0.0;
    long long timing_launches = 0;
    // the launch plan of the HIP library (csrc/host_plan.hpp), computed here too: nothing is launched on the host, but
    // the product's own plan arithmetic runs -- under the sanitizers in the _asan build -- and vamp_region_class answers
    int packing = 0;
    vamp::plan::ClassPlan classes;
};

namespace {

double xexp_logp(double v) {           // vpfits.py:239-244, literally
    if (!(v >= 0.0) || !std::isfinite(v)) return NEG_INF;
    return std::log(v * std::exp(-v));
}
double uniform_logp(double v, double lo, double hi, double lp) { return (v >= lo && v <= hi) ? lp : NEG_INF; }

// parameters -> line records + log-prior (the device's stage_lines)
double stage(const Region& R, const double* t0, Line* ln) {
    double lp = 0.0;
    for (int k = 0; k < R.K; ++k) {
        const double* t = t0 + R.q * k;
        double a, c, Lw = 0.0, G = 0.0, sg = 0.0, l;
        if (R.mode == VAMP_GAUSS3) {
            a = t[0]; c = t[1]; sg = t[2];
            l = xexp_logp(a) + uniform_logp(c, R.c_lo, R.c_hi, R.lp_c) + uniform_logp(sg, 0.0, R.w_max, R.lp_w);
        } else if (R.mode == VAMP_VOIGT4) {
            a = t[0]; c = t[1]; Lw = t[2]; G = t[3];
            l = xexp_logp(a) + uniform_logp(c, R.c_lo, R.c_hi, R.lp_c) + uniform_logp(Lw, 0.0, R.w_max, R.lp_w) +
                uniform_logp(G, 0.0, R.w_max, R.lp_w);
        } else {   // NBZ3: inverse of physics.py:15,27,120,134
            const double sig = t[1] * 1.0e3 * 1.41421356237309514547 / (2.355 * (R.line * 1.0e-10));
            a = t[0] * SIGMA0 / (sig * 2.50662827463100024161);
            c = (C_LIGHT / (R.line * (1.0 + t[2]) * 1.0e-10) - R.x_origin) / R.x_scale;
            G = (sig / R.x_scale) * FWHM_PER_SIGMA;
            Lw = R.l_fixed;
            l = xexp_logp(a) + uniform_logp(c, R.c_lo, R.c_hi, R.lp_c) + uniform_logp(G, 0.0, R.w_max, R.lp_w);
        }
        Line& r = ln[k];
        r.c = c;
        if (R.mode == VAMP_GAUSS3) {
            r.s = 1.0 / sg; r.y = 0.0; r.amp = a; r.pole = 0.0; r.hy = 0.0;
        } else {
            r.s = 2.0 * SQRT_LN2 / G;
            r.y = Lw * SQRT_LN2 / G;
            r.amp = a * r.y;
            r.pole = vamp::core_pole_factor(r.y);
            r.hy = vamp::core_hy(r.y);
            if (!(r.s < POS_INF) || !(r.y < POS_INF)) l = NEG_INF;     // degenerate width: rejected, as on the device
        }
        lp += l;
    }
    if (R.sample_sd) lp += uniform_logp(t0[R.D - 1], 0.0, 1.0, 0.0);      // sd ~ U(0,1), vpfits.py:39
    return lp;
}

double loglike_from_sum(const Region& R, const double* t0, double ssum) {
    if (R.sample_sd) {
        const double sd = t0[R.D - 1], t = 1.0 / (sd * sd);
        return (double)R.P * 0.5 * std::log(t / (2.0 * vamp::PI)) - 0.5 * t * ssum;      // vpfits.py:39,341
    }
    return -0.5 * ssum + R.norm_const;                                                   // vpfits.py:118
}

// log-posterior of one parameter vector; chi receives the (weighted) sum of squared residuals
double lnprob_one(const vamp_ctx* c, const Region& R, const double* t0, double* chi_out) {
    Line ln[KMAX];
    const double lp = stage(R, t0, ln);
    if (!(lp > NEG_INF) || lp != lp) {
        if (chi_out) *chi_out = std::numeric_limits<double>::quiet_NaN();
        return NEG_INF;
    }
    const double *x = c->x.data() + R.pix_off, *f = c->f.data() + R.pix_off, *wt = c->wt.data() + R.pix_off;
    double chi = 0.0;
    if (c->f32) {       // fp32 pixel arithmetic, Humlicek W4, chi^2 accumulated in fp64 (BASELINE.json config 5)
        for (int i = 0; i < R.P; ++i) {
            float tau = 0.0f;
            const float xi = (float)x[i];
            for (int k = 0; k < R.K; ++k) {
                const float u = std::fabs(xi - (float)ln[k].c) * (float)ln[k].s;
                if (R.mode == VAMP_GAUSS3) tau += (float)ln[k].amp * std::exp(-0.5f * (u * u));
                else tau += (float)(ln[k].amp * vamp::SQRT_PI) * vamp::humlicek_w4_re(std::fmin(u, vamp::W4_XMAX), (float)ln[k].y);
            }
            const float m = std::exp(-tau), r = ((float)f[i] - m) * (float)wt[i];
            chi += (double)r * (double)r;
        }
    } else {
        double dtab[KMAX][vamp::DTAB_N];
        if (R.mode != VAMP_GAUSS3)
            for (int k = 0; k < R.K; ++k)
                for (int n = 0; n < vamp::DTAB_N; ++n) dtab[k][n] = vamp::core_dtab_entry(n, ln[k].y);
        for (int i = 0; i < R.P; ++i) {
            double tau = 0.0;
            for (int k = 0; k < R.K; ++k) {
                if (R.mode == VAMP_GAUSS3) {
                    const double u = (x[i] - ln[k].c) * ln[k].s;
                    tau += ln[k].amp * std::exp(-0.5 * (u * u));
                } else {
                    tau += ln[k].amp * vamp::voigt_Hs(std::fabs(x[i] - ln[k].c) * ln[k].s, ln[k].y, dtab[k], ln[k].pole, ln[k].hy);
                }
            }
            const double r = (f[i] - std::exp(-tau)) * wt[i];
            chi += r * r;
        }
    }
    if (chi_out) *chi_out = chi;
    double v = lp + loglike_from_sum(R, t0, chi);
    if (v != v) v = NEG_INF;                 // NaN -> -inf (emcee convention)
    return v;
}

void lnprob_block(const vamp_ctx* c, int region, long long W, const double* theta, double* out, double* chi) {
    const Region& R = c->R[region];
#pragma omp parallel for schedule(dynamic, 4) num_threads(c->threads)
    for (long long w = 0; w < W; ++w) {
        double ch;
        out[w] = lnprob_one(c, R, theta + w * R.D, &ch);
        if (chi) chi[w] = ch;
    }
}

int lnprob_all_impl(const vamp_ctx* c, long long W, const double* theta, double* out, double* chi) {
    for (int r = 0; r < c->n_regions; ++r)
        lnprob_block(c, r, W, theta + W * c->R[r].d_before, out + (long long)r * W, chi ? chi + (long long)r * W : nullptr);
    return 0;
}

struct Move { long long ws, wc; double z, logu; };
Move draw_move(const vamp_ctx* c, unsigned step, int half, int region, long long a_loc) {
    const long long halfW = c->W / 2;
    const uint32_t hb = (uint32_t)(c->split_block / 2);
    const uint32_t chunk = (uint32_t)(a_loc / hb), pos = (uint32_t)(a_loc % hb);
    Move d;
    const uint32_t rid = (uint32_t)c->R[region].rng_id;      // the region's identity in the draw keys
    d.ws = (long long)chunk * c->split_block + split_perm(c->seed, step, chunk, rid, pos + (half ? hb : 0u), (uint32_t)c->split_block);
    const long long gid = (long long)rid * c->W + d.ws;
    const uint32_t k0 = (uint32_t)c->seed, k1 = (uint32_t)(c->seed >> 32);
    const U4 r = philox({(uint32_t)gid, step, ((uint32_t)half << 8) | STREAM_MOVE, (uint32_t)((uint64_t)gid >> 32)}, k0, k1);
    const double t = (c->a - 1.0) * u53(r.c0, r.c1) + 1.0;
    d.z = t * t / c->a;
    const uint64_t j = (uint64_t)(((unsigned __int128)(((uint64_t)r.c2 << 32) | r.c3) * (unsigned __int128)(uint64_t)halfW) >> 64);
    const uint32_t cchunk = (uint32_t)(j / hb), cpos = (uint32_t)(j % hb);
    d.wc = (long long)cchunk * c->split_block + split_perm(c->seed, step, cchunk, rid, cpos + (half ? 0u : hb), (uint32_t)c->split_block);
    const U4 r2 = philox({(uint32_t)gid, step, ((uint32_t)half << 8) | STREAM_ACCEPT, (uint32_t)((uint64_t)gid >> 32)}, k0, k1);
    const double u2 = u53(r2.c0, r2.c1);
    d.logu = u2 > 0.0 ? std::log(u2) : NEG_INF;
    return d;
}

// one mover: propose, evaluate, accept; pk (may be null) receives the row it ends with
void move_one(vamp_ctx* c, const Region& R, long long ws, long long wc, double z, double logu, double* pk) {
    double q[4 * KMAX + 1];
    double* Xs = c->X + R.theta_off + ws * R.D;
    const double* Xc = c->X + R.theta_off + wc * R.D;
    for (int d = 0; d < R.D; ++d) q[d] = Xc[d] - (Xc[d] - Xs[d]) * z;        // q = c - (c - s) z
    const double lnp_q = lnprob_one(c, R, q, nullptr);
    const long long wg = R.walker_off + ws;
    const double lnp_s = c->lnp[wg];
    const double diff = (double)(R.D - 1) * std::log(z) + lnp_q - lnp_s;
    const bool accept = logu < diff;                                        // false for NaN
    if (pk) {
        for (int d = 0; d < R.D; ++d) pk[d] = accept ? q[d] : Xs[d];
        pk[R.D] = accept ? lnp_q : lnp_s;
    }
    if (accept) {
        for (int d = 0; d < R.D; ++d) Xs[d] = q[d];
        c->lnp[wg] = lnp_q;
        c->nacc[wg] += 1;
    }
}

// piece `part` of this ctx's share of one half-step.  Movers write only their own rows and read only
// rows of the frozen colour, so the loop is parallel and the update is in place.
void half_step_part(vamp_ctx* c, int half, int part) {
    const auto t0 = std::chrono::steady_clock::now();
    const long long halfW = c->W / 2;
    long long lo, hi;
    if (c->n_regions == 1) {
        lo = c->slot_begin + part * c->part_stride;
        hi = c->shard_parts > 1 ? lo + c->part_slots : c->slot_end;
    } else {
        lo = 0;
        hi = c->total_walkers / 2;
    }
    const unsigned step = (unsigned)c->step;
    double* pack = c->send.empty() ? nullptr : c->send.data() + (long long)part * c->part_slots * (c->R[0].D + 1);
    if (pack) {
        c->part_step[part] = step;
        c->part_half[part] = half;
    }
#pragma omp parallel for schedule(dynamic, 4) num_threads(c->threads)
    for (long long slot = lo; slot < hi; ++slot) {
        const int region = (int)(slot / halfW);
        const Region& R = c->R[region];
        const Move d = draw_move(c, step, half, region, slot - (long long)region * halfW);
        move_one(c, R, d.ws, d.wc, d.z, d.logu, pack ? pack + (slot - lo) * (R.D + 1) : nullptr);
    }
    if (c->timing) {
        c->timing_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        c->timing_launches += 1;
    }
}

// rows of piece `part` gathered from all ranks -> walker rows (the device's k_scatter_rows)
void scatter_part(vamp_ctx* c, int part, const double* rows) {
    const Region& R = c->R[0];
    const long long n_rows = (long long)c->shard_world * c->part_slots, own_lo = (long long)c->shard_rank * c->part_slots;
    const unsigned step = c->part_step[part];
    const int half = c->part_half[part];
    const uint32_t hb = (uint32_t)(c->split_block / 2);
    for (long long i = 0; i < n_rows; ++i) {
        if (i >= own_lo && i < own_lo + c->part_slots) continue;
        const long long slot = (long long)part * c->part_stride + i;
        const uint32_t chunk = (uint32_t)(slot / hb), pos = (uint32_t)(slot % hb);
        const long long ws = (long long)chunk * c->split_block + split_perm(c->seed, step, chunk, (uint32_t)R.rng_id, pos + (half ? hb : 0u), (uint32_t)c->split_block);
        const double* src = rows + i * (R.D + 1);
        std::memcpy(c->X + R.theta_off + ws * R.D, src, R.D * sizeof(double));
        c->lnp[R.walker_off + ws] = src[R.D];
    }
}

void half_step_all(vamp_ctx* c, int half) {
    for (int p = 0; p < c->shard_parts; ++p) {
        half_step_part(c, half, p);
        if (c->comm && !c->send.empty()) {       // a communicator of one rank: the gather is a self-copy
            const size_t n = (size_t)c->part_slots * (c->R[0].D + 1);
            std::memcpy(c->recv.data() + (size_t)p * n, c->send.data() + (size_t)p * n, n * sizeof(double));
            scatter_part(c, p, c->recv.data() + (size_t)p * n);
        }
    }
}

void free_sampler(vamp_ctx* c) {
    c->X_own.clear(); c->lnp_own.clear(); c->nacc.clear(); c->send.clear(); c->recv.clear();
    if (!c->X_ext) { c->X = nullptr; c->lnp = nullptr; }
    c->ready = false;
}

}  // namespace

extern "C" {

int vamp_version(void) { return VAMP_ABI_VERSION; }
const char* vamp_last_error(void) { return g_err.c_str(); }
int vamp_device_count(int* n) {
    if (!n) return fail(VAMP_ERR_ARG, "vamp_device_count: n is NULL");
    *n = 1;                       // the host
    return VAMP_OK;
}

int vamp_ctx_create(vamp_ctx** out, int device, int dtype, int wofz_kind) {
    if (!out) return fail(VAMP_ERR_ARG, "vamp_ctx_create: out is NULL");
    if (!((dtype == VAMP_F64 && wofz_kind == VAMP_WOFZ_ACCURATE) || (dtype == VAMP_F32 && wofz_kind == VAMP_WOFZ_HUMLICEK_W4)))
        return fail(VAMP_ERR_ARG, "vamp_ctx_create: supported pairs are (F64, ACCURATE) and (F32, HUMLICEK_W4)");
    if (device != 0) return fail(VAMP_ERR_ARG, "vamp_ctx_create: no such device");
    vamp_ctx* c = new (std::nothrow) vamp_ctx();
    if (!c) return fail(VAMP_ERR_NOMEM, "vamp_ctx_create: host allocation failed");
    c->dtype = dtype;
    c->f32 = dtype == VAMP_F32;
    c->threads = omp_get_max_threads();
    if (const char* e = getenv("VAMP_CPU_THREADS")) c->threads = std::max(1, atoi(e));
    *out = c;
    return VAMP_OK;
}
int vamp_ctx_destroy(vamp_ctx* c) { delete c; return VAMP_OK; }
int vamp_ctx_set_stream(vamp_ctx* c, void*) { return c ? VAMP_OK : fail(VAMP_ERR_ARG, "vamp_ctx_set_stream: ctx is NULL"); }
int vamp_ctx_set_stream_default(vamp_ctx* c) { return c ? VAMP_OK : fail(VAMP_ERR_ARG, "vamp_ctx_set_stream_default: ctx is NULL"); }
int vamp_ctx_synchronize(vamp_ctx* c) { return c ? VAMP_OK : fail(VAMP_ERR_ARG, "vamp_ctx_synchronize: ctx is NULL"); }
int vamp_ctx_set_option(vamp_ctx* c, const char* name, int64_t) {
    // the switches choose between device execution forms with identical results: nothing to switch on the host
    if (!c || !name) return fail(VAMP_ERR_ARG, "vamp_ctx_set_option: NULL argument");
    const std::string key(name);
    if (key != "map_device" && key != "resident" && key != "class_streams")
        return fail(VAMP_ERR_ARG, "vamp_ctx_set_option: unknown option '" + key + "' (map_device, resident, class_streams)");
    return VAMP_OK;
}
int vamp_ctx_set_packing(vamp_ctx* c, int lanes) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_ctx_set_packing: ctx is NULL");
    if (lanes != 0 && lanes != 16 && lanes != 64 && lanes != 65 && lanes != 256)
        return fail(VAMP_ERR_ARG, "vamp_ctx_set_packing: lanes_per_walker must be 0 (auto), 16, 64, 65 (64 + per-walker tables) or 256");
    c->packing = lanes;           // a launch shape: nothing to launch on the host, but the class plan follows it
    return VAMP_OK;
}

int vamp_set_regions(vamp_ctx* c, int n_regions, const int64_t* pix_off, const double* x, const double* flux,
                     const double* noise, const int32_t* n_comp, int mode, int sample_sd, int include_norm,
                     const double* bounds, const double* nbz) {
    if (!c || n_regions <= 0 || !pix_off || !x || !flux || !noise || !n_comp)
        return fail(VAMP_ERR_ARG, "vamp_set_regions: NULL argument or n_regions <= 0");
    if (mode != VAMP_GAUSS3 && mode != VAMP_VOIGT4 && mode != VAMP_NBZ3) return fail(VAMP_ERR_ARG, "vamp_set_regions: bad mode");
    if (mode == VAMP_NBZ3 && !nbz) return fail(VAMP_ERR_ARG, "vamp_set_regions: VAMP_NBZ3 needs nbz");
    if (pix_off[0] != 0) return fail(VAMP_ERR_ARG, "vamp_set_regions: pix_off[0] must be 0");
    free_sampler(c);
    c->n_regions = 0;
    const int q = (mode == VAMP_VOIGT4) ? 4 : 3;
    std::vector<Region> R(n_regions);
    for (int r = 0; r < n_regions; ++r) {
        const long long P = pix_off[r + 1] - pix_off[r];
        if (P < 2 || P > 0x7fffffff) return fail(VAMP_ERR_ARG, "vamp_set_regions: a region needs >= 2 pixels");
        if (n_comp[r] < 1 || n_comp[r] > KMAX) return fail(VAMP_ERR_ARG, "vamp_set_regions: n_comp out of range (1..32)");
        Region d;
        d.pix_off = pix_off[r];
        d.P = (int)P; d.K = n_comp[r]; d.mode = mode; d.q = q; d.sample_sd = sample_sd ? 1 : 0;
        d.D = q * d.K + d.sample_sd;
        d.rng_id = r;
        d.d_before = r ? R[r - 1].d_before + R[r - 1].D : 0;
        const double* xr = x + pix_off[r];
        if (bounds) {
            d.c_lo = bounds[4 * r + 0];
            d.c_hi = bounds[4 * r + 1];
            d.w_max = (mode == VAMP_GAUSS3) ? bounds[4 * r + 2] : bounds[4 * r + 3];
        } else {
            d.c_lo = std::min(xr[0], xr[P - 1]);             // vpfits.py:250
            d.c_hi = std::max(xr[0], xr[P - 1]);
            const double sigma_max = (d.c_hi - d.c_lo) / 2.0;                                          // vpfits.py:320
            d.w_max = (mode == VAMP_GAUSS3) ? sigma_max : sigma_max * 2 * std::sqrt(2 * std::log(2.0)); // :326
        }
        if (!(d.c_hi > d.c_lo) || !(d.w_max > 0)) return fail(VAMP_ERR_ARG, "vamp_set_regions: empty prior range");
        d.lp_c = -std::log(d.c_hi - d.c_lo);
        d.lp_w = -std::log(d.w_max);
        if (mode == VAMP_NBZ3) {
            d.l_fixed = nbz[4 * r + 0]; d.line = nbz[4 * r + 1]; d.x_origin = nbz[4 * r + 2]; d.x_scale = nbz[4 * r + 3];
        }
        double nc = 0.0;
        if (include_norm && !sample_sd) {
            for (long long i = 0; i < P; ++i) {
                const double s = noise[pix_off[r] + i];
                nc += std::log(2.0 * M_PI * s * s);
            }
            nc *= -0.5;
        }
        d.norm_const = nc;
        const bool up = xr[1] > xr[0];
        for (long long i = 1; i < P; ++i) {
            const double dx = xr[i] - xr[i - 1];
            if (!std::isfinite(dx) || dx == 0.0 || (dx > 0.0) != up)
                return fail(VAMP_ERR_ARG, "vamp_set_regions: x must be finite and strictly monotonic within a region");
        }
        R[r] = d;
    }
    {
        std::vector<vamp::plan::RegionShape> shp(n_regions);
        for (int r = 0; r < n_regions; ++r) shp[r] = vamp::plan::RegionShape{R[r].P, R[r].K};
        const std::string err = vamp::plan::plan_classes(shp, c->packing, mode == VAMP_GAUSS3, c->f32, true, c->classes);
        if (!err.empty()) return fail(VAMP_ERR_ARG, "vamp_set_regions: " + err);
    }
    const long long N = pix_off[n_regions];
    c->x.assign(x, x + N);
    c->f.assign(flux, flux + N);
    c->wt.resize(N);
    for (long long i = 0; i < N; ++i) c->wt[i] = sample_sd ? 1.0 : 1.0 / noise[i];
    c->R = R;
    c->mode = mode;
    c->n_regions = n_regions;
    return VAMP_OK;
}

int vamp_region_class(vamp_ctx* c, int region, int* kind, int* n_classes) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_region_class: ctx is NULL");
    if (region < 0 || region >= c->n_regions) return fail(VAMP_ERR_ARG, "vamp_region_class: no such region");
    if (kind) *kind = c->classes.kind[c->classes.class_of[region]];
    if (n_classes) *n_classes = (int)c->classes.kind.size();
    return VAMP_OK;
}

int vamp_set_region_ids(vamp_ctx* c, const int32_t* ids) {
    if (!c || !ids) return fail(VAMP_ERR_ARG, "vamp_set_region_ids: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_set_region_ids: call vamp_set_regions first");
    for (int r = 0; r < c->n_regions; ++r)
        if (ids[r] < 0) return fail(VAMP_ERR_ARG, "vamp_set_region_ids: ids must be >= 0");
    for (int r = 0; r < c->n_regions; ++r) c->R[r].rng_id = ids[r];
    return VAMP_OK;
}

int vamp_region_ndim(vamp_ctx* c, int region, int* ndim) {
    if (!c || !ndim) return fail(VAMP_ERR_ARG, "vamp_region_ndim: NULL argument");
    if (region < 0 || region >= c->n_regions) return fail(VAMP_ERR_ARG, "vamp_region_ndim: no such region");
    *ndim = c->R[region].D;
    return VAMP_OK;
}

int vamp_lnprob(vamp_ctx* c, int region, int64_t W, const double* theta, double* lnprob, double* chi2) {
    if (!c || !theta || !lnprob) return fail(VAMP_ERR_ARG, "vamp_lnprob: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_lnprob: call vamp_set_regions first");
    if (region < 0 || region >= c->n_regions) return fail(VAMP_ERR_ARG, "vamp_lnprob: no such region");
    if (W <= 0) return fail(VAMP_ERR_ARG, "vamp_lnprob: W must be positive");
    lnprob_block(c, region, W, theta, lnprob, chi2);
    return VAMP_OK;
}

int vamp_lnprob_all(vamp_ctx* c, int64_t W, const double* theta, double* lnprob, double* chi2) {
    if (!c || !theta || !lnprob) return fail(VAMP_ERR_ARG, "vamp_lnprob_all: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_lnprob_all: call vamp_set_regions first");
    if (W <= 0) return fail(VAMP_ERR_ARG, "vamp_lnprob_all: W must be positive");
    return lnprob_all_impl(c, W, theta, lnprob, chi2);
}

int vamp_map_all(vamp_ctx* c, const double* theta0, const uint8_t* active, int64_t maxiter, int64_t maxfun, double xtol,
                 double ftol, double* theta_best, double* lnprob_best, double* chi2_best, int64_t* iterations) {
    if (!c || !theta0 || !theta_best || !lnprob_best) return fail(VAMP_ERR_ARG, "vamp_map_all: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_map_all: call vamp_set_regions first");
    if (maxiter < 0 || maxfun < 0 || !(xtol >= 0.0) || !(ftol >= 0.0)) return fail(VAMP_ERR_ARG, "vamp_map_all: bad limits");
    std::vector<int> dims(c->n_regions);
    std::vector<long long> offs(c->n_regions);
    for (int r = 0; r < c->n_regions; ++r) { dims[r] = c->R[r].D; offs[r] = c->R[r].d_before; }
    int rc = vamp::nelder_mead_all(c->n_regions, dims.data(), offs.data(), theta0, active, maxiter, maxfun, xtol, ftol, theta_best,
                                   iterations, [&](int W, const double* th, double* lp) { return lnprob_all_impl(c, W, th, lp, nullptr); });
    if (rc) return rc;
    return lnprob_all_impl(c, 1, theta_best, lnprob_best, chi2_best);
}

int vamp_model(vamp_ctx* c, int region, const double* theta1, double* tau_comp, double* flux_model) {
    if (!c || !theta1) return fail(VAMP_ERR_ARG, "vamp_model: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_model: call vamp_set_regions first");
    if (region < 0 || region >= c->n_regions) return fail(VAMP_ERR_ARG, "vamp_model: no such region");
    const Region& R = c->R[region];
    Line ln[KMAX];
    (void)stage(R, theta1, ln);
    double dtab[vamp::DTAB_N];
    std::vector<double> tau(R.P, 0.0);
    for (int k = 0; k < R.K; ++k) {
        if (R.mode != VAMP_GAUSS3)
            for (int n = 0; n < vamp::DTAB_N; ++n) dtab[n] = vamp::core_dtab_entry(n, ln[k].y);
        for (int i = 0; i < R.P; ++i) {
            const double xi = c->x[R.pix_off + i];
            double tk;
            if (R.mode == VAMP_GAUSS3) {
                const double u = (xi - ln[k].c) * ln[k].s;
                tk = ln[k].amp * std::exp(-0.5 * (u * u));
            } else {
                tk = ln[k].amp * vamp::voigt_Hs(std::fabs(xi - ln[k].c) * ln[k].s, ln[k].y, dtab, ln[k].pole, ln[k].hy);
            }
            if (tau_comp) tau_comp[(long long)k * R.P + i] = tk;
            tau[i] += tk;
        }
    }
    if (flux_model)
        for (int i = 0; i < R.P; ++i) flux_model[i] = std::exp(-tau[i]);
    return VAMP_OK;
}

int vamp_model_all(vamp_ctx* c, const double* theta, double* tau_comp, double* flux_model) {
    if (!c || !theta) return fail(VAMP_ERR_ARG, "vamp_model_all: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_model_all: call vamp_set_regions first");
    long long tau_off = 0;
    for (int r = 0; r < c->n_regions; ++r) {
        const Region& R = c->R[r];
        int rc = vamp_model(c, r, theta + R.d_before, tau_comp ? tau_comp + tau_off : nullptr,
                            flux_model ? flux_model + R.pix_off : nullptr);
        if (rc) return rc;
        tau_off += (long long)R.K * R.P;
    }
    return VAMP_OK;
}

int vamp_line_records(vamp_ctx* c, int region, const double* theta1, double* rec, double* lnprior) {
    if (!c || !theta1 || !rec || !lnprior) return fail(VAMP_ERR_ARG, "vamp_line_records: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_line_records: call vamp_set_regions first");
    if (region < 0 || region >= c->n_regions) return fail(VAMP_ERR_ARG, "vamp_line_records: no such region");
    const Region& R = c->R[region];
    Line ln[KMAX];
    *lnprior = stage(R, theta1, ln);
    for (int k = 0; k < R.K; ++k) {
        rec[5 * k + 0] = ln[k].c; rec[5 * k + 1] = ln[k].s; rec[5 * k + 2] = ln[k].y;
        rec[5 * k + 3] = (R.mode == VAMP_GAUSS3) ? ln[k].amp : ln[k].amp * vamp::SQRT_PI;
        rec[5 * k + 4] = ln[k].pole;
    }
    return VAMP_OK;
}

int vamp_wofz_re(vamp_ctx* c, int64_t n, const double* x, const double* y, double* re_w) {
    if (!c || !x || !y || !re_w || n <= 0) return fail(VAMP_ERR_ARG, "vamp_wofz_re: bad argument");
    for (int64_t i = 0; i < n; ++i) {
        if (c->f32) {
            re_w[i] = (double)vamp::humlicek_w4_re(std::fmin(std::fabs((float)x[i]), vamp::W4_XMAX), (float)y[i]);
        } else {
            double dtab[vamp::DTAB_N];
            for (int k = 0; k < vamp::DTAB_N; ++k) dtab[k] = vamp::core_dtab_entry(k, y[i]);
            re_w[i] = vamp::voigt_H(std::fabs(x[i]), y[i], dtab, vamp::core_pole_factor(y[i]), vamp::core_hy(y[i]));
        }
    }
    return VAMP_OK;
}

// ---- sampler ---------------------------------------------------------------------------------
int vamp_sampler_bind_state(vamp_ctx* c, void* X_dev, void* lnp_dev) {
    if (!c || !X_dev || !lnp_dev) return fail(VAMP_ERR_ARG, "vamp_sampler_bind_state: NULL argument");
    free_sampler(c);
    c->X = (double*)X_dev;       // host memory here
    c->lnp = (double*)lnp_dev;
    c->X_ext = true;
    return VAMP_OK;
}

int vamp_sampler_init(vamp_ctx* c, int64_t W, const double* theta0, uint64_t seed, double a, int32_t split_block) {
    if (!c || !theta0) return fail(VAMP_ERR_ARG, "vamp_sampler_init: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_sampler_init: call vamp_set_regions first");
    if (W < 2 || (W & 1)) return fail(VAMP_ERR_ARG, "vamp_sampler_init: W must be even and >= 2");
    if (split_block < 2 || (split_block & 1) || W % split_block) return fail(VAMP_ERR_ARG, "vamp_sampler_init: split_block must be even and divide W");
    if (!(a > 1.0)) return fail(VAMP_ERR_ARG, "vamp_sampler_init: a must be > 1");
    long long tt = 0;
    for (int r = 0; r < c->n_regions; ++r) {
        c->R[r].theta_off = tt;
        c->R[r].walker_off = (long long)r * W;
        tt += (long long)W * c->R[r].D;
    }
    const bool ext = c->X_ext && c->X;
    if (!ext) free_sampler(c);
    c->W = W; c->total_theta = tt; c->total_walkers = (long long)c->n_regions * W;
    c->split_block = split_block; c->a = a; c->seed = seed; c->step = 0;
    if (!ext) {
        c->X_ext = false;
        c->X_own.resize(tt);
        c->lnp_own.resize(c->total_walkers);
        c->X = c->X_own.data();
        c->lnp = c->lnp_own.data();
    }
    c->nacc.assign(c->total_walkers, 0);
    c->send.clear(); c->recv.clear();
    std::memcpy(c->X, theta0, tt * sizeof(double));
    lnprob_all_impl(c, W, c->X, c->lnp, nullptr);
    c->shard_rank = 0; c->shard_world = 1; c->shard_parts = 1;
    c->part_slots = c->part_stride = 0;
    c->slot_begin = 0;
    c->slot_end = c->total_walkers / 2;
    c->ready = true;
    return VAMP_OK;
}

int vamp_sampler_set_shard_parts(vamp_ctx* c, int rank, int world, int parts, int64_t* own_begin, int64_t* own_end) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_sampler_set_shard: ctx is NULL");
    if (!c->ready) return fail(VAMP_ERR_STATE, "vamp_sampler_set_shard: call vamp_sampler_init first");
    if (world < 1 || rank < 0 || rank >= world) return fail(VAMP_ERR_ARG, "vamp_sampler_set_shard: bad rank/world");
    if (parts < 1 || parts > 64) return fail(VAMP_ERR_ARG, "vamp_sampler_set_shard: parts must be in 1..64");
    if (c->n_regions != 1) return fail(VAMP_ERR_ARG, "vamp_sampler_set_shard: walker sharding is for single-region contexts (shard regions across devices otherwise)");
    if (c->comm && (world != 1 || rank != 0)) return fail(VAMP_ERR_ARG, "vamp_sampler_set_shard: rank/world differ from the communicator's (vamp_comm_init_rank)");
    vamp::plan::ShardPlan sp;                       // csrc/host_plan.hpp: the arithmetic the HIP library runs
    {
        const std::string err = vamp::plan::plan_shard(c->W, c->split_block, rank, world, parts, sp);
        if (!err.empty()) return fail(VAMP_ERR_ARG, "vamp_sampler_set_shard: " + err);
    }
    c->shard_rank = rank; c->shard_world = world; c->shard_parts = parts;
    c->part_slots = sp.part_slots;
    c->part_stride = sp.part_stride;
    c->slot_begin = sp.slot_begin;
    c->slot_end = c->slot_begin + c->part_slots;
    c->send.clear(); c->recv.clear();
    if (world > 1 || c->comm) {
        c->send.assign(vamp::plan::exchange_send_doubles(parts, c->part_slots, c->R[0].D), 0.0);
        c->recv.assign(vamp::plan::exchange_recv_doubles(parts, world, c->part_slots, c->R[0].D), 0.0);
        c->part_step.assign(parts, 0u);
        c->part_half.assign(parts, 0);
    }
    for (int p = 0; p < parts; ++p) {
        if (own_begin) own_begin[p] = sp.own_begin[p];
        if (own_end) own_end[p] = sp.own_end[p];
    }
    return VAMP_OK;
}
int vamp_sampler_set_shard(vamp_ctx* c, int rank, int world, int64_t* own_begin, int64_t* own_end) {
    return vamp_sampler_set_shard_parts(c, rank, world, 1, own_begin, own_end);
}

int vamp_sampler_state_ptrs(vamp_ctx* c, void** X_dev, void** lnp_dev, int64_t* total_theta, int64_t* total_walkers) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_sampler_state_ptrs: ctx is NULL");
    if (!c->ready) return fail(VAMP_ERR_STATE, "vamp_sampler_state_ptrs: call vamp_sampler_init first");
    if (X_dev) *X_dev = c->X;
    if (lnp_dev) *lnp_dev = c->lnp;
    if (total_theta) *total_theta = c->total_theta;
    if (total_walkers) *total_walkers = c->total_walkers;
    return VAMP_OK;
}

int vamp_sampler_half_step(vamp_ctx* c, int half) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step: ctx is NULL");
    if (!c->ready) return fail(VAMP_ERR_STATE, "vamp_sampler_half_step: call vamp_sampler_init first");
    if (half != 0 && half != 1) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step: half must be 0 or 1");
    half_step_all(c, half);
    if (half == 1) c->step += 1;
    return VAMP_OK;
}

int vamp_sampler_half_step_part(vamp_ctx* c, int half, int part) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_part: ctx is NULL");
    if (!c->ready) return fail(VAMP_ERR_STATE, "vamp_sampler_half_step_part: call vamp_sampler_init first");
    if (half != 0 && half != 1) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_part: half must be 0 or 1");
    if (part < 0 || part >= c->shard_parts) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_part: no such part");
    if (c->comm && !c->send.empty())
        return fail(VAMP_ERR_STATE, "vamp_sampler_half_step_part: with a communicator the exchange is part of vamp_sampler_half_step / vamp_sampler_run");
    half_step_part(c, half, part);
    if (half == 1 && part == c->shard_parts - 1) c->step += 1;
    return VAMP_OK;
}

int vamp_sampler_half_step_ext(vamp_ctx* c, int region, int64_t n, const int32_t* active_idx, const int32_t* partner_idx,
                               const double* zz, const double* logu) {
    if (!c || !active_idx || !partner_idx || !zz || !logu) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_ext: NULL argument");
    if (!c->ready) return fail(VAMP_ERR_STATE, "vamp_sampler_half_step_ext: call vamp_sampler_init first");
    if (region < 0 || region >= c->n_regions) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_ext: no such region");
    if (n <= 0 || n > c->W) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_ext: bad n");
    std::vector<char> is_active(c->W, 0);
    for (int64_t i = 0; i < n; ++i) {
        if (active_idx[i] < 0 || active_idx[i] >= c->W || partner_idx[i] < 0 || partner_idx[i] >= c->W)
            return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_ext: walker index out of range");
        if (is_active[active_idx[i]]) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_ext: duplicate active walker");
        is_active[active_idx[i]] = 1;
        if (!(zz[i] > 0.0)) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_ext: stretch factor must be positive");
    }
    for (int64_t i = 0; i < n; ++i)
        if (is_active[partner_idx[i]]) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_ext: partner must belong to the frozen complement");
    const Region& R = c->R[region];
#pragma omp parallel for schedule(dynamic, 4) num_threads(c->threads)
    for (int64_t i = 0; i < n; ++i) move_one(c, R, active_idx[i], partner_idx[i], zz[i], logu[i], nullptr);
    return VAMP_OK;
}

int vamp_sampler_run_dev(vamp_ctx* c, int64_t n_steps, int thin, double* chain_dev, double* lnprob_chain_dev, double* seconds) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_sampler_run_dev: ctx is NULL");
    if (!c->ready) return fail(VAMP_ERR_STATE, "vamp_sampler_run_dev: call vamp_sampler_init first");
    if (n_steps < 0 || thin < 1) return fail(VAMP_ERR_ARG, "vamp_sampler_run_dev: n_steps >= 0 and thin >= 1 required");
    if (c->shard_world != 1 && !(c->comm && !c->send.empty()))
        return fail(VAMP_ERR_STATE, "vamp_sampler_run_dev: a sharded context without a communicator is stepped by the host "
                                    "(half_step_part + pack_get / scatter_put)");
    const long long n_keep = n_steps / thin;
    const auto t0 = std::chrono::steady_clock::now();
    long long kept = 0;
    for (long long it = 0; it < n_steps; ++it) {
        half_step_all(c, 0);
        half_step_all(c, 1);
        c->step += 1;
        if ((it + 1) % thin == 0 && kept < n_keep) {
            if (chain_dev) std::memcpy(chain_dev + kept * c->total_theta, c->X, c->total_theta * sizeof(double));
            if (lnprob_chain_dev) std::memcpy(lnprob_chain_dev + kept * c->total_walkers, c->lnp, c->total_walkers * sizeof(double));
            ++kept;
        }
    }
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return VAMP_OK;
}

int vamp_sampler_run(vamp_ctx* c, int64_t n_steps, int thin, double* chain, double* lnprob_chain, int64_t* n_accept, double* seconds) {
    int rc = vamp_sampler_run_dev(c, n_steps, thin, chain, lnprob_chain, seconds);      // "device" memory is host memory here
    if (rc) return rc;
    if (n_accept) std::memcpy(n_accept, c->nacc.data(), c->total_walkers * sizeof(long long));
    return VAMP_OK;
}

int vamp_sampler_get_state(vamp_ctx* c, double* theta, double* lnprob, int64_t* n_accept, int64_t* step) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_sampler_get_state: ctx is NULL");
    if (!c->ready) return fail(VAMP_ERR_STATE, "vamp_sampler_get_state: call vamp_sampler_init first");
    if (theta) std::memcpy(theta, c->X, c->total_theta * sizeof(double));
    if (lnprob) std::memcpy(lnprob, c->lnp, c->total_walkers * sizeof(double));
    if (n_accept) std::memcpy(n_accept, c->nacc.data(), c->total_walkers * sizeof(long long));
    if (step) *step = c->step;
    return VAMP_OK;
}

int vamp_sampler_set_state(vamp_ctx* c, const double* theta, const double* lnprob, int64_t step) {
    if (!c || !theta || !lnprob) return fail(VAMP_ERR_ARG, "vamp_sampler_set_state: NULL argument");
    if (!c->ready) return fail(VAMP_ERR_STATE, "vamp_sampler_set_state: call vamp_sampler_init first");
    if (step < 0) return fail(VAMP_ERR_ARG, "vamp_sampler_set_state: step must be >= 0");
    std::memcpy(c->X, theta, c->total_theta * sizeof(double));
    std::memcpy(c->lnp, lnprob, c->total_walkers * sizeof(double));
    c->step = step;
    return VAMP_OK;
}

// ---- multi-device entry points: the host build has no RCCL; a communicator of ONE rank is accepted
//      so that the single-rank rehearsal of the exchange runs through the same call sequence ----
int vamp_comm_unique_id(char* id) {
    if (!id) return fail(VAMP_ERR_ARG, "vamp_comm_unique_id: id is NULL");
    std::memset(id, 0, VAMP_COMM_ID_BYTES);
    std::memcpy(id, "vamp-cpu", 8);
    return VAMP_OK;
}
int vamp_comm_init_rank(vamp_ctx* c, const char* id, int rank, int world) {
    if (!c || !id) return fail(VAMP_ERR_ARG, "vamp_comm_init_rank: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(VAMP_ERR_ARG, "vamp_comm_init_rank: bad rank/world");
    if (c->comm) return fail(VAMP_ERR_STATE, "vamp_comm_init_rank: the context already has a communicator");
    if (world != 1) return fail(VAMP_ERR_COMM, "vamp_comm_init_rank: the host build has no RCCL (exchange through vamp_sampler_pack_get / scatter_put)");
    c->comm = true;
    return VAMP_OK;
}
int vamp_comm_library(char* path, int64_t capacity) {
    // the host build has no RCCL: its one-rank "communicator" is this library itself
    if (!path || capacity < 2) return fail(VAMP_ERR_ARG, "vamp_comm_library: no room for a path");
    Dl_info info;
    const char* name = (dladdr(reinterpret_cast<void*>(&vamp_comm_library), &info) && info.dli_fname) ? info.dli_fname : "";
    std::snprintf(path, (size_t)capacity, "%s", name);
    return VAMP_OK;
}
int vamp_comm_destroy(vamp_ctx* c) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_comm_destroy: ctx is NULL");
    c->comm = false;
    return VAMP_OK;
}
int vamp_comm_info(vamp_ctx* c, int* rank, int* world, int* queried) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_comm_info: ctx is NULL");
    if (!c->comm) return fail(VAMP_ERR_STATE, "vamp_comm_info: the context has no communicator (vamp_comm_init_rank)");
    if (rank) *rank = 0;
    if (world) *world = 1;
    if (queried) *queried = 0;
    return VAMP_OK;
}

int vamp_sampler_pack_get(vamp_ctx* c, int part, double* rows) {
    if (!c || !rows) return fail(VAMP_ERR_ARG, "vamp_sampler_pack_get: NULL argument");
    if (!c->ready || c->send.empty()) return fail(VAMP_ERR_STATE, "vamp_sampler_pack_get: no sharded sampler (vamp_sampler_set_shard_parts with world > 1)");
    if (part < 0 || part >= c->shard_parts) return fail(VAMP_ERR_ARG, "vamp_sampler_pack_get: no such part");
    const size_t n = (size_t)c->part_slots * (c->R[0].D + 1);
    std::memcpy(rows, c->send.data() + (size_t)part * n, n * sizeof(double));
    return VAMP_OK;
}
int vamp_sampler_scatter_put(vamp_ctx* c, int part, const double* rows_all) {
    if (!c || !rows_all) return fail(VAMP_ERR_ARG, "vamp_sampler_scatter_put: NULL argument");
    if (!c->ready || c->recv.empty()) return fail(VAMP_ERR_STATE, "vamp_sampler_scatter_put: no sharded sampler (vamp_sampler_set_shard_parts with world > 1)");
    if (part < 0 || part >= c->shard_parts) return fail(VAMP_ERR_ARG, "vamp_sampler_scatter_put: no such part");
    scatter_part(c, part, rows_all);
    return VAMP_OK;
}

// ---- test hooks (not part of the ABI: no vamp_ prefix): the launch-plan arithmetic of csrc/host_plan.hpp that no host
//      entry point reaches, so that tests/test_cpu_boundary.py -- and its sanitizer run -- can drive it -----------------
long long vampdbg_xcd_map(long long b, long long n_regions, long long bpr) { return vamp::plan::xcd_map(b, n_regions, bpr); }
void vampdbg_packed_grid(long long n_regions, long long half_w, int subs, int waves_per_block, long long* out3) {
    const vamp::plan::PackedGrid g = vamp::plan::plan_packed_grid(n_regions, half_w, subs, waves_per_block);
    out3[0] = g.wpr; out3[1] = g.grid; out3[2] = g.bpr;
}
int vampdbg_resident_class_ok(int kind, long long half_w, int compute_waves, int walkers_per_wave, int automatic) {
    return vamp::plan::resident_class_ok(kind, half_w, compute_waves, walkers_per_wave, automatic != 0) ? 1 : 0;
}

int vamp_exchange_timing(vamp_ctx* c, double* total_ms, int64_t* exchanges) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_exchange_timing: ctx is NULL");
    if (total_ms) *total_ms = 0.0;          // the host build has no in-library exchange
    if (exchanges) *exchanges = 0;
    return VAMP_OK;
}

int vamp_kernel_timing(vamp_ctx* c, int enable, double* total_ms, int64_t* launches) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_kernel_timing: ctx is NULL");
    if (total_ms) *total_ms = c->timing_ms;
    if (launches) *launches = c->timing_launches;
    c->timing_ms = 0.0;
    c->timing_launches = 0;
    c->timing = enable != 0;
    return VAMP_OK;
}

}  // extern "C"
