// vamp_hip.hip -- MI355X (gfx950) kernels + C ABI for VAMP's MCMC hot path.
//
// Replaces, below the Python surface of class VPfit (reference vamp_1.0/vpfits.py:33):
//   * the per-proposal evaluation of the PyMC model graph -- profile closures
//     (vpfits.py:254-260, 299-305), `total` = Tau2flux(sum(profiles)) (vpfits.py:334-336,
//     physics.py:105), the observed Normal likelihood (vpfits.py:39,341) and the priors
//     (vpfits.py:239-252, 283-297);
//   * the sampler loop of mcmc_fit / find_bic (vpfits.py:361-395, 420-425), by the stretch move
//     of Goodman & Weare (2010) with emcee's red/blue semantics.
//
// Execution shape (struct Pack): a walker is served by one 64-lane wavefront, by a quarter of one
// (short regions), or by a 4-wavefront workgroup whose wavefronts take every 4th 256-pixel tile
// (long regions: the headline).
//   stage   lanes 0..K-1 turn the walker's parameters into per-line records in LDS
//           (centroid, x-scale, damping y, tau scale, pole factor) and all lanes fill the per-line
//           table 1/(u_n^2 + y^2) of the near-axis Voigt rule; priors are summed here.  A
//           workgroup then builds the Taylor tables of the line cores (struct LineTables).
//   sweep   lane l takes pixels l, l+64, ... of a tile: coalesced 512-byte reads of x / flux /
//           1/sigma (shared by every walker -> L2/MALL resident); near lines are evaluated per
//           pixel, all distant lines of the tile through ONE Chebyshev interpolant (ff_*);
//           tau in registers, flux = exp(-tau), chi^2 partial per lane.
//   reduce  xor-shuffle tree over the wavefront (+ LDS across the workgroup); lane 0 owns the result.
//   move    (sampler kernel) proposal q = c - (c - s) z is formed in LDS before `stage`, and the
//           accept test / state update follow `reduce` in the same launch.
// Kernels: k_lnprob / k_half_step (one launch per evaluation / per half-step of the stretch move), k_draws (the draws
// of packed launches), k_run_resident (small ensembles: one workgroup per region runs the region's WHOLE step loop),
// k_map_search (one workgroup per region runs its WHOLE Nelder-Mead MAP search, vpfits.py:352-358, 426), k_model,
// k_scatter_rows (walker-sharded runs), k_wofz / k_line_records (test hooks).
// No MFMA (nothing here is a contraction), no atomics, no inter-workgroup communication.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/vamp_hip.h"
#include "host_plan.hpp"
#include "map_search.hpp"
#include "voigt_math.hpp"

#ifndef VAMP_EARLY_LOADS
#define VAMP_EARLY_LOADS 1
#endif
#ifndef VAMP_BLEND_PASS_LINES
#define VAMP_BLEND_PASS_LINES 4
#endif
#ifndef VAMP_MID_MIN_K
#define VAMP_MID_MIN_K 3
#endif
#ifndef VAMP_MID_MIN_P
#define VAMP_MID_MIN_P 96
#endif
#ifndef VAMP_X_PREFETCH
#define VAMP_X_PREFETCH 0
#endif

#ifndef VAMP_MIN_WAVES
#define VAMP_MIN_WAVES 3
#endif
#ifndef VAMP_EARLY_LNP
#define VAMP_EARLY_LNP 1
#endif
#ifndef VAMP_SR_EARLY
#define VAMP_SR_EARLY 0
#endif
#ifndef VAMP_XCD_MAP
#define VAMP_XCD_MAP 0        // region -> XCD mapping of multi-region launches: measured, off (k_half_step; profiles/r04_d_xcd_mapping.txt)
#endif
#ifndef VAMP_WIDE_NODES
#define VAMP_WIDE_NODES 1      // lines far wider than a tile join the tile's interpolant (sweep_range_ff)
#endif
#ifndef VAMP_LPT
#define VAMP_LPT 1             // small ensembles: the regions of a launch class in order of decreasing work
#endif
#ifndef VAMP_FLUX_EXP_DROP
#define VAMP_FLUX_EXP_DROP 2   // far-field sweep: exp(-tau) of the model flux by the degree-11 kernel (6e-15) instead of degree 13
#endif
#ifndef VAMP_FF_R2_M4
#define VAMP_FF_R2_M4 100.0    // far-field node values: below this |z|^2 the 6-level fraction, ...
#define VAMP_FF_R2_M3 300.0    // ... below this the 4-level, ...
#define VAMP_FF_R2_M2 2000.0   // ... below this the 3-level, beyond it the 2-level one
#endif
#ifndef VAMP_F32_FAR_CORE
#define VAMP_F32_FAR_CORE 1    // fp32 far field: far = outside the Gaussian core (|z|^2 >= VAMP_MID_Z2) instead of outside |z| < 8
#endif
#ifndef VAMP_CORE_FASTPATH
#define VAMP_CORE_FASTPATH 1   // tile_voigt: a wavefront wholly inside |z| < 8 of a line evaluates its pixels in straight-line code
#endif
#ifndef VAMP_MID_NODES
#define VAMP_MID_NODES 1       // lines >= FF_DIST half-widths beyond a tile whose |z| < 8 zone still reaches into it join the interpolant too ...
#endif
#ifndef VAMP_MID_Z2
#define VAMP_MID_Z2 30.25      // ... when the whole tile is outside |z|^2 < this (e^{-30.25} = 7e-14 of the line's peak)
#endif
#ifndef VAMP_WIDE_MAX
#define VAMP_WIDE_MAX 0.75     // ... when the tile's half-width is at most this many units of the line's z.  Worst relative lnprob
#endif                         // error at the switch on well-fitted data at S/N 200 (tests/wide_probe.py, profiles/r04_e_wide_lines.txt):
                               // 0.5: 4e-16, 0.75: 4e-14, 1.0: 1.2e-12 (3e-11 with amplitudes of 20), 1.25: 7e-11; the bar is 1e-9
#ifndef VAMP_F32_LEAN_STAGE
#define VAMP_F32_LEAN_STAGE 1
#endif
namespace {

constexpr int KMAX = 16;                       // lines per region of the fast shapes (far field, Taylor tables, packing)
constexpr int KMAX_ALL = VAMP_MAX_COMPONENTS;  // lines per region of the ABI: 17 .. 32 run the plain shape PackXL
static_assert(KMAX_ALL >= KMAX && KMAX_ALL < 64, "lane k stages line k; one spare lane for sd");
constexpr int WAVES_PER_BLOCK = 4;
constexpr long long PACK_MIN_WALKERS = 16384;   // movers per launch from which the packed shapes fill the chip
constexpr int RES_MAX_MOVERS = 128;             // movers per half-step (W / 2) a resident workgroup serves; also: "a small ensemble"
#ifndef VAMP_RES_WAVES
#define VAMP_RES_WAVES 8
#endif
constexpr int RES_MAX_WAVES = VAMP_RES_WAVES;   // compute wavefronts of a resident workgroup (+ the one that draws)
constexpr int BLOCK = 64 * WAVES_PER_BLOCK;
constexpr int DMAX = 4 * KMAX_ALL + 1;
// automatic choice of a 4-wave workgroup per walker (shared Taylor tables of the line cores, see
// LineTables): regions that give every wave >= 2 tiles.  The choice must not depend on the launch
// size: a shard of an ensemble has to run the same arithmetic as the whole ensemble on one GPU.
#ifndef VAMP_SPLIT_MAX_WALKERS
#define VAMP_SPLIT_MAX_WALKERS (1ll << 62)
#endif
constexpr long long SPLIT_MAX_WALKERS = VAMP_SPLIT_MAX_WALKERS;

constexpr double C_LIGHT = 2.98e8;     // physics.py:3 (the reference's value)
constexpr double SIGMA0 = 0.0263;      // physics.py:4
constexpr double SQRT_LN2 = 0.83255461115769775635;
constexpr double SQRT_PI = 1.77245385090551602730;
constexpr double FWHM_PER_SIGMA = 2.35482004503094938202;   // 2 sqrt(2 ln 2), vpfits.py:88,326
constexpr double NEG_INF = -__builtin_huge_val();

// ---------------------------------------------------------------------------------------
// device-side description of one region (one posterior)
// ---------------------------------------------------------------------------------------
struct RegionDev {
    long long pix_off;     // into x / flux / wt
    long long theta_off;   // doubles: start of this region's [W, D] block in the sampler state
    long long walker_off;  // first global walker id of this region in the sampler state
    long long d_before;    // sum of D over the preceding regions (vamp_lnprob_all: block r starts at W * d_before)
    long long tau_off;     // sum of K * P over the preceding regions (vamp_model_all: this region's tau_comp block)
    long long sim_off;     // sum of (D + 1) * D over the preceding regions (k_map_search: this region's simplex)
    int P, K, mode, D;     // D = q*K (+1 if sample_sd)
    int sample_sd, q, rng_id, pad1;   // rng_id: the region's identity in the draw keys (default: its index)
    double c_lo, c_hi;     // centroid prior (vpfits.py:250,293)
    double w_max;          // sigma_max (GAUSS3, vpfits.py:320) or fwhm_max (vpfits.py:326)
    double lp_c, lp_w;     // -log(c_hi - c_lo), -log(w_max): uniform log-densities
    double l_fixed, line, x_origin, x_scale;   // NBZ3
    double norm_const;     // -1/2 sum log(2 pi sigma^2) or 0
    double tile_span;      // 64 * TPIX * (largest pixel spacing of the region)
};

struct LineRec {           // per (walker, component), lives in LDS
    double c;              // centroid
    double s;              // Voigt: 2 sqrt(ln2)/G ; Gauss: 1/sigma
    double y;              // Voigt: L sqrt(ln2)/G
    double amp;            // Voigt: A y (tau_k = A y sqrt(pi) H: the evaluators return sqrt(pi) H) ; Gauss: A
    double pole;           // core_pole_factor(y)
    double hy;             // core_hy(y)
    double xcap;           // +inf, or X_FAR when one tile of pixels spans > 16 units of |z| (narrow line)
    double w8, w25;        // half-widths, in x, of |z|^2 < 64 and |z|^2 < 625 around the centre (far-field classification)
    double wmid;           // ... and of |z|^2 < VAMP_MID_Z2 (the Gaussian core's reach)
};

// Packing of walkers onto wavefronts.  LPW lanes serve one walker (SUBS = 64/LPW walkers share a
// wave); KCAP bounds the lines per walker and so the LDS footprint.  <64,16> is the headline shape
// (thousands of pixels per region); <16,8> serves the 9..478-pixel regions of real spectra, where
// one walker cannot fill a wave and the per-walker fixed work (staging, draws, reduction) dominates.
//
// SPLIT: the full tiles of a region are dealt round-robin into PARTS classes and chi^2 is summed
// class by class (so the summation order does not depend on the packing).  SPLIT = false: one wavefront
// sweeps all classes of its walker.  SPLIT = true: the workgroup serves ONE walker and wavefront j
// sweeps class j -- four times as many, four times shorter wavefronts, for launches with too few
// walkers to fill 256 CUs evenly (a shard of the headline ensemble on one of 8 GPUs is 4096
// walkers per half-step = 1.3 wavefronts per SIMD slot; the drain at the end of a launch costs
// about half a wavefront lifetime).  The group shares one set of line records and tables.
#ifndef VAMP_PARTS
#define VAMP_PARTS 4
#endif
constexpr int PARTS = VAMP_PARTS;
template <int KCAP, bool OWN_DTAB> struct WalkerLds;
// Timing-only builds (-DVAMP_STAMPS, tools/stamps.py): the first lane of workgroup 0 records (tag, shader clock) pairs
// along one walker's path through a half-step -- where a latency-bound wavefront's time goes
#ifdef VAMP_STAMPS
__device__ unsigned long long g_stamps[2 * 8192];
__device__ unsigned int g_stamp_n;
#define VAMP_STAMP(tag)                                                                          \
    do {                                                                                         \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                               \
            const unsigned int n_ = g_stamp_n;                                                   \
            if (n_ < 8192) { g_stamps[2 * n_] = (tag); g_stamps[2 * n_ + 1] = __builtin_amdgcn_s_memtime(); g_stamp_n = n_ + 1; } \
        }                                                                                        \
    } while (0)
#define VAMP_STAMP_AFTER(tag, v) do { asm volatile("" :: "v"(v)); VAMP_STAMP(tag); } while (0)     // once `v` has arrived
#else
#define VAMP_STAMP(tag) do {} while (0)
#define VAMP_STAMP_AFTER(tag, v) do {} while (0)
#endif
#ifndef VAMP_SPLIT_WAVES
#define VAMP_SPLIT_WAVES 4
#endif
#ifndef VAMP_SPLIT_WAVES_F32
#define VAMP_SPLIT_WAVES_F32 4
#endif
#ifndef VAMP_F32_TABLES
#define VAMP_F32_TABLES 1
#endif
template <int LPW_, int KCAP_, bool TAIL_, int WPB_, bool SPLIT_ = false, bool TABS_ = SPLIT_, bool FF_ = (LPW_ == 64)>
struct Pack {
    static constexpr int LPW = LPW_, KCAP = KCAP_, SUBS = 64 / LPW_;
    static constexpr bool TAIL = TAIL_;     // false: every region is a whole number of full tiles
    static constexpr bool SPLIT = SPLIT_;
    static constexpr bool TABS = TABS_;     // per-walker Taylor tables of the line cores (fp64 Voigt)
    static constexpr bool FF = FF_;         // far-field interpolant in full tiles
    static constexpr int WPB = WPB_;        // wavefronts per workgroup
    static constexpr int THREADS = 64 * WPB_;
    static constexpr int WALKERS_PER_BLOCK = SPLIT_ ? 1 : WPB_ * SUBS;
    // wavefronts per SIMD the register allocation aims for (MIN_WAVES below): 3 (<= 168 VGPRs), 4 (128 VGPRs,
    // 40 KB of LDS) for the workgroup-per-walker shape with the far field, 2 for a blend shape with every
    // line's tables resident
    // the blend shape keeps the Taylor tables of only LINES_PER_PASS lines in LDS at a time (0: of all
    // KCAP lines) and sweeps its pixels once per pass with the optical depths held in registers:
    // 9 KB instead of 18 KB of tables per walker, 3 instead of 1.7 wavefronts per SIMD
    static constexpr int LINES_PER_PASS = (SPLIT_ && TABS_ && !FF_ && WPB_ == 1) ? VAMP_BLEND_PASS_LINES : 0;
    static constexpr int TAB_LINES = LINES_PER_PASS > 0 ? LINES_PER_PASS : KCAP_;
    // the workgroup-per-walker shape with every line's tables resident needs the near-axis tables (dtab) only
    // until the Taylor rows are built from them: they live in the tail of the tables' own space (5.6 KB
    // less LDS: 39.7 KB per workgroup)
    static constexpr bool DTAB_IN_TABLES = SPLIT_ && TABS_ && FF_;
    using Lds = WalkerLds<KCAP_, !DTAB_IN_TABLES>;
    static constexpr int MIN_WAVES = (SPLIT_ && TABS_ && !FF_ && LINES_PER_PASS == 0) ? 2 : DTAB_IN_TABLES ? VAMP_SPLIT_WAVES : VAMP_MIN_WAVES;
    static_assert(!SPLIT_ || (LPW_ == 64 && (WPB_ == PARTS || !FF_) && WPB_ <= PARTS),
                  "a split workgroup is PARTS wavefronts on one walker (tile classes), or up to PARTS wavefronts without far field (contiguous shares)");
};
using PackWide = Pack<64, KMAX, true, WAVES_PER_BLOCK>;
using PackWideFull = Pack<64, KMAX, false, WAVES_PER_BLOCK>;   // the headline shape: no tail code
using PackSplit = Pack<64, KMAX, true, PARTS, true>;
using PackSplitFull = Pack<64, KMAX, false, PARTS, true>;
// 8 walkers x 3.7 KB of LDS per 128-thread workgroup: 5 workgroups (10 waves) per CU
using PackSmall = Pack<16, 8, true, 2>;
// the same for regions of one or two lines (most regions of a real spectrum): 0.9 KB instead of
// 3.7 KB of LDS per walker, so that registers, not LDS, set the occupancy
#ifndef VAMP_SMALL2_WAVES
#define VAMP_SMALL2_WAVES 2
#endif
#ifndef VAMP_SMALL2_LANES
#define VAMP_SMALL2_LANES 8
#endif
using PackSmall2 = Pack<VAMP_SMALL2_LANES, 2, true, VAMP_SMALL2_WAVES>;
// one walker per small workgroup WITH its own Taylor tables (<= 8 lines: 18 KB + 3.7 KB of LDS),
// no far field; the group's wavefronts share the staging and take contiguous shares of the pixels: the blended regions of real spectra (3..8 lines,
// ~100..500 px), where every pixel lies in some line's core and the near-axis rule (~190 issue
// slots per evaluation against ~40 through a table) is the whole cost
#ifndef VAMP_MID_WAVES
#define VAMP_MID_WAVES 1
#endif
#ifndef VAMP_MID_PREDRAW
#define VAMP_MID_PREDRAW 1
#endif
using PackMid = Pack<64, 8, true, VAMP_MID_WAVES, true, true, false>;
// regions of 17 .. 32 lines (the reference sets no limit, vpspectrum.py:287-294): one walker per wavefront, every
// line evaluated per pixel, no far field and no tables; 15 KB of LDS per walker, two walkers per workgroup
using PackXL = Pack<64, KMAX_ALL, true, 2, false, false, false>;

constexpr int FF_NODES = 16;          // Chebyshev nodes of the far-field interpolant of one tile
#ifndef VAMP_FF_DIST
#define VAMP_FF_DIST 2.0        // (round 4: was 4.  The series reproduce a wing from 2 half-widths to 3e-11 of ITS value there,
                                //  tests/test_ff_matrix.py; on well-fitted data with strong damped wings the log-posterior moves
                                //  by 6e-15, tests/ff_probe.py; one tile fewer per side where a narrow line is near: -2.5 %)
#endif
constexpr double FF_DIST = VAMP_FF_DIST;   // a line is "far" from a tile when it lies >= FF_DIST half-widths beyond its edge
#include "ff_matrix.inc"               // FF_M, FF_DEG, FF_ROWS, FF_MAT (tools/gen_ff_matrix.py)
#include "ff_matrix32.inc"             // FF32_M, FF32_NODES, FF32_DEG, FF32_ROWS, FF32_MAT: the fp32 contexts' 8-node form

template <int KCAP, bool OWN_DTAB = true>
struct WalkerLds {
    double theta[4 * KCAP + 4];
    LineRec line[KCAP];
    double dtab[OWN_DTAB ? KCAP : 1][vamp::DTAB_N];     // !OWN_DTAB: in the tail of the Taylor tables (Pack::DTAB_IN_TABLES)
    float linef[KCAP][4];  // fp32 path: c, s, y, amp
};
struct alignas(16) TileScratch {   // per wavefront: far-field working set of the tile in flight
    double coef[FF_ROWS];   // optical depth of the far lines at the tile's 16 Chebyshev nodes, then (every lane has
                            // read the node values by then) the tile's four local power series, one per quarter,
                            // 14 coefficients each; fp32 contexts store those as floats in the same space
    int farlist[KMAX];      // lines treated through the far field, as byte offsets of their records (fp64: LineRec,
                            // fp32: linef rows) -- the list is read four times per tile, an index would cost a
                            // quarter-rate 32-bit multiply each time
    int widelist[KMAX];     // near lines far WIDER than the tile (fp64 tables shape): line indices
};
// Taylor tables of the near-axis zone of every line of ONE walker (voigt_math.hpp): 28 KiB, which
// only a workgroup that serves a single walker can afford (4 workgroups per CU).
template <int NDBL>
struct alignas(16) LineTables { double a[NDBL]; };
template <bool F32, int MODE, class PK>
constexpr bool use_tables() { return PK::TABS && !F32 && MODE != VAMP_GAUSS3; }
// fp32 contexts, workgroup-per-walker shape: the line cores through single-precision Taylor rows (voigt_math.hpp,
// TAB32_*) instead of Humlicek's regions III / IV.  The space holds the rows of all KMAX lines (8 KB) followed by
// the near-axis tables (5.6 KB of doubles) the rows' centre values are computed from.
template <bool F32, int MODE, class PK>
constexpr bool use_tables32() { return VAMP_F32_TABLES && PK::DTAB_IN_TABLES && F32 && MODE != VAMP_GAUSS3; }
constexpr int TAB32_DOUBLES = KMAX * vamp::TAB32_LINE / 2;       // the float rows, counted in doubles
// fp32 contexts, blend shape (one wavefront per walker, <= 8 lines over <= 512 pixels): the fp32 rows of ALL its
// lines resident (512 B per line), built from the walker's own near-axis tables
template <bool F32, int MODE, class PK>
constexpr bool use_blend32() { return VAMP_F32_TABLES && F32 && PK::LINES_PER_PASS > 0 && MODE != VAMP_GAUSS3; }
template <bool F32, int MODE, class PK>
constexpr int table_doubles() {
    return use_tables<F32, MODE, PK>() ? PK::TAB_LINES * vamp::TAB_LINE
           : use_tables32<F32, MODE, PK>() ? TAB32_DOUBLES + KMAX * vamp::DTAB_N
           : use_blend32<F32, MODE, PK>() ? PK::KCAP * vamp::TAB32_LINE / 2 : 2;
}
// wavefronts per SIMD the register allocation aims for: fp32 instructions issue in 2 cycles on a SIMD but one
// wavefront can issue only every 4, so the fp32 form of the workgroup-per-walker shape wants MORE resident
// wavefronts than the fp64 one (it has no Taylor tables: 12 KB of LDS per workgroup)
template <bool F32, class PK>
constexpr int min_waves() { return (F32 && PK::DTAB_IN_TABLES) ? VAMP_SPLIT_WAVES_F32 : PK::MIN_WAVES; }

// the near-axis table of line k as the evaluators want it (unused by the table look-ups; not addressable
// through L when it lives in the tables' space)
template <class PK>
__device__ __forceinline__ const double* dtab_row(const typename PK::Lds& L, int k) {
    if constexpr (PK::DTAB_IN_TABLES) return nullptr;
    else return L.dtab[k];
}

// barrier over the lanes that stage and sweep one walker together
// (a split group of ONE wavefront -- the blend shape -- synchronises like any single wavefront: its lanes' LDS
//  operations execute in program order, so the device-resident step loop can run several such walkers, one per
//  wavefront, in one workgroup)
template <class PK>
__device__ __forceinline__ void group_barrier() {
    if constexpr (PK::SPLIT && PK::WPB > 1) __syncthreads();
    else __builtin_amdgcn_wave_barrier();
}
using WaveLds = WalkerLds<KMAX_ALL, true>;  // the one-walker-per-wavefront kernels (k_model, k_line_records): = PackXL::Lds

// ---------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------
template <int LPW = 64>
__device__ __forceinline__ double wave_sum(double v) {      // sum over the LPW lanes of one walker
#pragma unroll
    for (int off = LPW / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// vpfits.py:239-244: -inf for v < 0, else log(v*exp(-v)).  For v <= 700 that is log(v) - v to an
// ulp of the result (one transcendental instead of two); beyond, exp(-v) leaves the normal range
// and the reference's literal form -- which ends in log(0) = -inf from v ~ 745 -- is kept.
__device__ __forceinline__ double xexp_logp(double v) {
    if (!(v >= 0.0) || !isfinite(v)) return NEG_INF;
    if (v <= 700.0) return log(v) - v;
    return log(v * exp(-v));
}
// the same in fp32 contexts: the logarithm in single precision (~1e-7 of a term of order 1 in a log-posterior whose
// chi^2 carries W4's 1e-4; the stated tolerance is 1e-3 of chi^2), v itself subtracted in double
__device__ __forceinline__ double xexp_logp32(double v) {
    if (!(v >= 0.0) || !isfinite(v)) return NEG_INF;
    if (v <= 700.0) return (double)__logf((float)v) - v;
    return log(v * exp(-v));
}
__device__ __forceinline__ double uniform_logp(double v, double lo, double hi, double lp) {
    return (v >= lo && v <= hi) ? lp : NEG_INF;
}

// Turn theta (in LDS) into line records + prior.  Returns log-prior on every lane.
// MODE is a compile-time parameter: one specialised kernel per parameterisation, no mode
// branches in the staging code or in the pixel loop.
template <int MODE, class PK = PackWide, bool TAB = false, bool TAB32 = false, bool DT32 = false>
__device__ __forceinline__ double stage_lines(const RegionDev& R, typename PK::Lds& L, int lane, bool want_f32, int part,
                                              double* tab = nullptr) {
    // `lane` is the lane index inside the walker's group (0 .. LPW-1).  In a split workgroup
    // (`part` = wavefront index) the FIRST wavefront evaluates the records and the prior and leaves the
    // prior in a spare slot of the parameter block for the others (each needs it to decide whether to
    // sweep); the per-line tables are filled by all 256 threads.  (All four used to evaluate them: the
    // kernel is bound by VALU throughput, and the staging -- divisions, square roots, a logarithm per
    // line -- was 3 % of its instructions three times over.)
    const bool writer = !PK::SPLIT || part == 0;
    constexpr int LP_SLOT = 4 * PK::KCAP + 3;       // theta holds at most 4 KCAP + 1 parameters
    constexpr int WIDE_SLOT = 4 * PK::KCAP + 2;     // far-field shapes: bit k set = line k is far wider than a tile (ff_wide_nodes)
    bool my_wide = false;
    double lp = 0.0;
    const int K = R.K;
    constexpr int Q = (MODE == VAMP_VOIGT4) ? 4 : 3;
    // fp32 contexts (VAMP_F32_LEAN_STAGE): the amplitude prior's logarithm in single precision, and no pole factor /
    // h y where no near-axis table will be built from them (Humlicek's W4 reads neither) -- the staging is ~1/4 of a
    // short region's half-step in fp32 (profiles/r04_c_small_ensembles.txt, stamps 2 -> 3)
    const bool lean = VAMP_F32_LEAN_STAGE && want_f32;
    const bool need_core = !(VAMP_F32_LEAN_STAGE && want_f32 && !TAB32 && !DT32);
    if (lane < K && writer) {
        const double* t = &L.theta[Q * lane];
        double a, c, Lw = 0.0, G = 0.0, sg = 0.0;
        if constexpr (MODE == VAMP_GAUSS3) {
            a = t[0]; c = t[1]; sg = t[2];
            lp = (lean ? xexp_logp32(a) : xexp_logp(a)) + uniform_logp(c, R.c_lo, R.c_hi, R.lp_c) + uniform_logp(sg, 0.0, R.w_max, R.lp_w);
        } else if constexpr (MODE == VAMP_VOIGT4) {
            a = t[0]; c = t[1]; Lw = t[2]; G = t[3];
            lp = (lean ? xexp_logp32(a) : xexp_logp(a)) + uniform_logp(c, R.c_lo, R.c_hi, R.lp_c) +
                 uniform_logp(Lw, 0.0, R.w_max, R.lp_w) + uniform_logp(G, 0.0, R.w_max, R.lp_w);
        } else {   // NBZ3: inverse of physics.py:15,27,120,134
            const double sig = t[1] * 1.0e3 * 1.41421356237309514547 / (2.355 * (R.line * 1.0e-10));
            a = t[0] * SIGMA0 / (sig * 2.50662827463100024161);
            c = (C_LIGHT / (R.line * (1.0 + t[2]) * 1.0e-10) - R.x_origin) / R.x_scale;
            G = (sig / R.x_scale) * FWHM_PER_SIGMA;
            Lw = R.l_fixed;
            lp = (lean ? xexp_logp32(a) : xexp_logp(a)) + uniform_logp(c, R.c_lo, R.c_hi, R.lp_c) + uniform_logp(G, 0.0, R.w_max, R.lp_w);
        }
        LineRec rec;
        rec.c = c;
        if constexpr (MODE == VAMP_GAUSS3) {
            rec.s = 1.0 / sg; rec.y = 0.0; rec.amp = a; rec.pole = 0.0; rec.hy = 0.0; rec.xcap = __builtin_huge_val();
            rec.w8 = rec.w25 = rec.wmid = 0.0;
        } else {
            const double rG = vamp::rcp_nr(G);          // one reciprocal for both scales (G = 0 -> inf -> rejected below)
            rec.s = (2.0 * SQRT_LN2) * rG;
            rec.y = (Lw * SQRT_LN2) * rG;
            rec.amp = a * rec.y;
            rec.pole = need_core ? vamp::core_pole_factor(rec.y) : 0.0;
            rec.hy = need_core ? vamp::core_hy(rec.y) : 0.0;
            // packed waves: another walker of the wave may force a deep fraction on this one, so every
            // line is capped there
            rec.xcap = (PK::SUBS == 1 && rec.s * R.tile_span <= 16.0) ? __builtin_huge_val() : vamp::X_FAR;   // NaN -> capped
            if constexpr (PK::FF) {
                rec.w8 = sqrt(fmax(vamp::R2_CORE - rec.y * rec.y, 0.0)) / rec.s;
                rec.w25 = sqrt(fmax(vamp::R2_M3 - rec.y * rec.y, 0.0)) / rec.s;
                rec.wmid = sqrt(fmax(VAMP_MID_Z2 - rec.y * rec.y, 0.0)) / rec.s;
                // half of the region's widest tile is at most VAMP_WIDE_MAX in this line's z: smooth across every tile
                my_wide = VAMP_WIDE_NODES && TAB && rec.s * (0.5 * R.tile_span) <= VAMP_WIDE_MAX;
            } else {
                rec.w8 = rec.w25 = rec.wmid = 0.0;
            }
            // a degenerate width (G = 0 or non-finite scale) makes the reference's profile NaN, which
            // its sampler rejects; reject here, before the sweep
            if (!(rec.s < __builtin_huge_val()) || !(rec.y < __builtin_huge_val())) lp = NEG_INF;
        }
        if (writer) L.line[lane] = rec;
        if (want_f32 && writer) {
            L.linef[lane][0] = (float)rec.c; L.linef[lane][1] = (float)rec.s;
            L.linef[lane][2] = (float)rec.y;
            L.linef[lane][3] = (float)(MODE == VAMP_GAUSS3 ? rec.amp : rec.amp * SQRT_PI);   // W4 returns H itself
        }
    }
    if constexpr (PK::FF && TAB && VAMP_WIDE_NODES) {
        const unsigned long long wm = __ballot(my_wide);
        if (writer && lane == 0) L.theta[WIDE_SLOT] = __longlong_as_double((long long)wm);
    }
    if (R.sample_sd && lane == PK::KCAP && writer) {      // one otherwise idle lane: sd ~ U(0,1), vpfits.py:39
        lp = uniform_logp(L.theta[R.D - 1], 0.0, 1.0, 0.0);
    }
    lp = wave_sum<PK::LPW>(lp);
    if constexpr (PK::SPLIT) {
        if (writer && lane == 0) L.theta[LP_SLOT] = lp;
    }
    group_barrier<PK>();
    if constexpr (PK::SPLIT) lp = L.theta[LP_SLOT];
    double (*dt)[vamp::DTAB_N];
    if constexpr (PK::DTAB_IN_TABLES && TAB)
        dt = reinterpret_cast<double (*)[vamp::DTAB_N]>(tab + PK::KCAP * vamp::TAB_LINE - PK::KCAP * vamp::DTAB_N);
    else if constexpr (TAB32)
        dt = reinterpret_cast<double (*)[vamp::DTAB_N]>(tab + TAB32_DOUBLES);     // behind the float rows
    else
        dt = L.dtab;
#ifdef VAMP_SKIP_DTAB     // timing-only builds (tools/variants.py)
    if (false) {
#else
    if (MODE != VAMP_GAUSS3 && (!want_f32 || TAB32 || DT32)) {      // (DT32: an fp32 blend builds its rows from L.dtab)
#endif
        constexpr int STEP = PK::SPLIT ? PK::THREADS : PK::LPW;
        for (int e = PK::SPLIT ? 64 * part + lane : lane; e < K * vamp::DTAB_N; e += STEP) {
            const int k = e / vamp::DTAB_N, n = e % vamp::DTAB_N;
            dt[k][n] = vamp::core_dtab_entry(n, L.line[k].y);
        }
    }
    group_barrier<PK>();
#ifdef VAMP_SKIP_TAB      // timing-only builds (tools/variants.py)
    if constexpr (false) {
#else
    if constexpr (TAB) {
#endif
        // one (line, interval) pair per thread: 16 lines x 16 intervals = the 256 threads of a split
        // group; a single wavefront takes its walker's pairs 64 at a time
        constexpr int TSTEP = PK::SPLIT ? PK::THREADS : PK::LPW;
        if constexpr (PK::DTAB_IN_TABLES) {
            // the near-axis tables occupy the tail of the Taylor tables: every thread reads what its row's centre
            // needs, and only when all have done so are the rows written (one (line, interval) pair per thread)
            static_assert(KMAX * vamp::TAB_NI <= PK::THREADS && PK::KCAP * vamp::DTAB_N <= PK::KCAP * vamp::TAB_LINE, "one row per thread");
            const int e = 64 * part + lane, k = e / vamp::TAB_NI, i = e % vamp::TAB_NI;
            double c0r = 0.0, c0i = 0.0;
            if (e < K * vamp::TAB_NI) vamp::core_centre(i, L.line[k].y, dt[k], L.line[k].pole, L.line[k].hy, c0r, c0i);
            group_barrier<PK>();
            if (e < K * vamp::TAB_NI) vamp::taylor_table_row_from_centre(i, L.line[k].y, c0r, c0i, tab + k * vamp::TAB_LINE + i * vamp::TAB_NT);
        } else {
            for (int e = PK::SPLIT ? 64 * part + lane : lane; e < K * vamp::TAB_NI; e += TSTEP) {
                const int k = e / vamp::TAB_NI, i = e % vamp::TAB_NI;
                vamp::taylor_table_row(i, L.line[k].y, dt[k], L.line[k].pole, L.line[k].hy,
                                       tab + k * vamp::TAB_LINE + i * vamp::TAB_NT);
            }
        }
        group_barrier<PK>();
    }
    if constexpr (TAB32) {
        // fp32 rows: one (line, interval) pair per thread, centre values in double from the near-axis tables
        static_assert(PK::SPLIT && KMAX * vamp::TAB_NI <= PK::THREADS, "one row per thread");
        const int e = 64 * part + lane, k = e / vamp::TAB_NI, i = e % vamp::TAB_NI;
        if (e < K * vamp::TAB_NI)
            vamp::taylor_table_row32(i, L.line[k].y, dt[k], L.line[k].pole, L.line[k].hy,
                                     reinterpret_cast<float*>(tab) + k * vamp::TAB32_LINE + i * vamp::TAB32_NT);
        group_barrier<PK>();
    }
    return lp;
}

// sqrt(pi) H for the TPIX pixels a lane holds, with ONE branch for the whole wavefront.
// Consecutive lanes hold consecutive pixels, so |z| is monotone along the wave on either side of
// the line centre; the wave takes the deepest branch any of its pixels needs (a deeper J-fraction
// is valid wherever a shallower one is).  Only waves that straddle |z|^2 = 64 diverge.
#ifndef VAMP_TPIX
#define VAMP_TPIX 4
#endif
constexpr int TPIX = VAMP_TPIX;      // pixels per lane per iteration in full tiles (the tail of a
                                     // region runs one pixel per lane, so short regions waste nothing)
#ifndef VAMP_SMALL_TPIX
#define VAMP_SMALL_TPIX 4
#endif
// several walkers per wavefront: pixels per lane and iteration (each near-axis evaluation in flight holds ~40 VGPRs)
template <class PK>
constexpr int pixels_per_lane() { return PK::SUBS > 1 ? VAMP_SMALL_TPIX : TPIX; }

template <int M, int T>
__device__ __forceinline__ void tile_jfrac(const double (&X)[T], const double (&r2)[T], double y, double (&H)[T]) {
    int t = 0;
#pragma unroll
    for (; t + 1 < T; t += 2) vamp::voigt_jfrac_x2<M>(X[t], X[t + 1], y, r2[t], r2[t + 1], H[t], H[t + 1]);
    if (t < T) H[t] = vamp::voigt_jfrac<M>(X[t], y, r2[t]);
}

// sqrt(pi) H from the line's Taylor table, 0 <= x < 8: TAB_NT / 2 16-byte LDS reads + TAB_NT - 1 fused multiply-adds
__device__ __forceinline__ double table_eval(const double* tab, double x) {
    // interval i = floor(2 x) as round-to-nearest of 2 (x - 1/4): adding 2^52 + 2^51 leaves the integer in the
    // low word of the sum (no conversions), and on a boundary either neighbour is right (|d| = 1/4 in both)
    constexpr double MAGIC = 6755399441055744.0;
    const double xs = x - 0.5 * vamp::CORE_H;
    const double t = fma(xs, 2.0, MAGIC);
    const int i = __double2loint(t);
    const double d = fma(t - MAGIC, -vamp::CORE_H, xs);
    static_assert(vamp::CORE_H == 0.5, "interval width 1/2");
    static_assert(vamp::TAB_NT % 2 == 0, "coefficient pairs below");
    constexpr int NP = vamp::TAB_NT / 2;
    // (v_mad_u32_u24: the 32-bit integer multiply runs at a quarter of the rate)
    const double2* a = reinterpret_cast<const double2*>(reinterpret_cast<const char*>(tab) + __mul24(i, vamp::TAB_NT * (int)sizeof(double)));
    double2 c[NP];
#pragma unroll
    for (int n = NP - 1; n >= 0; --n) c[n] = a[n];
    double r = fma(c[NP - 1].y, d, c[NP - 1].x);
#pragma unroll
    for (int n = NP - 2; n >= 0; --n) {
        r = fma(r, d, c[n].y);
        r = fma(r, d, c[n].x);
    }
    return r;
}

// `tab` (TAB = true): the line's Taylor table replaces the near-axis rule for |z|^2 < 64
template <int T, bool TAB = false>
__device__ __forceinline__ void tile_voigt(const LineRec& ln, const double* dtab, const double (&Xin)[T], double (&H)[T],
                                           const double* tab = nullptr, const double* ec = nullptr) {
    // ec: the exp constants in LDS (tile kernels with the far-field table) or null (literals)
    auto gauss_tail = [ec](double xx) { return ec ? vamp::exp_neg_sq_tab(xx, ec) : vamp::exp_neg_sq(xx); };
    const double y = ln.y;
    const double y2 = y * y;
    // narrow line: |z| spans many units (possibly decades) inside one tile, and a lane promoted to a
    // deep fraction would overflow |Q|^2 ~ |z|^(4m).  Such lines carry xcap = X_FAR (else +inf):
    // lanes beyond it are clamped into the range of the fractions here and get the closed far form
    // below.  One v_min per evaluation; the patch is a wave-uniform branch.
    const double xcap = ln.xcap;
    double X[T];
    if (xcap < __builtin_huge_val()) {           // (wave-uniform: four v_min less per line and tile for ordinary lines)
#pragma unroll
        for (int t = 0; t < T; ++t) X[t] = fmin(Xin[t], xcap);
    } else {
#pragma unroll
        for (int t = 0; t < T; ++t) X[t] = Xin[t];
    }
    double r2[T];
    double lo;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        r2[t] = fma(X[t], X[t], y2);
        lo = t ? fmin(lo, r2[t]) : r2[0];
    }
#ifdef VAMP_FORCE_TIER   // timing-only builds (tools/tier_cost.py): every tile takes one branch
    {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const double xx = X[t] < 6.4 ? X[t] : 6.4;
            if (VAMP_FORCE_TIER == 0) H[t] = vamp::voigt_core(xx, y, dtab, ln.pole, ln.hy);
            else if (VAMP_FORCE_TIER == 1) H[t] = vamp::voigt_jfrac<6>(X[t], y, r2[t]);
            else if (VAMP_FORCE_TIER == 2) H[t] = vamp::voigt_jfrac<4>(X[t], y, r2[t]);
            else if (VAMP_FORCE_TIER == 3) H[t] = vamp::voigt_jfrac<3>(X[t], y, r2[t]);
            else if (VAMP_FORCE_TIER == 4) H[t] = vamp::voigt_jfrac<2>(X[t], y, r2[t]);
            else if (VAMP_FORCE_TIER == 5) H[t] = vamp::voigt_far(X[t], y, r2[t]);
            else H[t] = X[t] * y;
        }
        return;
    }
#endif
    if (__any(lo < vamp::R2_M3)) {
        if (__any(lo < vamp::R2_M4)) {
            if (__any(lo < vamp::R2_CORE)) {
#if VAMP_CORE_FASTPATH
                // every pixel of the wavefront inside the zone (the usual case where a line is near): straight-line code,
                // T independent chains with all their LDS reads in flight together.  Behind per-pixel branches (below:
                // a wavefront that straddles |z| = 8) the T evaluations run one after the other, each waiting for its own
                // reads -- what a latency-bound small ensemble spends most of its sweep on
                double hi = r2[0];
#pragma unroll
                for (int t = 1; t < T; ++t) hi = fmax(hi, r2[t]);
                if (!__any(!(hi < vamp::R2_CORE))) {
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        if constexpr (TAB) H[t] = table_eval(tab, X[t]);
                        else H[t] = vamp::voigt_core(X[t], y, dtab, ln.pole, ln.hy);
                    }
                    return;          // (inside |z| < 8 no pixel is beyond xcap = X_FAR)
                }
#endif
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    if (r2[t] < vamp::R2_CORE) {
                        if constexpr (TAB) H[t] = table_eval(tab, X[t]);
                        else H[t] = vamp::voigt_core(X[t], y, dtab, ln.pole, ln.hy);
                    } else {
                        H[t] = vamp::voigt_jfrac<6>(X[t], y, r2[t]);
                    }
                }
                if (y < vamp::Y_TINY) {
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        if (!(r2[t] < vamp::R2_CORE)) H[t] += vamp::SQRT_PI * gauss_tail(X[t]);
                }
            } else {
                tile_jfrac<6, T>(X, r2, y, H);
                if (y < vamp::Y_TINY) {
#pragma unroll
                    for (int t = 0; t < T; ++t) H[t] += vamp::SQRT_PI * gauss_tail(X[t]);
                }
            }
        } else {
            tile_jfrac<4, T>(X, r2, y, H);
        }
    } else if (__any(lo < vamp::R2_M2)) {
        tile_jfrac<3, T>(X, r2, y, H);
    } else if (__any(lo < vamp::R2_M1)) {
        tile_jfrac<2, T>(X, r2, y, H);
    } else {
#pragma unroll
        for (int t = 0; t < T; ++t) H[t] = vamp::voigt_far(Xin[t], y, fma(Xin[t], Xin[t], y2));
    }
    // (for |z|^2 >= 196 the missing e^{-x^2} is < 1e-85: no tiny-y correction needed there)
    if (xcap < __builtin_huge_val()) {
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (Xin[t] > vamp::X_FAR) H[t] = vamp::voigt_far(Xin[t], y, fma(Xin[t], Xin[t], y2));
    }
}

// chi^2 sweep, fp64 pixel arithmetic.  Returns sum over the walker's pixels of ((f-m) w)^2.
// Full tiles: a lane holds TPIX pixels (i, i+LPW, ...) per iteration -- TPIX independent dependency
// chains and one LDS read of the line record per TPIX evaluations; the remaining pixels of the
// region run one per lane.
template <int MODE, class PK, int T, bool TAB = false>
__device__ __forceinline__ void sweep_range(const RegionDev& R, const typename PK::Lds& L, const double* __restrict__ x,
                                            const double* __restrict__ f, const double* __restrict__ wt, int lane,
                                            int base0, int base1, int stride, double& chi, const double* tab = nullptr) {
    const int K = R.K, P = R.P;
    constexpr bool gauss = (MODE == VAMP_GAUSS3);
    constexpr int LPW = PK::LPW;
    for (int base = base0; base < base1; base += stride) {
        double xi[T], tau[T];
        int idx[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int i = base + LPW * t + lane;
            idx[t] = i < P ? i : P - 1;          // tail lanes recompute the last pixel and drop it
            xi[t] = x[idx[t]];
            tau[t] = 0.0;
        }
#if VAMP_SR_EARLY
        // flux and weights requested now, used after the evaluations: their L2 round trip overlaps the
        // arithmetic instead of following it (the compiler sinks loads to their use)
        double fi[T], wi[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            fi[t] = f[idx[t]];
            wi[t] = wt[idx[t]];
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
        VAMP_STAMP_AFTER(31, xi[0]);
        if constexpr (gauss) {
            for (int k = 0; k < K; ++k) {
                const double c = L.line[k].c, s = L.line[k].s, a = L.line[k].amp;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const double u = (xi[t] - c) * s;
                    tau[t] += a * exp(-0.5 * (u * u));
                }
            }
        } else {
            for (int k = 0; k < K; ++k) {
                const LineRec ln = L.line[k];
                double X[T], H[T];
#pragma unroll
                for (int t = 0; t < T; ++t) X[t] = fabs(xi[t] - ln.c) * ln.s;
                tile_voigt<T, TAB>(ln, dtab_row<PK>(L, k), X, H, TAB ? tab + k * vamp::TAB_LINE : nullptr);
#pragma unroll
                for (int t = 0; t < T; ++t) tau[t] = fma(ln.amp, H[t], tau[t]);
            }
        }
        VAMP_STAMP_AFTER(32, tau[0]);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const double m = vamp::exp_taylor(-tau[t]);
#if VAMP_SR_EARLY
            const double r = (fi[t] - m) * wi[t];
#else
            const double r = (f[idx[t]] - m) * wt[idx[t]];
#endif
            const bool live = (base + LPW * t + lane) < P;
            chi += live ? r * r : 0.0;
        }
        VAMP_STAMP_AFTER(33, chi);
    }
}

// ---- far-field aggregation ----------------------------------------------------------------
// Seen from a tile of 256 consecutive pixels, a line whose centre lies >= FF_DIST half-widths
// beyond the tile's edge contributes an optical depth that is analytic across the tile with its
// nearest singularity (the line centre) that far away: a degree-15 Chebyshev interpolant
// reproduces it to 6e-13 relative (singularity at (FF_DIST + 1) half-widths from the tile centre:
// Bernstein ellipse parameter 3 + sqrt(8) = 5.8 for FF_DIST = 2, and 5.8^-16 = 6e-13; FF_DIST = 4: 9.9^-16 = 1.2e-16).  Polynomials add, so ALL far lines of the tile share one interpolant:
//   1. the (node, far line) pairs -- 16 nodes x up to 16 lines -- are spread over the 64 lanes
//      (lane = 16 * (line slot) + node) and evaluated with the same Voigt code, 4 lines per pass;
//   2. node values are summed over lines (two xor shuffles) and turned by ONE 56 x 16 matrix held in
//      LDS (one row per lane, tools/gen_ff_matrix.py) into four local power series, one per quarter
//      of the tile -- a lane holds one pixel of each quarter -- with 14 coefficients each: seen from
//      a quarter's centre the far lines are >= 9 quarter half-widths away, the coefficients fall
//      like 9^-j and the terms beyond u^13 are < 3e-11 of the value (of a wing that is itself a small part
//      of the optical depth there);
//   3. every pixel of the tile runs Horner's rule on its quarter's series: 13 fused multiply-adds
//      (Clenshaw's recurrence on the tile's Chebyshev coefficients cost 30) instead of ~27
//      instructions per far line.
// On the headline workload ~90 % of the (pixel, line) evaluations are far: 64 wave-evaluations
// per tile shrink to ~6 direct ones + ~4 at the nodes.
constexpr int FF_EXP = FF_MAT + FF_NODES;     // constants of the exp kernel (vamp::exp_taylor_tab), 16-byte aligned
constexpr int FF_TABLE = FF_EXP + vamp::EXP_TAB_N;   // the matrix [n/2][lane][n%2], the node abscissae, the exp constants
__device__ const double EXP_TAB[vamp::EXP_TAB_N] = VAMP_EXP_TAB_INIT;

// fp32 contexts: the 32 x 8 matrix of the 8-node interpolant and its node abscissae, as floats (1 KB instead of 7.4)
constexpr int FF32_TABLE = FF32_MAT + FF32_NODES;
template <bool F32>
constexpr int ff_table_doubles() { return F32 ? (FF32_TABLE + 1) / 2 : FF_TABLE; }
// every thread of the workgroup copies its share; call before any thread can leave the kernel
template <bool F32>
__device__ __forceinline__ void ff_fill_table(double* dct) {
    if constexpr (F32) {
        float* d32 = reinterpret_cast<float*>(dct);
        for (int e = threadIdx.x; e < FF32_TABLE; e += blockDim.x) d32[e] = FF32_M[e];
    } else {
        for (int e = threadIdx.x; e < FF_TABLE; e += blockDim.x) dct[e] = e < FF_EXP ? FF_M[e] : EXP_TAB[e - FF_EXP];
    }
    __syncthreads();
}

// sqrt(pi) H at one point per slot for different lines per lane (x[t], y[t]), two slots at a time;
// every point has |z|^2 >= 64.  One branch for the wavefront per slot PAIR, the deepest any of its
// 128 points needs, and one reciprocal per pair.  (The far list is ordered deep lines first, so the
// second pair usually runs the cheapest fraction.)
template <int M>
__device__ __forceinline__ void ff_frac2(const double (&X)[2], const double (&y)[2], const double (&r2)[2], double (&H)[2]) {
    double n[2], d[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) vamp::voigt_jfrac_nd<M>(X[t], y[t], r2[t], n[t], d[t]);
    const double ra = vamp::rcp_nr(d[0] * d[1]);
    H[0] = n[0] * (ra * d[1]);
    H[1] = n[1] * (ra * d[0]);
}

__device__ __forceinline__ void ff_eval2(const double (&Xin)[2], const double (&y)[2], double (&H)[2], const double* ec) {
    double X[2], r2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        X[t] = fmin(Xin[t], vamp::X_FAR);       // lanes hold different lines: always guard the promotion
        r2[t] = fma(X[t], X[t], y[t] * y[t]);
    }
    const double lo = fmin(r2[0], r2[1]), hi = fmax(Xin[0], Xin[1]), ymin = fmin(y[0], y[1]);
    // the fractions' ranges for NODE values: 4e-13 / 2e-13 / 9e-13 of the value at the lower ends (the m-level fraction is
    // the 2m-point Gauss-Hermite rule of w's integral; the pixel evaluator's own ranges, R2_M4 / M3 / M2 = 196 / 625 / 1e4,
    // hold 2e-15) -- the series these values feed reproduce a wing to 3e-11 of its value
    constexpr double N_M4 = VAMP_FF_R2_M4, N_M3 = VAMP_FF_R2_M3, N_M2 = VAMP_FF_R2_M2;
    if (__any(lo < N_M3)) {
        if (__any(lo < N_M4)) ff_frac2<6>(X, y, r2, H);
        else ff_frac2<4>(X, y, r2, H);
    } else if (__any(lo < N_M2)) {
        ff_frac2<3>(X, y, r2, H);
    } else {
        ff_frac2<2>(X, y, r2, H);               // valid (more than accurate) beyond 1e8 too, up to X_FAR
    }
    if (__any(hi > vamp::X_FAR)) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (Xin[t] > vamp::X_FAR) H[t] = vamp::voigt_far(Xin[t], y[t], fma(Xin[t], Xin[t], y[t] * y[t]));
    }
    if (__any(ymin < vamp::Y_TINY)) {           // the fractions miss e^{-x^2}; it matters for y < ~1e-11 near |z| = 8
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (y[t] < vamp::Y_TINY) H[t] += vamp::SQRT_PI * vamp::exp_neg_sq_tab(X[t], ec);
    }
}

// far-field steps shared by the fp64 and the fp32 sweep.  All wave-uniform in control flow.
// (a) classification of all lines at once: far = centre >= FF_DIST half-widths beyond the tile's
//     edge and the whole tile outside |z|^2 < 64 of that line; compacted list -> Sx.farlist
//     (my_w25: half-width of |z|^2 < 625; far lines that reach into it at the tile's edge need the
//     deep fractions and are listed first)
template <int FARLIST_SCALE>       // bytes per line record of the list's readers
__device__ __forceinline__ unsigned long long ff_classify(TileScratch& Sx, int K, int lane, double my_c, double my_w8,
                                                          double my_w25, double mid, double half) {
    const double dist = fabs(mid - my_c) - half;
    const bool my_far = lane < K && dist >= FF_DIST * half && dist >= my_w8;
    const bool my_deep = my_far && dist < my_w25;
    const unsigned long long farmask = __ballot(my_far), deepmask = __ballot(my_deep);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (my_far)
        Sx.farlist[my_deep ? __builtin_popcountll(deepmask & below)
                           : __builtin_popcountll(deepmask) + __builtin_popcountll(farmask & ~deepmask & below)] = lane * FARLIST_SCALE;
    return farmask;
}
// node sums (lanes 0..15 hold them) -> the four local power series of the tile: lane l = 14 q + j < 56 owns
// coefficient j of quarter q, a = sum_n M[l][n] f_n, stored in the pixel arithmetic type
template <class real>
__device__ __forceinline__ void ff_series(TileScratch& Sx, const double* __restrict__ dct, int lane, double fs) {
    if (lane < FF_NODES) Sx.coef[lane] = fs;
    __builtin_amdgcn_wave_barrier();
    const double2* mrow = reinterpret_cast<const double2*>(dct) + (lane < FF_ROWS ? lane : FF_ROWS - 1);
    const double2* fv = reinterpret_cast<const double2*>(Sx.coef);
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int n2 = 0; n2 < FF_NODES / 2; ++n2) {
        const double2 m = mrow[n2 * FF_ROWS], f2 = fv[n2];
        a0 = fma(m.x, f2.x, a0);
        a1 = fma(m.y, f2.y, a1);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < FF_ROWS) reinterpret_cast<real*>(Sx.coef)[lane] = (real)(a0 + a1);
    __builtin_amdgcn_wave_barrier();
}
// (b) optical depth of the far lines at the tile's Chebyshev nodes -> the tile's local power series in Sx.coef
//     (fp32 contexts: ff32_coefficients below, 8 nodes)
template <class LDS>
__device__ __forceinline__ void ff_coefficients(const LDS& L, TileScratch& Sx, const double* __restrict__ dct,
                                                int lane, int nfar, double mid, double half, double fs_wide = 0.0) {
    const int node = lane & (FF_NODES - 1), grp = lane >> 4;
    const double tnode = dct[FF_MAT + node];                       // cos(pi (node + 1/2) / 16)
    __builtin_amdgcn_wave_barrier();
    // 1. lane = (slot group, node), four lines per lane (line q = 4 t + group of the compacted far list)
    const double xnode = fma(half, tnode, mid);
    double Xn[4], yn[4], an[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int q = 4 * t + grp;
        const LineRec& ln = *reinterpret_cast<const LineRec*>(reinterpret_cast<const char*>(L.line) + (nfar > 0 ? Sx.farlist[q < nfar ? q : nfar - 1] : 0));
        Xn[t] = fabs(xnode - ln.c) * ln.s;
        yn[t] = ln.y;
        an[t] = q < nfar ? ln.amp : 0.0;
    }
    double fs = 0.0;
    if (nfar > 0) {
        const double Xa[2] = {Xn[0], Xn[1]}, ya[2] = {yn[0], yn[1]};
        double Ha[2];
        ff_eval2(Xa, ya, Ha, dct + FF_EXP);
        fs = fma(an[0], Ha[0], an[1] * Ha[1]);
    }
    if (nfar > 8) {                               // list entries 8..15 (slots 2, 3)
        const double Xb[2] = {Xn[2], Xn[3]}, yb[2] = {yn[2], yn[3]};
        double Hb[2];
        ff_eval2(Xb, yb, Hb, dct + FF_EXP);
        fs += fma(an[2], Hb[0], an[3] * Hb[1]);
    }
    fs += fs_wide;
    fs += __shfl_xor(fs, 16, 64);
    fs += __shfl_xor(fs, 32, 64);
    ff_series<double>(Sx, dct, lane, fs);
}
// ---- lines far wider than the tile -------------------------------------------------------------------------------
// An ensemble drawn from the priors (widths ~ U(0, fwhm_max): where every find_bic repeat STARTS, vpfits.py:283-297) has
// nothing far: every line is wider than the region, every (line, tile) pair is near and costs four table look-ups per
// lane (profiles/r03_c_headline_robustness.txt: 12.5 ms per swept half-step against 3.4).  But seen from a tile, such a
// line is as smooth as a far one: w is entire, and over a tile whose half-width is at most VAMP_WIDE_MAX = 3/4 in the
// line's own z (G_fwhm >~ 284 px on a unit grid) the degree-15 interpolant through the tile's 16 Chebyshev nodes
// reproduces its optical depth to ~5e-12 of its peak (the first dropped coefficient, 2 e^{-a^2/2} I_8(a^2/2)).  So the roles of "far" and "wide" are exchanged: wide lines are evaluated
// at the NODES -- (node, line) pairs over the lanes as for the far lines, 4 lines per lane, through the line's own
// Taylor table (or the 6-level fraction beyond |z| = 8) -- and join the far lines' node sums before the one transform.
// 16 wide lines: 4 look-ups per lane and tile instead of 64.
template <class LDS>
__device__ __forceinline__ double ff_wide_nodes(const LDS& L, const TileScratch& Sx, const double* __restrict__ dct, const double* tab,
                                                int lane, int nwide, double mid, double half) {
    const int node = lane & (FF_NODES - 1), grp = lane >> 4;
    const double xnode = fma(half, dct[FF_MAT + node], mid);
    double fs = 0.0;
    for (int t = 0; 4 * t < nwide; ++t) {
        const int q = 4 * t + grp;
        const int k = Sx.widelist[q < nwide ? q : nwide - 1];
        const LineRec& ln = L.line[k];
        const double X = fabs(xnode - ln.c) * ln.s, r2 = fma(X, X, ln.y * ln.y);
        double H;
        if (r2 < vamp::R2_CORE) {
            H = table_eval(tab + k * vamp::TAB_LINE, X);
        } else {
            H = vamp::voigt_jfrac<6>(X, ln.y, r2);
            if (ln.y < vamp::Y_TINY) H += vamp::SQRT_PI * vamp::exp_neg_sq_tab(X, dct + FF_EXP);
        }
        fs = fma(q < nwide ? ln.amp : 0.0, H, fs);
    }
    return fs;
}
// (c) Horner's rule at the tile's pixels, added to tau (in the pixel arithmetic type).  Register t of a
//     lane is pixel 64 t + lane of the tile: quarter t of an ascending grid, 3 - t of a descending one.
template <class real, int T>
__device__ __forceinline__ void ff_horner(const TileScratch& Sx, const real (&xi)[T], double mid, double half, bool up,
                                          real (&tau)[T]) {
    static_assert(T == 4, "one pixel per quarter of the tile");
    const real scale = (real)(4.0 * vamp::rcp_nr(half));     // (a last-bit error in the scale moves u by 1e-16)
    real u[T], acc[T];
    const real* cq[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int q = up ? t : T - 1 - t;
        cq[t] = reinterpret_cast<const real*>(Sx.coef) + q * (FF_DEG + 1);
        u[t] = (xi[t] - (real)fma(half, 0.5 * q - 0.75, mid)) * scale;
        acc[t] = cq[t][FF_DEG];
    }
#pragma unroll
    for (int j = FF_DEG - 1; j >= 0; --j) {
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = fma(acc[t], u[t], cq[t][j]);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) tau[t] += acc[t];
    __builtin_amdgcn_wave_barrier();
}

// WIDE: this walker has lines far wider than a tile (ff_wide_nodes).  A copy of the loop of its own, chosen per walker
// by a wave-uniform branch: compiled into the one loop, the wide path cost the converged headline ensemble -- which has no
// such line -- 2.8 % through the loop's register allocation (profiles/r04_e_wide_lines.txt).
template <int MODE, class PK, bool TAB, bool WIDE = false>
__device__ __forceinline__ void sweep_range_ff(const RegionDev& R, const typename PK::Lds& L, TileScratch& Sx, const double* __restrict__ dct,
                                               const double* __restrict__ x, const double* __restrict__ f,
                                               const double* __restrict__ wt, int lane, int base0, int base1, int stride,
                                               double& chi, const double* tab, unsigned long long wide_all = 0ull) {
    constexpr int T = TPIX;
    static_assert(PK::LPW == 64 && PK::KCAP <= 16, "far-field tiles: one walker per wavefront, <= 16 lines");
    const int K = R.K;
    // lane k < K classifies line k; w8 = half-width of |z|^2 < 64 around the line centre, in x units
    // (centre and zone half-widths of "its" line are read from the record in every tile: three LDS reads
    // instead of six registers held through the loop; the opaque index keeps the reads in the loop)
    const int kk = lane < K ? lane : 0;
#if VAMP_X_PREFETCH
    // the abscissae of a tile are requested one tile ahead: they head every dependency chain of the
    // tile (classification, near lines, far-field series), and an L2 round trip at the top of each of the
    // 16 iterations is paid by all the wavefronts of a workgroup together
    double xn[T], xn_lo = 0.0, xn_hi = 0.0;
    if (base0 < base1) {
#pragma unroll
        for (int t = 0; t < T; ++t) xn[t] = x[base0 + 64 * t + lane];
        xn_lo = x[base0];
        xn_hi = x[base0 + 64 * T - 1];
    }
#endif
    for (int base = base0; base < base1; base += stride) {
        double xi[T], tau[T];
#if VAMP_X_PREFETCH
        const double x_lo = xn_lo, x_hi = xn_hi;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            xi[t] = xn[t];
            tau[t] = 0.0;
        }
        {
            const int nb = base + stride < base1 ? base + stride : base;     // the last tile re-reads itself
#pragma unroll
            for (int t = 0; t < T; ++t) xn[t] = x[nb + 64 * t + lane];
            xn_lo = x[nb];
            xn_hi = x[nb + 64 * T - 1];
        }
#else
#pragma unroll
        for (int t = 0; t < T; ++t) {
            xi[t] = x[base + 64 * t + lane];
            tau[t] = 0.0;
        }
        // tile geometry (wave-uniform)
        const double x_lo = x[base], x_hi = x[base + 64 * T - 1];
#endif
        const double mid = 0.5 * (x_lo + x_hi), half = 0.5 * fabs(x_hi - x_lo);
        int kq = kk;
        asm volatile("" : "+v"(kq));
        const LineRec& me = L.line[kq];
        const unsigned long long farmask = ff_classify<(int)sizeof(LineRec)>(Sx, K, lane, me.c, me.w8, me.w25, mid, half);
        const int nfar = __builtin_popcountll(farmask);
        // near lines far wider than the tile join the interpolant (ff_wide_nodes); needs the lines' Taylor tables
        unsigned long long widemask = 0ull;
        int nwide = 0;
        if constexpr (WIDE) {
            widemask = wide_all & ~farmask;
            if constexpr (VAMP_MID_NODES) {
                // between "far" and "near": the centre is FF_DIST half-widths away, the tile still inside |z| < 8 of the line
                // (where the node values cannot be continued fractions) but outside its Gaussian core: as smooth as a far
                // line, evaluated at the nodes through its Taylor table like a wide one
                const double dist = fabs(mid - me.c) - half;
                widemask |= __ballot(lane < K && dist >= FF_DIST * half && dist >= me.wmid) & ~farmask;
            }
            nwide = __builtin_popcountll(widemask);
            if ((widemask >> lane) & 1ull) Sx.widelist[__builtin_popcountll(widemask & ((1ull << lane) - 1ull))] = lane;
        }
        // near lines: walk the set bits of the complement of the far mask (1-3 of 16 on the headline)
        // (VAMP_SKIP_*: timing-only builds of tools/variants.py -- the phase split in profiles/)
#ifndef VAMP_SKIP_NEAR
        for (unsigned long long near = ~(farmask | widemask) & ((1ull << K) - 1ull); near; near &= near - 1ull) {
            const int k = __builtin_ctzll(near);
            const LineRec ln = L.line[k];
            double X[T], H[T];
#pragma unroll
            for (int t = 0; t < T; ++t) X[t] = fabs(xi[t] - ln.c) * ln.s;
            tile_voigt<T, TAB>(ln, dtab_row<PK>(L, k), X, H, TAB ? tab + k * vamp::TAB_LINE : nullptr, dct + FF_EXP);
#pragma unroll
            for (int t = 0; t < T; ++t) tau[t] = fma(ln.amp, H[t], tau[t]);
        }
#endif
        // flux and weights of the tile are requested here, ahead of the far-field series and the
        // exponentials that separate them from their use: left to itself the compiler sinks each
        // load to its use and the wavefront sits through eight L2 round trips per tile
        double fi[T], wi[T];
#ifndef VAMP_SKIP_FFNODES
        if (nfar + nwide > 0) {
            double fs_wide = 0.0;
            if constexpr (WIDE) {
                if (nwide > 0) {
                    __builtin_amdgcn_wave_barrier();          // (the wide list is in LDS)
                    fs_wide = ff_wide_nodes<typename PK::Lds>(L, Sx, dct, tab, lane, nwide, mid, half);
                }
            }
            ff_coefficients<typename PK::Lds>(L, Sx, dct, lane, nfar, mid, half, fs_wide);
        }
#endif
#if VAMP_EARLY_LOADS
#pragma unroll
        for (int t = 0; t < T; ++t) {
            fi[t] = f[base + 64 * t + lane];
            wi[t] = wt[base + 64 * t + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
#ifndef VAMP_SKIP_CLENSHAW
        if (nfar + nwide > 0) ff_horner<double, T>(Sx, xi, mid, half, x_hi > x_lo, tau);
#endif
        // the exp constants are read from LDS here, per tile: as literals they sit in ~30 SGPRs (or VGPRs) through
        // the whole loop, and the pointers and masks they displace are then reloaded from VGPR lanes in every tile
        int eoff = FF_EXP;
        asm volatile("" : "+s"(eoff));          // (keeps the compiler from hoisting the reads out of the loop)
        const double* ec = dct + eoff;
#pragma unroll
        for (int t = 0; t < T; ++t) {
#if !VAMP_EARLY_LOADS
            fi[t] = f[base + 64 * t + lane];
            wi[t] = wt[base + 64 * t + lane];
#endif
#ifdef VAMP_SKIP_EXP
            const double m = 1.0 - tau[t];
#else
            const double m = vamp::exp_taylor_tab<VAMP_FLUX_EXP_DROP>(-tau[t], ec);
#endif
            const double r = (fi[t] - m) * wi[t];
            chi = fma(r, r, chi);
        }
    }
}

#ifndef VAMP_FARFIELD
#define VAMP_FARFIELD 1
#endif

// fp32 pixel arithmetic (Humlicek W4), chi^2 accumulated in fp64 (SURVEY section 7 hard parts).
// Same shape as the fp64 sweep: TPIX pixels per lane in full tiles and one wave-uniform region per
// (tile, line) -- region I when every lane has s >= 15, region II (valid for all s >= 5.5) when
// every lane has s >= 5.5, per-lane selection only for tiles that touch the line core.
template <int T>
__device__ __forceinline__ void tile_w4(float y, const float (&X)[T], float (&H)[T]) {
    float lo = X[0];
#pragma unroll
    for (int t = 1; t < T; ++t) lo = fminf(lo, X[t]);
    lo += y;
    if (!__any(!(lo >= 15.0f))) {
#pragma unroll
        for (int t = 0; t < T; ++t) H[t] = vamp::w4_region1(X[t], y);
    } else if (!__any(!(lo >= 5.5f))) {
#pragma unroll
        for (int t = 0; t < T; ++t) H[t] = vamp::w4_region2(X[t], y);
    } else {
#pragma unroll
        for (int t = 0; t < T; ++t) H[t] = vamp::humlicek_w4_re(X[t], y);
    }
}

// the same with the line's fp32 Taylor rows for |z|^2 < 64 (the zone the far-field classification calls a line's
// core): region I / II outside it (|z| >= 8 implies s >= 5.5), one table look-up per pixel inside
__device__ __forceinline__ float table32_eval(const float* tab, float x) {
    // interval i = floor(2 x) as round-to-nearest of 2 (x - 1/4): adding 2^23 + 2^22 leaves the integer in the low
    // bits of the sum (no conversions); on a boundary either neighbour is right (|d| = 1/4 in both)
    constexpr float MAGIC = 12582912.0f;
    const float xs = x - 0.25f;
    const float t = fmaf(xs, 2.0f, MAGIC);
    const int i = __float_as_int(t) & 15;
    const float d = fmaf(t - MAGIC, -0.5f, xs);
    const float4* a = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tab) + i * (vamp::TAB32_NT * (int)sizeof(float)));
    const float4 lo = a[0], hi = a[1];
    float r = fmaf(hi.w, d, hi.z);
    r = fmaf(r, d, hi.y);
    r = fmaf(r, d, hi.x);
    r = fmaf(r, d, lo.w);
    r = fmaf(r, d, lo.z);
    r = fmaf(r, d, lo.y);
    return fmaf(r, d, lo.x);
}
template <int T>
__device__ __forceinline__ void tile_w4_tab(float y, const float (&X)[T], float (&H)[T], const float* tab) {
    float lo = X[0];
#pragma unroll
    for (int t = 1; t < T; ++t) lo = fminf(lo, X[t]);
    const float y2 = y * y;
    if (!__any(!(lo + y >= 15.0f))) {
#pragma unroll
        for (int t = 0; t < T; ++t) H[t] = vamp::w4_region1(X[t], y);
    } else if (!__any(!(fmaf(lo, lo, y2) >= (float)vamp::R2_CORE))) {
#pragma unroll
        for (int t = 0; t < T; ++t) H[t] = vamp::w4_region2(X[t], y);
    } else {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (fmaf(X[t], X[t], y2) < (float)vamp::R2_CORE) H[t] = table32_eval(tab, X[t]);
            else H[t] = vamp::w4_region2(X[t], y);
        }
    }
}

template <int MODE, class PK, int T>
__device__ __forceinline__ void sweep_range_f32(const RegionDev& R, const typename PK::Lds& L, const float* __restrict__ x,
                                                const float* __restrict__ f, const float* __restrict__ wt, int lane,
                                                int base0, int base1, int stride, double& chi) {
    const int K = R.K, P = R.P;
    constexpr bool gauss = (MODE == VAMP_GAUSS3);
    constexpr int LPW = PK::LPW;
    for (int base = base0; base < base1; base += stride) {
        float xi[T], tau[T];
        int idx[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int i = base + LPW * t + lane;
            idx[t] = i < P ? i : P - 1;
            xi[t] = x[idx[t]];
            tau[t] = 0.0f;
        }
        for (int k = 0; k < K; ++k) {
            const float c = L.linef[k][0], s = L.linef[k][1], y = L.linef[k][2], a = L.linef[k][3];
            if constexpr (gauss) {
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float u = (xi[t] - c) * s;
                    tau[t] += a * __expf(-0.5f * (u * u));
                }
            } else {
                float X[T], H[T];
#pragma unroll
                for (int t = 0; t < T; ++t) X[t] = fabsf(xi[t] - c) * s;
                tile_w4<T>(y, X, H);
#pragma unroll
                for (int t = 0; t < T; ++t) tau[t] = fmaf(a, H[t], tau[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float m = __expf(-tau[t]);
            const float r = (f[idx[t]] - m) * wt[idx[t]];
            const bool live = (base + LPW * t + lane) < P;
            chi += live ? (double)r * (double)r : 0.0;
        }
    }
}

// ---- far field of fp32 contexts: 8 nodes ------------------------------------------------------------
// Single precision holds 6e-8; an 8-node (degree-7) Chebyshev interpolant already reproduces the optical depth of
// a line >= FF_DIST half-widths beyond the tile's edge to 3e-7 (tests/test_ff_matrix.py), so the fp32 path uses
// that instead of the 16-node one: lane = 8 (line slot) + node, two lines per lane, node sums over three xor
// shuffles, a 32 x 8 float matrix (row sums <= 3.4: no conditioning problem in fp32) -> four local series of 8
// coefficients, 7 multiply-adds per pixel.  Node values through W4 regions I / II (every far point has |z| >= 8).
template <class LDS>
__device__ __forceinline__ void ff32_coefficients(const LDS& L, TileScratch& Sx, const float* __restrict__ dct32, int lane, int nfar,
                                                  double mid, double half) {
    const int node = lane & (FF32_NODES - 1), grp = lane >> 3;
    const float tnode = dct32[FF32_MAT + node];                    // cos(pi (node + 1/2) / 8)
    __builtin_amdgcn_wave_barrier();
    const float xn = (float)fma(half, (double)tnode, mid);
    float X[2], yv[2], av[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int q = 8 * t + grp;
        const float* lf = reinterpret_cast<const float*>(reinterpret_cast<const char*>(L.linef) + Sx.farlist[q < nfar ? q : nfar - 1]);
        X[t] = fabsf(xn - lf[0]) * lf[1];
        yv[t] = lf[2];
        av[t] = q < nfar ? lf[3] : 0.0f;
    }
    float fs;
    if (nfar > 8) {
        const float lo = fminf(X[0] + yv[0], X[1] + yv[1]);
        if (!__any(!(lo >= 15.0f))) fs = fmaf(av[0], vamp::w4_region1(X[0], yv[0]), av[1] * vamp::w4_region1(X[1], yv[1]));
        else fs = fmaf(av[0], vamp::w4_region2(X[0], yv[0]), av[1] * vamp::w4_region2(X[1], yv[1]));
    } else {
        if (!__any(!(X[0] + yv[0] >= 15.0f))) fs = av[0] * vamp::w4_region1(X[0], yv[0]);
        else fs = av[0] * vamp::w4_region2(X[0], yv[0]);
    }
    fs += __shfl_xor(fs, 8, 64);
    fs += __shfl_xor(fs, 16, 64);
    fs += __shfl_xor(fs, 32, 64);
    // node sums (lanes 0..7 hold them) -> the four local series: lane l = 8 q + j < 32 owns coefficient j of quarter q
    float* cf = reinterpret_cast<float*>(Sx.coef);
    if (lane < FF32_NODES) cf[lane] = fs;
    __builtin_amdgcn_wave_barrier();
    const float4* mrow = reinterpret_cast<const float4*>(dct32) + (lane < FF32_ROWS ? lane : FF32_ROWS - 1);
    const float4* fv = reinterpret_cast<const float4*>(Sx.coef);
    const float4 m0 = mrow[0], m1 = mrow[FF32_ROWS], f0 = fv[0], f1 = fv[1];
    float a0 = m0.x * f0.x, a1 = m0.y * f0.y;
    a0 = fmaf(m0.z, f0.z, a0); a1 = fmaf(m0.w, f0.w, a1);
    a0 = fmaf(m1.x, f1.x, a0); a1 = fmaf(m1.y, f1.y, a1);
    a0 = fmaf(m1.z, f1.z, a0); a1 = fmaf(m1.w, f1.w, a1);
    __builtin_amdgcn_wave_barrier();
    if (lane < FF32_ROWS) cf[lane] = a0 + a1;
    __builtin_amdgcn_wave_barrier();
}
// Horner's rule at the tile's pixels (register t of a lane = pixel 64 t + lane: quarter t of an ascending grid)
template <int T>
__device__ __forceinline__ void ff32_horner(const TileScratch& Sx, const float (&xi)[T], double mid, double half, bool up,
                                            float (&tau)[T]) {
    static_assert(T == 4, "one pixel per quarter of the tile");
    const float hf = (float)half, mf = (float)mid;
    const float scale = 4.0f * __builtin_amdgcn_rcpf(hf);
    float u[T], acc[T];
    const float* cq[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int q = up ? t : T - 1 - t;
        cq[t] = reinterpret_cast<const float*>(Sx.coef) + q * (FF32_DEG + 1);
        u[t] = (xi[t] - fmaf(hf, 0.5f * q - 0.75f, mf)) * scale;
        acc[t] = cq[t][FF32_DEG];
    }
#pragma unroll
    for (int j = FF32_DEG - 1; j >= 0; --j) {
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = fmaf(acc[t], u[t], cq[t][j]);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) tau[t] += acc[t];
    __builtin_amdgcn_wave_barrier();
}

// fp32 sweep of full tiles with the far field: near lines through W4 in fp32, all far lines through
// the tile's interpolant -- node values and cosine transform in fp64 (one evaluation per lane,
// same code as the fp64 path), transform in fp64, Horner per pixel in fp32.
template <int MODE, class PK, bool TAB32>
__device__ __forceinline__ void sweep_range_f32_ff(const RegionDev& R, const typename PK::Lds& L, TileScratch& Sx,
                                                   const double* __restrict__ dct, const float* __restrict__ x,
                                                   const float* __restrict__ f, const float* __restrict__ wt, int lane,
                                                   int base0, int base1, int stride, double& chi, const float* tab32) {
    constexpr int T = TPIX;
    static_assert(PK::LPW == 64 && PK::KCAP <= 16 && MODE != VAMP_GAUSS3, "far-field tiles: one walker per wavefront, Voigt lines");
    const int K = R.K;
    const int kk = lane < K ? lane : 0;
    for (int base = base0; base < base1; base += stride) {
        float xi[T], tau[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            xi[t] = x[base + 64 * t + lane];
            tau[t] = 0.0f;
        }
        const double x_lo = (double)x[base], x_hi = (double)x[base + 64 * T - 1];
        const double mid = 0.5 * (x_lo + x_hi), half = 0.5 * fabs(x_hi - x_lo);
        int kq = kk;
        asm volatile("" : "+v"(kq));
        const LineRec& me = L.line[kq];
        // (fp32: the node values are W4 regions I / II, valid from |x| + y = 5.5 -- no |z| >= 8 condition as for the fp64
        //  fractions: a line is far once the tile is outside its Gaussian core, |z|^2 >= VAMP_MID_Z2)
        const unsigned long long farmask = ff_classify<4 * (int)sizeof(float)>(Sx, K, lane, me.c, VAMP_F32_FAR_CORE ? me.wmid : me.w8, me.w25, mid, half);
        const int nfar = __builtin_popcountll(farmask);
        // (VAMP_SKIP_*: timing-only builds of tools/variants.py -- the phase split in profiles/)
#ifndef VAMP_SKIP_NEAR
        for (unsigned long long near = ~farmask & ((1ull << K) - 1ull); near; near &= near - 1ull) {
            const int k = __builtin_ctzll(near);
            const float c = L.linef[k][0], sc = L.linef[k][1], y = L.linef[k][2], a = L.linef[k][3];
            float X[T], H[T];
#pragma unroll
            for (int t = 0; t < T; ++t) X[t] = fabsf(xi[t] - c) * sc;
            if constexpr (TAB32) tile_w4_tab<T>(y, X, H, tab32 + k * vamp::TAB32_LINE);
            else tile_w4<T>(y, X, H);
#pragma unroll
            for (int t = 0; t < T; ++t) tau[t] = fmaf(a, H[t], tau[t]);
        }
#endif
        if (nfar > 0) {
#ifndef VAMP_SKIP_FFNODES
            ff32_coefficients<typename PK::Lds>(L, Sx, reinterpret_cast<const float*>(dct), lane, nfar, mid, half);
#endif
#ifndef VAMP_SKIP_CLENSHAW
            ff32_horner<T>(Sx, xi, mid, half, x_hi > x_lo, tau);
#endif
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int i = base + 64 * t + lane;
#ifdef VAMP_SKIP_EXP
            const float m = 1.0f - tau[t];
#else
            const float m = __expf(-tau[t]);
#endif
            const float r = (f[i] - m) * wt[i];
            chi += (double)r * (double)r;
        }
    }
}

struct PixPtrs {
    const double* x; const double* f; const double* wt;       // fp64 copies
    const float* xf; const float* ff; const float* wtf;       // fp32 copies (may be null)
};

// Sum over the walker's pixels of ((f - m) w)^2, on every lane.  One walker per wavefront (LPW = 64):
// full tiles are dealt round-robin into PARTS classes, each class is summed over its tiles and
// over the wave, and the class sums are added in class order -- by this wavefront (SPLIT = false)
// or through `red` by the PARTS wavefronts of the workgroup (SPLIT = true, `part` = this wave's
// class): the same order of additions either way.  The tail (pixels beyond the last full tile) belongs to class 0.
template <bool F32, int MODE, class PK, bool TAB = use_tables<F32, MODE, PK>()>
__device__ __forceinline__ void sweep_class(const RegionDev& R, const typename PK::Lds& L, TileScratch& Sx, const double* __restrict__ dct,
                                            const PixPtrs& px, int lane, int base0, int full, int stride, bool tail, double& chi,
                                            const double* tab) {
    constexpr int TPIX = pixels_per_lane<PK>();      // (shadows the global: packed shapes may hold fewer pixels per lane)
    if constexpr (F32) {
        const float* x = px.xf + R.pix_off; const float* f = px.ff + R.pix_off; const float* wt = px.wtf + R.pix_off;
        if constexpr (VAMP_FARFIELD && PK::FF && MODE != VAMP_GAUSS3 && TPIX == 4)
            sweep_range_f32_ff<MODE, PK, use_tables32<F32, MODE, PK>()>(R, L, Sx, dct, x, f, wt, lane, base0, full, stride, chi,
                                                                        reinterpret_cast<const float*>(tab));
        else if (TPIX > 1) sweep_range_f32<MODE, PK, TPIX>(R, L, x, f, wt, lane, base0, full, stride, chi);
        if constexpr (PK::TAIL || TPIX == 1)
            if (tail) {
                const int from = TPIX > 1 ? full : 0;
                if constexpr (!PK::FF && TPIX > 1) {
                    // short regions are all "tail": one iteration with just enough pixels per lane (1..TPIX
                    // independent evaluation chains, all pixel loads in flight together) instead of up
                    // to TPIX dependent one-pixel rounds
                    const int nt = (R.P - from + PK::LPW - 1) / PK::LPW;
                    if constexpr (TPIX >= 4) { if (nt == 4) sweep_range_f32<MODE, PK, 4>(R, L, x, f, wt, lane, from, from + 1, 1 << 28, chi); }
                    if constexpr (TPIX >= 3) { if (nt == 3) sweep_range_f32<MODE, PK, 3>(R, L, x, f, wt, lane, from, from + 1, 1 << 28, chi); }
                    if (nt == 2) sweep_range_f32<MODE, PK, 2>(R, L, x, f, wt, lane, from, from + 1, 1 << 28, chi);
                    else if (nt == 1) sweep_range_f32<MODE, PK, 1>(R, L, x, f, wt, lane, from, from + 1, 1 << 28, chi);
                } else {
                    sweep_range_f32<MODE, PK, 1>(R, L, x, f, wt, lane, from, R.P, PK::LPW, chi);
                }
            }
    } else {
        const double* x = px.x + R.pix_off; const double* f = px.f + R.pix_off; const double* wt = px.wt + R.pix_off;
        if constexpr (VAMP_FARFIELD && PK::FF && MODE != VAMP_GAUSS3 && TPIX == 4) {
            if constexpr (VAMP_WIDE_NODES && TAB) {
                // lines far wider than a tile (stage_lines left their mask in the parameter block): none on a converged ensemble
                const unsigned long long wide_all =
                    (unsigned long long)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(L.theta[4 * PK::KCAP + 2]) & 0xffff));
                if (VAMP_MID_NODES || wide_all) sweep_range_ff<MODE, PK, TAB, true>(R, L, Sx, dct, x, f, wt, lane, base0, full, stride, chi, tab, wide_all);
                else sweep_range_ff<MODE, PK, TAB, false>(R, L, Sx, dct, x, f, wt, lane, base0, full, stride, chi, tab);
            } else {
                sweep_range_ff<MODE, PK, TAB>(R, L, Sx, dct, x, f, wt, lane, base0, full, stride, chi, tab);
            }
        } else if (TPIX > 1) sweep_range<MODE, PK, TPIX, TAB>(R, L, x, f, wt, lane, base0, full, stride, chi, tab);
        if constexpr (PK::TAIL || TPIX == 1)
            if (tail) {
                const int from = TPIX > 1 ? full : 0;
                if constexpr (!PK::FF && TPIX > 1) {
                    const int nt = (R.P - from + PK::LPW - 1) / PK::LPW;      // see the fp32 branch
                    if constexpr (TPIX >= 4) { if (nt == 4) sweep_range<MODE, PK, 4, TAB>(R, L, x, f, wt, lane, from, from + 1, 1 << 28, chi, tab); }
                    if constexpr (TPIX >= 3) { if (nt == 3) sweep_range<MODE, PK, 3, TAB>(R, L, x, f, wt, lane, from, from + 1, 1 << 28, chi, tab); }
                    if (nt == 2) sweep_range<MODE, PK, 2, TAB>(R, L, x, f, wt, lane, from, from + 1, 1 << 28, chi, tab);
                    else if (nt == 1) sweep_range<MODE, PK, 1, TAB>(R, L, x, f, wt, lane, from, from + 1, 1 << 28, chi, tab);
                } else {
                    sweep_range<MODE, PK, 1, TAB>(R, L, x, f, wt, lane, from, R.P, PK::LPW, chi, tab);
                }
            }
    }
}

// ---- blends: tables of a few lines at a time, optical depths in registers across the passes -------
// T pixels per lane (base + 64 t + lane) of lines k0 .. k0 + kn - 1, whose tables sit in `tab`
template <int MODE, class PK, int T>
__device__ __forceinline__ void blend_chunk(const RegionDev& R, const typename PK::Lds& L, const double* __restrict__ x, int lane,
                                            int base, int k0, int kn, const double* tab, double (&tau)[4]) {
    double xi[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int i = base + 64 * t + lane;
        xi[t] = x[i < R.P ? i : R.P - 1];
    }
    for (int kk = 0; kk < kn; ++kk) {
        const LineRec ln = L.line[k0 + kk];
        double X[T], H[T];
#pragma unroll
        for (int t = 0; t < T; ++t) X[t] = fabs(xi[t] - ln.c) * ln.s;
        tile_voigt<T, true>(ln, L.dtab[k0 + kk], X, H, tab + kk * vamp::TAB_LINE);
#pragma unroll
        for (int t = 0; t < T; ++t) tau[t] = fma(ln.amp, H[t], tau[t]);
    }
}
template <int MODE, class PK>
__device__ __forceinline__ void blend_chunk_n(const RegionDev& R, const typename PK::Lds& L, const double* __restrict__ x, int lane,
                                              int base, int tn, int k0, int kn, const double* tab, double (&tau)[4]) {
    if (tn >= 4) blend_chunk<MODE, PK, 4>(R, L, x, lane, base, k0, kn, tab, tau);
    else if (tn == 3) blend_chunk<MODE, PK, 3>(R, L, x, lane, base, k0, kn, tab, tau);
    else if (tn == 2) blend_chunk<MODE, PK, 2>(R, L, x, lane, base, k0, kn, tab, tau);
    else if (tn == 1) blend_chunk<MODE, PK, 1>(R, L, x, lane, base, k0, kn, tab, tau);
}
constexpr int BLEND_MAX_PIXELS = 512;     // two chunks of up to 4 pixels per lane
template <int MODE, class PK>
__device__ __forceinline__ double sweep_blend_passes(const RegionDev& R, const typename PK::Lds& L, const PixPtrs& px, int lane,
                                                     double* tab) {
    static_assert(PK::WPB == 1 && PK::LPW == 64, "one wavefront per walker");
    constexpr int LP = PK::LINES_PER_PASS;
    const double* __restrict__ x = px.x + R.pix_off;
    const double* __restrict__ f = px.f + R.pix_off;
    const double* __restrict__ wt = px.wt + R.pix_off;
    const int P = R.P, K = R.K;
    const int nt = (P + 63) >> 6;                  // pixels per lane, <= 8
    const int tn0 = nt < 4 ? nt : 4, tn1 = nt - tn0;
    double tau0[4] = {0.0, 0.0, 0.0, 0.0}, tau1[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += LP) {
        const int kn = K - k0 < LP ? K - k0 : LP;
        // the tables of this pass: one (line, interval) pair per lane and round
        for (int e = lane; e < kn * vamp::TAB_NI; e += 64) {
            const int kk = e / vamp::TAB_NI, i = e % vamp::TAB_NI, k = k0 + kk;
            vamp::taylor_table_row(i, L.line[k].y, L.dtab[k], L.line[k].pole, L.line[k].hy, tab + kk * vamp::TAB_LINE + i * vamp::TAB_NT);
        }
        group_barrier<PK>();
        blend_chunk_n<MODE, PK>(R, L, x, lane, 0, tn0, k0, kn, tab, tau0);
        if (tn1 > 0) blend_chunk_n<MODE, PK>(R, L, x, lane, 256, tn1, k0, kn, tab, tau1);
        group_barrier<PK>();                        // the next pass overwrites the tables
    }
    double chi = 0.0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = 256 * c + 64 * t + lane;
            if (64 * (4 * c + t) < P) {             // wave-uniform: this pixel slot exists
                const int idx = i < P ? i : P - 1;
                const double m = vamp::exp_taylor(-(c == 0 ? tau0[t] : tau1[t]));
                const double r = (f[idx] - m) * wt[idx];
                chi += i < P ? r * r : 0.0;
            }
        }
    }
    return wave_sum<64>(chi);
}

// fp32 blends: rows of every line built once (one (line, interval) pair per lane and round), then ONE sweep of the
// region's <= 512 pixels, 1..4 per lane and chunk, lines innermost: W4 regions I / II in the wings, a table look-up
// in the cores (tile_w4_tab)
template <int MODE, class PK, int T>
__device__ __forceinline__ void blend32_chunk(const RegionDev& R, const typename PK::Lds& L, const float* __restrict__ x,
                                              const float* __restrict__ f, const float* __restrict__ wt, int lane, int base,
                                              const float* tab32, double& chi) {
    float xi[T], tau[T];
    int idx[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int i = base + 64 * t + lane;
        idx[t] = i < R.P ? i : R.P - 1;
        xi[t] = x[idx[t]];
        tau[t] = 0.0f;
    }
    for (int k = 0; k < R.K; ++k) {
        const float c = L.linef[k][0], sc = L.linef[k][1], y = L.linef[k][2], a = L.linef[k][3];
        float X[T], H[T];
#pragma unroll
        for (int t = 0; t < T; ++t) X[t] = fabsf(xi[t] - c) * sc;
        tile_w4_tab<T>(y, X, H, tab32 + k * vamp::TAB32_LINE);
#pragma unroll
        for (int t = 0; t < T; ++t) tau[t] = fmaf(a, H[t], tau[t]);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float m = __expf(-tau[t]);
        const float r = (f[idx[t]] - m) * wt[idx[t]];
        chi += (base + 64 * t + lane) < R.P ? (double)r * (double)r : 0.0;
    }
}
template <int MODE, class PK>
__device__ __forceinline__ double sweep_blend32(const RegionDev& R, const typename PK::Lds& L, const PixPtrs& px, int lane, float* tab32) {
    static_assert(PK::WPB == 1 && PK::LPW == 64, "one wavefront per walker");
    const float* __restrict__ x = px.xf + R.pix_off;
    const float* __restrict__ f = px.ff + R.pix_off;
    const float* __restrict__ wt = px.wtf + R.pix_off;
    for (int e = lane; e < R.K * vamp::TAB_NI; e += 64) {
        const int k = e / vamp::TAB_NI, i = e % vamp::TAB_NI;
        vamp::taylor_table_row32(i, L.line[k].y, L.dtab[k], L.line[k].pole, L.line[k].hy, tab32 + k * vamp::TAB32_LINE + i * vamp::TAB32_NT);
    }
    group_barrier<PK>();
    double chi = 0.0;
    for (int base = 0; base < R.P; base += 256) {
        const int nt = (R.P - base + 63) >> 6;
        if (nt >= 4) blend32_chunk<MODE, PK, 4>(R, L, x, f, wt, lane, base, tab32, chi);
        else if (nt == 3) blend32_chunk<MODE, PK, 3>(R, L, x, f, wt, lane, base, tab32, chi);
        else if (nt == 2) blend32_chunk<MODE, PK, 2>(R, L, x, f, wt, lane, base, tab32, chi);
        else blend32_chunk<MODE, PK, 1>(R, L, x, f, wt, lane, base, tab32, chi);
    }
    return wave_sum<64>(chi);
}

template <bool F32, int MODE, class PK = PackWide>
__device__ __forceinline__ double sweep_pixels(const RegionDev& R, const typename PK::Lds& L, TileScratch& Sx, const double* __restrict__ dct,
                                               const PixPtrs& px, int lane, int part, double* red, const double* tab) {
    constexpr int TILE = PK::LPW * pixels_per_lane<PK>();
    const int full = (R.P / TILE) * TILE;
    if constexpr (PK::SUBS > 1) {            // several walkers per wavefront: short regions, one pass
        double chi = 0.0;
        sweep_class<F32, MODE, PK>(R, L, Sx, dct, px, lane, 0, full, TILE, true, chi, tab);
        return wave_sum<PK::LPW>(chi);
    } else if constexpr (use_blend32<F32, MODE, PK>()) {
        if (R.P <= BLEND_MAX_PIXELS) return sweep_blend32<MODE, PK>(R, L, px, lane, reinterpret_cast<float*>(const_cast<double*>(tab)));
        double chi = 0.0;      // (longer than a blend: only when this shape is forced on a long region)
        sweep_class<F32, MODE, PK>(R, L, Sx, dct, px, lane, 0, full, TILE, true, chi, nullptr);
        return wave_sum<64>(chi);
    } else if constexpr (PK::LINES_PER_PASS > 0 && use_tables<F32, MODE, PK>()) {
        if (R.P <= BLEND_MAX_PIXELS) return sweep_blend_passes<MODE, PK>(R, L, px, lane, const_cast<double*>(tab));
        // longer than the registers hold (only when this shape is forced on a long region): every
        // line through the per-pixel evaluator, no tables
        double chi = 0.0;
        sweep_class<F32, MODE, PK, false>(R, L, Sx, dct, px, lane, 0, full, TILE, true, chi, nullptr);
        return wave_sum<64>(chi);
    } else if constexpr (PK::SPLIT && !PK::FF) {
        // a blend of a few hundred pixels: the group's wavefronts take contiguous shares of the region
        // (whole 16-pixel runs), each swept as full tiles + one iteration of 1..4 pixels per lane
        const int share = ((R.P + PK::WPB - 1) / PK::WPB + 15) & ~15;
        const int lo = part * share;
        RegionDev Rs = R;
        Rs.P = lo + share < R.P ? lo + share : R.P;           // this wave's pixels end here
        double chi = 0.0;
        if (lo < Rs.P)
            sweep_class<F32, MODE, PK>(Rs, L, Sx, dct, px, lane, lo, lo + ((Rs.P - lo) / TILE) * TILE, TILE, true, chi, tab);
        chi = wave_sum<64>(chi);
        if (lane == 0) red[part] = chi;
        group_barrier<PK>();
        double total = 0.0;
#pragma unroll
        for (int p = 0; p < PK::WPB; ++p) total += red[p];
        return total;
    } else if constexpr (PK::SPLIT) {
        double chi = 0.0;
        sweep_class<F32, MODE, PK>(R, L, Sx, dct, px, lane, part * TILE, full, PARTS * TILE, part == 0, chi, tab);
        chi = wave_sum<64>(chi);
        if (lane == 0) red[part] = chi;
        __syncthreads();
        double total = 0.0;
#pragma unroll
        for (int p = 0; p < PARTS; ++p) total += red[p];
        return total;
    } else {
        double total = 0.0;
        for (int p = 0; p < PARTS; ++p) {
            double chi = 0.0;
            sweep_class<F32, MODE, PK>(R, L, Sx, dct, px, lane, p * TILE, full, PARTS * TILE, p == 0, chi, tab);
            total += wave_sum<64>(chi);
        }
        return total;
    }
}

// log-likelihood from the reduced sum (both forms of SURVEY Appendix A)
template <class LDS>
__device__ __forceinline__ double loglike_from_sum(const RegionDev& R, const LDS& L, double ssum) {
    if (R.sample_sd) {
        const double sd = L.theta[R.D - 1];
        const double t = 1.0 / (sd * sd);
        return (double)R.P * 0.5 * log(t / (2.0 * vamp::PI)) - 0.5 * t * ssum;   // vpfits.py:39,341
    }
    return -0.5 * ssum + R.norm_const;                                            // vpfits.py:118
}

// log-posterior of the walker whose parameters sit in L.theta; `lane` = lane inside the walker's
// group.  Groups of one wave may leave early independently: everything below communicates only
// inside a group (xor shuffles with offsets < LPW) or through __any, which ignores inactive lanes.
template <bool F32, int MODE, class PK = PackWide>
__device__ __forceinline__ double wave_lnprob(const RegionDev& R, typename PK::Lds& L, TileScratch& Sx, const double* dct,
                                              const PixPtrs& px, int lane, double* chi_out, int part, double* red, double* tab) {
    const double lp = stage_lines<MODE, PK, use_tables<F32, MODE, PK>() && PK::LINES_PER_PASS == 0, use_tables32<F32, MODE, PK>(),
                                  use_blend32<F32, MODE, PK>()>(R, L, lane, F32, part, tab);
    VAMP_STAMP(3);
    if (!(lp > NEG_INF) || lp != lp) {       // outside the prior (or NaN): skip the sweep
        if (chi_out) *chi_out = __builtin_nan("");
        return NEG_INF;
    }
#ifdef VAMP_SKIP_SWEEP    // timing-only builds (tools/variants.py)
    const double ssum = L.line[0].y + L.theta[lane & 31] + (tab ? tab[lane] : 0.0);
#else
    const double ssum = sweep_pixels<F32, MODE, PK>(R, L, Sx, dct, px, lane, part, red, tab);
#endif
    VAMP_STAMP(4);
    if (chi_out) *chi_out = ssum;
    double v = lp + loglike_from_sum(R, L, ssum);
    if (v != v) v = NEG_INF;                 // NaN -> -inf (emcee convention)
    return v;
}

// ---------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------
template <bool F32, int MODE, class PK>
__global__ __launch_bounds__(PK::THREADS, (min_waves<F32, PK>())) void k_lnprob(const RegionDev* __restrict__ regions, int region, PixPtrs px,
                                                  long long W, const double* __restrict__ theta,
                                                  double* __restrict__ lnprob, double* __restrict__ chi2,
                                                  const int* __restrict__ region_list) {
    // region < 0: several regions in one launch (blockIdx.y indexes region_list, or the regions
    // themselves when it is null); theta holds the regions' [W, D_r] blocks one after the other
    // (block r starts at W * d_before(r)), lnprob / chi2 are [n_regions, W]
    const bool all = region < 0;
    if (all) region = region_list ? region_list[blockIdx.y] : (int)blockIdx.y;
    __shared__ typename PK::Lds lds[PK::SPLIT ? 1 : PK::WPB * PK::SUBS];
    __shared__ TileScratch scr[PK::FF ? PK::WPB : 1];
    __shared__ alignas(16) double dct[PK::FF ? ff_table_doubles<F32>() : 1];
    __shared__ double red[PARTS];
    __shared__ LineTables<table_doubles<F32, MODE, PK>()> tabs[PK::SPLIT ? 1 : PK::WPB];
    if constexpr (PK::FF) ff_fill_table<F32>(dct);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub = lane / PK::LPW, l = lane % PK::LPW;
    const long long w = PK::SPLIT ? (long long)blockIdx.x : ((long long)blockIdx.x * PK::WPB + wave) * PK::SUBS + sub;
    if (w >= W) return;
    const RegionDev R = regions[region];
    if (all) {
        theta += W * R.d_before;
        lnprob += (long long)region * W;
        if (chi2) chi2 += (long long)region * W;
    }
    typename PK::Lds& L = lds[PK::SPLIT ? 0 : wave * PK::SUBS + sub];
    if (!PK::SPLIT || wave == 0)
        for (int d = l; d < R.D; d += PK::LPW) L.theta[d] = theta[w * R.D + d];
    group_barrier<PK>();
    double chi;
    const double v = wave_lnprob<F32, MODE, PK>(R, L, scr[PK::FF ? wave : 0], dct, px, l, &chi, wave, red,
                                                tabs[PK::SPLIT ? 0 : wave].a);
    if (PK::SPLIT && wave != 0) return;
    if (l == 0) {
        lnprob[w] = v;
        if (chi2) chi2[w] = chi;
    }
}

// ---- MAP search on the device (row a9: VPfit.map_estimate / the MAP calls of find_bic, vpfits.py:352-358, 426) -------
// One workgroup per region runs the WHOLE Nelder-Mead search of that region -- scipy fmin's rules as restated in
// csrc/map_search.hpp, with the arithmetic of a simplex update shared with it (vamp::nm_*) -- in one launch: the
// host-driven form of the same search costs one launch + one stream synchronisation per ITERATION (~55 us each,
// ~60 000 per q1422 fit).  The workgroup has the shape k_lnprob runs for the region's launch class, so a point's
// objective has the bits vamp_lnprob gives it; its SLOTS walker slots evaluate the candidates of an iteration
// together (reflection, expansion and both contractions when SLOTS >= 4 -- the speculative form of the host
// search -- otherwise reflection first and the one point fmin asks for next).  The simplex lives in LDS
// up to D = 33, else in global scratch ((D + 1) x D doubles per region, L2-resident), values / order / centroid in LDS; the vertices are never
// moved, `ord` holds their order.  Regions are independent: no inter-workgroup communication.
constexpr int NM_INIT = 0, NM_CAND = 1, NM_CAND2 = 2, NM_SHRINK = 3;
constexpr int NM_LDS_DOUBLES = 34 * 33 + 2;      // simplices up to this size live in LDS (9 KB)
struct MapLds {
    double xbar[DMAX], worst[DMAX];   // centroid of the N best vertices, the worst vertex
    double f[DMAX + 1];               // objective values, ascending
    int ord[DMAX + 1];                // ord[k] = row of the k-th best vertex
    double fv[16];                    // values of the points evaluated in this round, by slot
    double redx[WAVES_PER_BLOCK], redf[WAVES_PER_BLOCK];
    double f0;                        // value at the start point
};
template <bool F32, int MODE, class PK>
__global__ __launch_bounds__(PK::THREADS, (min_waves<F32, PK>())) void k_map_search(
    const RegionDev* __restrict__ regions, PixPtrs px, const int* __restrict__ region_list, const double* __restrict__ theta0,
    const unsigned char* __restrict__ active, long long maxiter, long long maxfun, double xtol, double ftol,
    double* __restrict__ scratch, double* __restrict__ theta_best, long long* __restrict__ iterations) {
    constexpr int SLOTS = PK::WALKERS_PER_BLOCK, NT = PK::THREADS;
    static_assert(SLOTS <= 16 && PK::WPB <= WAVES_PER_BLOCK, "MapLds::fv / redx");
    __shared__ typename PK::Lds lds[PK::SPLIT ? 1 : PK::WPB * PK::SUBS];
    __shared__ TileScratch scr[PK::FF ? PK::WPB : 1];
    __shared__ alignas(16) double dct[PK::FF ? ff_table_doubles<F32>() : 1];
    __shared__ double red[PARTS];
    __shared__ LineTables<table_doubles<F32, MODE, PK>()> tabs[PK::SPLIT ? 1 : PK::WPB];
    __shared__ MapLds M;
    if constexpr (PK::FF) ff_fill_table<F32>(dct);
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int sub = lane / PK::LPW, l = lane % PK::LPW;
    const int slot = PK::SPLIT ? 0 : wave * PK::SUBS + sub;
    const int region = region_list ? region_list[blockIdx.x] : (int)blockIdx.x;
    const RegionDev R = regions[region];
    const int N = R.D;
    theta0 += R.d_before;
    theta_best += R.d_before;
    if (active && !active[region]) {            // not searched: returned unchanged (uniform over the workgroup)
        for (int d = tid; d < N; d += NT) theta_best[d] = theta0[d];
        if (tid == 0) iterations[region] = 0;
        return;
    }
    // rows of N doubles: the N + 1 vertices -- in LDS up to N = 33 (8 Voigt lines + sd), else in global scratch
    __shared__ double sim_lds[NM_LDS_DOUBLES];
    double* sim = (N + 1) * N <= NM_LDS_DOUBLES ? sim_lds : scratch + R.sim_off;
    typename PK::Lds& L = lds[slot];
    // start simplex: the start point and one vertex per coordinate (vamp::nm_start_coordinate)
    for (int e = tid; e < (N + 1) * N; e += NT) {
        const int k = e / N, d = e - k * N;
        const double v = theta0[d];
        sim[e] = (k >= 1 && d == k - 1) ? vamp::nm_start_coordinate(v) : v;
    }
    for (int k = tid; k <= N; k += NT) M.ord[k] = k;
    const long long fun_cap = maxfun > 0 ? maxfun : 200ll * N;     // scipy's default: 200 evaluations per dimension
    int phase = NM_INIT, base = 0, need = 0;
    long long it = 0, calls = 0;
    for (;;) {
        __syncthreads();            // the simplex (global), the values and the order (LDS) as the last round left them
        int n_pts = 0;
        if (phase == NM_CAND) {
            // fmin's limits, then its stopping test: max |x_k - x_0| <= xtol and max |f_0 - f_k| <= ftol
            if (calls >= fun_cap || it + 1 >= maxiter) break;         // fmin counts iterations from 1
            double mx = 0.0, mf = 0.0;
            for (int d = tid; d < N; d += NT) {
                const double r0 = sim[(long long)M.ord[0] * N + d];
                double sum = 0.0 + r0;          // (the host search starts its sum at +0)
                for (int k = 1; k < N; ++k) {
                    const double v = sim[(long long)M.ord[k] * N + d];
                    sum += v;
                    mx = fmax(mx, fabs(v - r0));
                }
                const double vw = sim[(long long)M.ord[N] * N + d];
                mx = fmax(mx, fabs(vw - r0));
                M.xbar[d] = sum / (double)N;
                M.worst[d] = vw;
            }
            for (int k = 1 + tid; k <= N; k += NT) mf = fmax(mf, fabs(M.f[0] - M.f[k]));
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                mx = fmax(mx, __shfl_xor(mx, off, 64));
                mf = fmax(mf, __shfl_xor(mf, off, 64));
            }
            if (lane == 0) { M.redx[wave] = mx; M.redf[wave] = mf; }
            __syncthreads();
            mx = M.redx[0]; mf = M.redf[0];
#pragma unroll
            for (int w = 1; w < PK::WPB; ++w) { mx = fmax(mx, M.redx[w]); mf = fmax(mf, M.redf[w]); }
            if (mx <= xtol && mf <= ftol) break;
            n_pts = SLOTS < 4 ? SLOTS : 4;
        } else if (phase == NM_CAND2) {
            n_pts = 1;
        } else {
            const int left = (phase == NM_INIT ? N + 1 : N) - base;
            n_pts = left < SLOTS ? left : SLOTS;
        }
        // this slot's point -> its parameter block in LDS, evaluated by the slot's lanes
        const bool mine = slot < n_pts;
        if (mine) {
            if (!PK::SPLIT || wave == 0) {
                if (phase == NM_CAND || phase == NM_CAND2) {
                    const int which = phase == NM_CAND ? slot : need;
                    for (int d = l; d < N; d += PK::LPW) L.theta[d] = vamp::nm_candidate(which, M.xbar[d], M.worst[d]);
                } else {
                    const double* row = sim + (long long)M.ord[(phase == NM_INIT ? 0 : 1) + base + slot] * N;
                    for (int d = l; d < N; d += PK::LPW) L.theta[d] = row[d];
                }
            }
            group_barrier<PK>();
            const double v = wave_lnprob<F32, MODE, PK>(R, L, scr[PK::FF ? wave : 0], dct, px, l, nullptr, wave, red,
                                                        tabs[PK::SPLIT ? 0 : wave].a);
            if (l == 0 && (!PK::SPLIT || wave == 0)) M.fv[slot] = vamp::nm_objective(v);
        }
        __syncthreads();
        // consume the round: every thread takes the same decisions from the same LDS values
        if (phase == NM_INIT || phase == NM_SHRINK) {
            const int first = (phase == NM_INIT ? 0 : 1) + base;
            if (tid < n_pts) M.f[first + tid] = M.fv[tid];
            base += n_pts;
            if (base < (phase == NM_INIT ? N + 1 : N)) continue;
            __syncthreads();
            if (phase == NM_INIT) calls = N + 1;
            else { calls += N; it += 1; }
            if (tid == 0) {
                if (phase == NM_INIT) M.f0 = M.f[0];
                for (int i = 1; i <= N; ++i) {              // insertion sort, stable: ties keep their order
                    const double fi = M.f[i];
                    const int oi = M.ord[i];
                    int j = i;
                    while (j > 0 && fi < M.f[j - 1]) { M.f[j] = M.f[j - 1]; M.ord[j] = M.ord[j - 1]; --j; }
                    M.f[j] = fi; M.ord[j] = oi;
                }
            }
            phase = NM_CAND;
            continue;
        }
        // NM_CAND / NM_CAND2: fmin's decision; a value this round did not hold is asked for (slots < 4)
        const int have = phase == NM_CAND ? n_pts : 0;           // candidates 0 .. have - 1 sit in fv[0 .. have)
        const double f_best = M.f[0], f_second = M.f[N - 1], f_worst = M.f[N];
        double fr;
        if (phase == NM_CAND) { fr = M.fv[0]; if (tid == 0) M.redx[0] = fr; }      // (kept for a second round)
        else fr = M.redx[0];
        int want;                               // the candidate whose value decides the update
        if (fr < f_best) want = 1;
        else if (fr < f_second) want = 0;
        else if (fr < f_worst) want = 2;
        else want = 3;
        double fw;
        if (want < have) fw = M.fv[want];
        else if (phase == NM_CAND2) fw = M.fv[0];
        else { phase = NM_CAND2; need = want; continue; }
        int take = -1;                          // candidate that replaces the worst vertex, or -1: shrink
        double ft = 0.0;
        calls += 1;
        if (want == 1) { calls += 1; if (fw < fr) { take = 1; ft = fw; } else { take = 0; ft = fr; } }
        else if (want == 0) { take = 0; ft = fr; }
        else if (want == 2) { calls += 1; if (fw <= fr) { take = 2; ft = fw; } }
        else { calls += 1; if (fw < f_worst) { take = 3; ft = fw; } }
        const int on = M.ord[N];                // row of the worst vertex
        __syncthreads();                        // (every thread has read fv / redx / f / ord before they change)
        if (take >= 0) {
            double* row = sim + (long long)on * N;
            for (int d = tid; d < N; d += NT) row[d] = vamp::nm_candidate(take, M.xbar[d], M.worst[d]);
            if (tid == 0) {                     // the new value finds its place among the N others (stable)
                int j = N;
                while (j > 0 && ft < M.f[j - 1]) { M.f[j] = M.f[j - 1]; M.ord[j] = M.ord[j - 1]; --j; }
                M.f[j] = ft; M.ord[j] = on;
            }
            it += 1;
            phase = NM_CAND;
        } else {
            for (int e = tid; e < N * N; e += NT) {
                const int k = 1 + e / N, d = e % N;
                double* row = sim + (long long)M.ord[k] * N;
                row[d] = vamp::nm_shrink(sim[(long long)M.ord[0] * N + d], row[d]);
            }
            phase = NM_SHRINK;
            base = 0;
        }
    }
    // the best vertex, unless it is worse than the start point
    const bool keep_start = M.f[0] > M.f0;
    const double* best = keep_start ? theta0 : sim + (long long)M.ord[0] * N;
    for (int d = tid; d < N; d += NT) theta_best[d] = best[d];
    if (tid == 0) iterations[region] = it;
}

// tau_k[P] and flux[P] for one parameter vector (one thread per pixel).  region < 0: every region in
// one launch (blockIdx.y = region): theta holds the regions' D_r-vectors one after the other,
// tau_comp the [K_r, P_r] blocks one after the other, flux_model is laid out like the pixels.
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_model(const RegionDev* __restrict__ regions, int region, PixPtrs px,
                                                 const double* __restrict__ theta, double* __restrict__ tau_comp,
                                                 double* __restrict__ flux_model) {
    __shared__ WaveLds lds[WAVES_PER_BLOCK];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool all = region < 0;
    if (all) region = blockIdx.y;
    const RegionDev R = regions[region];
    if ((int)(blockIdx.x * BLOCK) >= R.P) return;          // (uniform per block: regions differ in length)
    if (all) {
        theta += R.d_before;
        if (tau_comp) tau_comp += R.tau_off;
        if (flux_model) flux_model += R.pix_off;
    }
    WaveLds& L = lds[wave];
    for (int d = lane; d < R.D; d += 64) L.theta[d] = theta[d];
    __builtin_amdgcn_wave_barrier();
    (void)stage_lines<MODE, PackXL>(R, L, lane, false, 0);
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= R.P) return;
    const double xi = px.x[R.pix_off + i];
    double tau = 0.0;
    for (int k = 0; k < R.K; ++k) {
        double tk;
        if constexpr (MODE == VAMP_GAUSS3) {
            const double u = (xi - L.line[k].c) * L.line[k].s;
            tk = L.line[k].amp * exp(-0.5 * (u * u));
        } else {
            const double X = fabs(xi - L.line[k].c) * L.line[k].s;
            tk = L.line[k].amp * vamp::voigt_Hs(X, L.line[k].y, L.dtab[k], L.line[k].pole, L.line[k].hy);
        }
        if (tau_comp) tau_comp[(long long)k * R.P + i] = tk;
        tau += tk;
    }
    if (flux_model) flux_model[i] = exp(-tau);
}

// staged per-line records of one parameter vector (test hook for the parameter maps)
template <int MODE>
__global__ __launch_bounds__(64) void k_line_records(const RegionDev* __restrict__ regions, int region,
                                                     const double* __restrict__ theta, double* __restrict__ rec,
                                                     double* __restrict__ lnprior) {
    __shared__ WaveLds lds[1];
    const int lane = threadIdx.x & 63;
    const RegionDev R = regions[region];
    WaveLds& L = lds[0];
    for (int d = lane; d < R.D; d += 64) L.theta[d] = theta[d];
    __builtin_amdgcn_wave_barrier();
    const double lp = stage_lines<MODE, PackXL>(R, L, lane, false, 0);
    if (lane < R.K) {
        rec[5 * lane + 0] = L.line[lane].c;
        rec[5 * lane + 1] = L.line[lane].s;
        rec[5 * lane + 2] = L.line[lane].y;
        rec[5 * lane + 3] = (MODE == VAMP_GAUSS3) ? L.line[lane].amp : L.line[lane].amp * SQRT_PI;   // oracle's tau scale
        rec[5 * lane + 4] = L.line[lane].pole;
    }
    if (lane == 0) *lnprior = lp;
}

template <bool F32>
__global__ __launch_bounds__(BLOCK) void k_wofz(long long n, const double* __restrict__ x, const double* __restrict__ y,
                                                double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (F32) {
        out[i] = (double)vamp::humlicek_w4_re(fminf(fabsf((float)x[i]), vamp::W4_XMAX), (float)y[i]);
    } else {
        double dtab[vamp::DTAB_N];
        for (int k = 0; k < vamp::DTAB_N; ++k) dtab[k] = vamp::core_dtab_entry(k, y[i]);
        out[i] = vamp::voigt_H(fabs(x[i]), y[i], dtab, vamp::core_pole_factor(y[i]), vamp::core_hy(y[i]));
    }
}

// ---- counter-based RNG: Philox4x32-10 (Salmon et al., SC'11) ------------------------------
struct U4 { unsigned c0, c1, c2, c3; };
__device__ __forceinline__ U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c.c0;
        const unsigned long long p1 = 0xCD9E8D57ull * c.c2;
        U4 n;
        n.c0 = (unsigned)(p1 >> 32) ^ c.c1 ^ k0;
        n.c1 = (unsigned)p1;
        n.c2 = (unsigned)(p0 >> 32) ^ c.c3 ^ k1;
        n.c3 = (unsigned)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ double u53(unsigned hi, unsigned lo) {
    return (double)((((unsigned long long)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}
constexpr unsigned STREAM_MOVE = 0, STREAM_ACCEPT = 1, STREAM_SPLIT = 2;

// keyed bijection of [0, block): affine-multiply / xorshift rounds on the next power of two,
// cycle-walking back into range (DESIGN.md "red/blue split")
__device__ __forceinline__ unsigned split_perm(unsigned long long seed, unsigned step, unsigned chunk, unsigned region,
                                               unsigned slot, unsigned block) {
    const U4 r = philox4x32_10({chunk, step, STREAM_SPLIT, region}, (unsigned)seed, (unsigned)(seed >> 32));
    int bits = 32 - __builtin_clz((block - 1) | 1u);
    if (bits < 1) bits = 1;
    const unsigned long long mask = (1ull << bits) - 1ull;
    int sh = bits / 2;
    if (sh < 1) sh = 1;
    const unsigned long long m0 = ((unsigned long long)r.c0 << 1) | 1ull, m2 = ((unsigned long long)r.c2 << 1) | 1ull;
    unsigned long long v = slot;
    for (;;) {
        v = (v * m0 + r.c1) & mask;  v ^= v >> sh;
        v = (v * m2 + r.c3) & mask;  v ^= v >> sh;
        v = (v * 0x9E3779B1ull + (r.c0 ^ r.c3)) & mask;  v ^= v >> sh;
        if (v < block) return (unsigned)v;
    }
}

struct SamplerDev {
    const RegionDev* regions;
    int n_regions;
    long long W;                 // walkers per region
    int split_block;
    double a;
    unsigned long long seed;
    double* X;                   // concatenated [W, D_r] blocks
    double* lnp;                 // [n_regions * W]
    long long* n_accept;         // [n_regions * W]
    long long slot_begin, slot_end;   // this ctx's share of the n_regions*W/2 active slots
    const int* region_list;      // this launch's regions (a launch class of the ctx), or nullptr = all, in order
    double* pack;                // walker-sharded runs: [slot - slot_begin][D + 1] = the mover's row and lnprob
                                 // after the accept step (what the other devices need), else nullptr
    int bpr, n_cls_regions;      // several regions: workgroups per region (0: workgroups do not align with regions) and the
                                 // regions of this launch, for the region -> XCD mapping of k_half_step
    int wpr;                     // packed shapes, several regions: wavefronts per region = ceil((W/2) / SUBS), so that every
                                 // wavefront lies inside one region whatever W is (the last one of a region may have idle
                                 // groups); 0: slots are dealt to wavefronts linearly (one region, or one walker per wavefront)
};

// The draws of one mover: which walker holds active slot `a_loc` of `region` in this (step, half),
// its stretch factor, its partner from the frozen colour and log(u2) for the accept test.
struct MoveDraw { long long ws, wc; double z, logu; };
__device__ __forceinline__ MoveDraw draw_move(const SamplerDev& S, unsigned step, int half, int region, long long a_loc) {
    const long long halfW = S.W >> 1;
    const unsigned hb = (unsigned)(S.split_block >> 1);
    const unsigned chunk = (unsigned)(a_loc / hb);
    const unsigned pos = (unsigned)(a_loc % hb);
    MoveDraw d;
    // draws are keyed by the region's rng_id, not by its position in this context: a region follows
    // the same trajectory whichever device (and whichever subset of a spectrum's regions) holds it
    const unsigned rid = (unsigned)S.regions[region].rng_id;
    d.ws = (long long)chunk * S.split_block +
           split_perm(S.seed, step, chunk, rid, pos + (half ? hb : 0u), (unsigned)S.split_block);
    const long long gid = (long long)rid * S.W + d.ws;
    const unsigned k0 = (unsigned)S.seed, k1 = (unsigned)(S.seed >> 32);
    const U4 r = philox4x32_10({(unsigned)gid, step, ((unsigned)half << 8) | STREAM_MOVE, (unsigned)(gid >> 32)}, k0, k1);
    const double u1 = u53(r.c0, r.c1);
    const double t = (S.a - 1.0) * u1 + 1.0;
    d.z = t * t / S.a;
    const unsigned long long j = __umul64hi(((unsigned long long)r.c2 << 32) | r.c3, (unsigned long long)halfW);
    const unsigned cchunk = (unsigned)(j / hb);
    const unsigned cpos = (unsigned)(j % hb);
    d.wc = (long long)cchunk * S.split_block +
           split_perm(S.seed, step, cchunk, rid, cpos + (half ? 0u : hb), (unsigned)S.split_block);
    const U4 r2 = philox4x32_10({(unsigned)gid, step, ((unsigned)half << 8) | STREAM_ACCEPT, (unsigned)(gid >> 32)}, k0, k1);
    const double u2 = u53(r2.c0, r2.c1);
    d.logu = u2 > 0.0 ? log(u2) : NEG_INF;
    return d;
}

// Draws of a whole launch, one THREAD per mover.  A wavefront that serves one walker computes its
// draws on the scalar unit for free; packed four to a wavefront the same integer arithmetic runs
// on the vector unit with 4 distinct values in 64 lanes (~550 of ~1300 fixed instructions per
// wavefront, measured through profiles/r02_a_config3_pmc_summary.txt) -- so packed launches draw
// here, 64 distinct movers per instruction, and the half-step kernel reads the result.
__global__ __launch_bounds__(256) void k_draws(SamplerDev S, unsigned step, int half, long long n, int* __restrict__ ws,
                                               int* __restrict__ wc, double* __restrict__ z, double* __restrict__ logu,
                                               double* __restrict__ logz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long halfW = S.W >> 1;
    const long long slot = i + S.slot_begin;
    const int ridx = (int)(slot / halfW);
    const int region = S.region_list ? S.region_list[ridx] : ridx;
    const MoveDraw d = draw_move(S, step, half, region, slot - (long long)ridx * halfW);
    ws[i] = (int)d.ws;
    wc[i] = (int)d.wc;
    z[i] = d.z;
    logu[i] = d.logu;
    logz[i] = log(d.z);
}

constexpr int DRAW_INLINE = 0, DRAW_HOST = 1, DRAW_PRE = 2;      // where a mover's draws come from (k_half_step)
#ifndef VAMP_UNIFORM_ACCEPT
#define VAMP_UNIFORM_ACCEPT 1
#endif
__device__ __forceinline__ double uniform_double(double v) {      // a wave-uniform value, moved to scalar registers
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// One mover of a half-step: proposal q = c - (c - s) z into the walker's parameter block in LDS, its log-posterior,
// the accept test and the state update.  `l` = lane inside the walker's group, `part` = the
// wavefront's index in a split group; pack_slot >= 0: the row this walker ends with also goes to the exchange buffer.
// Shared by k_half_step (one launch per half-step) and k_run_resident (the step loop inside the kernel): a walker
// follows the same arithmetic through either.
template <bool F32, int DRAWS, int MODE, class PK>
__device__ __forceinline__ void stretch_move(const SamplerDev& S, const RegionDev& R, typename PK::Lds& L, TileScratch& Sx,
                                             const double* dct, const PixPtrs& px, int l, int part, double* red, double* tab,
                                             long long ws, long long wc, double z, double logu, double logz, long long pack_slot) {
#ifdef VAMP_ROWS_CACHED    // timing-only builds: every row read hits a 64-row window (no HBM latency)
    double* Xs = S.X + R.theta_off + (ws & 63) * R.D;
    const double* Xc = S.X + R.theta_off + (wc & 63) * R.D;
    const long long wg = R.walker_off + (ws & 63);
#else
    double* Xs = S.X + R.theta_off + ws * R.D;
    const double* Xc = S.X + R.theta_off + wc * R.D;
    const long long wg = R.walker_off + ws;
#endif
    // the mover's current lnprob is requested together with the two rows: three random reads of a
    // state far larger than L2, one exposed round trip instead of two (it is needed only for the
    // accept test, and left there its miss is paid in full by the one or two waves a SIMD holds)
    VAMP_STAMP(1);
    double lnp_s = S.lnp[wg];
    if (!PK::SPLIT || part == 0)
        for (int d = l; d < R.D; d += PK::LPW) {
            const double c = Xc[d];
            L.theta[d] = c - (c - Xs[d]) * z;                 // q = c - (c - s) z
        }
    VAMP_STAMP(2);
#if VAMP_EARLY_LNP
    asm volatile("" : "+v"(lnp_s));                          // keep the read up here
#endif
    group_barrier<PK>();
    if constexpr (PK::LPW == 64 && VAMP_UNIFORM_ACCEPT) {
        // one walker per wavefront: the values the accept step needs after the sweep are the same in every lane -- hand
        // them to the scalar registers, where they do not compete with the sweep for the 128 vector registers of the
        // workgroup-per-walker shape (they were spilled around the sweep: 10 dwords per lane, 230 MB of scratch stores
        // per launch of the headline)
        lnp_s = uniform_double(lnp_s);
        logu = uniform_double(logu);
        z = uniform_double(z);
        logz = uniform_double(logz);
    }
    const double lnp_q = wave_lnprob<F32, MODE, PK>(R, L, Sx, dct, px, l, nullptr, part, red, tab);
    VAMP_STAMP(5);
    if (PK::SPLIT && part != 0) return;     // the group's first wave carries out the accept step
    if constexpr (DRAWS != DRAW_PRE) logz = log(z);
    const double diff = (double)(R.D - 1) * logz + lnp_q - lnp_s;
    const bool accept = logu < diff;                      // false for NaN
    if (pack_slot >= 0 && S.pack) {
        // active-colour exchange: the row this walker ends the half-step with, in slot order
        double* pk = S.pack + pack_slot * (long long)(R.D + 1);
        for (int d = l; d < R.D; d += PK::LPW) pk[d] = accept ? L.theta[d] : Xs[d];
        if (l == 0) pk[R.D] = accept ? lnp_q : lnp_s;
    }
    if (accept) {
        for (int d = l; d < R.D; d += PK::LPW) Xs[d] = L.theta[d];
        if (l == 0) {
            S.lnp[wg] = lnp_q;
            S.n_accept[wg] += 1;
        }
    }
    VAMP_STAMP(6);
}

// One half-step of the stretch move, one wavefront per active walker.
//   DRAWS = DRAW_INLINE: Philox in-kernel; DRAW_HOST: every draw supplied by the host for ONE region
//   (deterministic-parity hook); DRAW_PRE: read from the arrays k_draws filled for this launch.
// Wavefronts per SIMD are set per shape (Pack::MIN_WAVES): the tile code is latency-bound in places (LDS and
// scalar-cache round trips); the headline shape measured 4.29 / 3.46 / 3.28 ms at 2 / 3 / 4 of them
template <bool F32, int DRAWS, int MODE, class PK>
__global__ __launch_bounds__(PK::THREADS, (min_waves<F32, PK>())) void k_half_step(SamplerDev S, PixPtrs px, unsigned step, int half, int ext_region,
                                                     long long ext_n, const int* __restrict__ ext_active,
                                                     const int* __restrict__ ext_partner, const double* __restrict__ ext_z,
                                                     const double* __restrict__ ext_logu, const double* __restrict__ ext_logz) {
    constexpr bool EXT = DRAWS == DRAW_HOST;
    __shared__ typename PK::Lds lds[PK::SPLIT ? 1 : PK::WPB * PK::SUBS];
    __shared__ TileScratch scr[PK::FF ? PK::WPB : 1];
    __shared__ alignas(16) double dct[PK::FF ? ff_table_doubles<F32>() : 1];
    __shared__ double red[PARTS];
    __shared__ LineTables<table_doubles<F32, MODE, PK>()> tabs[PK::SPLIT ? 1 : PK::WPB];
    if constexpr (PK::FF) ff_fill_table<F32>(dct);
    const int lane = threadIdx.x & 63;
    // the wave index is the same in every lane: say so, and the draws below (Philox rounds, the
    // split bijection -- all integer) run on the scalar unit when a wave serves one walker
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane / PK::LPW, l = lane % PK::LPW;
    const long long halfW = S.W >> 1;
    // first walker of this wave (a split workgroup: the one walker all of its waves serve)
    // Workgroups are dealt round-robin to the eight XCDs (workgroup b runs on XCD b % 8), each with its own L2.  In a
    // context of several regions the movers of a region read rows of that region only, so a mapping that keeps a
    // region on ONE XCD (-DVAMP_XCD_MAP=1: region i of the launch on XCD i % 8, vamp::plan::xcd_map) fetches its state
    // into one L2 instead of up to eight.  MEASURED on q1422 (profiles/r04_d_xcd_mapping.txt): FETCH_SIZE per half-step
    // 782 -> 309 MB (TCC misses 11.8 M -> 5.0 M) as predicted -- and 3.64 ms against 3.37 per half-step in fp64, 1.84
    // against 1.71 in fp32: regions differ in cost (the 54 blends: 3 .. 8 lines over 96 .. 512 px, six or seven per
    // XCD), the round-robin deal balances them and the mapping does not; contiguous eighths of the launch per XCD
    // were worse still (4.13 ms).  These kernels are bound by instruction issue, not by the row gathers.  Off.
    // (One region -- the headline: every workgroup reads the same spectrum and random rows of one state: nothing to align.)
    long long b = blockIdx.x;
#if VAMP_XCD_MAP
    if (DRAWS != DRAW_HOST && S.bpr > 0) b = vamp::plan::xcd_map(b, S.n_cls_regions, S.bpr);   // (the last n % 8 regions keep their place)
#endif
    long long slot0 = PK::SPLIT ? b : (b * PK::WPB + wave) * PK::SUBS;
    if (PK::SUBS > 1 && DRAWS != DRAW_HOST && S.wpr > 0) {
        // several regions, several walkers per wavefront: wavefront g serves SUBS consecutive movers of region
        // g / wpr; the groups beyond the region's W/2 movers idle (below: slot0 + sub leaves the region's range)
        const long long g = b * PK::WPB + wave;
        const long long ri = g / S.wpr;
        slot0 = ri * halfW + (g - ri * S.wpr) * PK::SUBS;
        if (slot0 + sub >= (ri + 1) * halfW) return;
    }
    long long slot = slot0 + sub;
    int region;
    long long ws, wc;            // local walker ids (within the region) of mover and partner
    double z, logu, logz = 0.0;
    if constexpr (DRAWS == DRAW_HOST) {
        if (slot >= ext_n) return;
        region = ext_region;
        ws = ext_active[slot]; wc = ext_partner[slot]; z = ext_z[slot]; logu = ext_logu[slot];
    } else {
        const long long i = slot;                     // index inside this launch
        slot += S.slot_begin;
        if (slot >= S.slot_end) return;
        // every walker of a wave lies in one region (one region in all, or wavefronts dealt per region: S.wpr):
        // keep the region description in scalar registers
        const int ridx = __builtin_amdgcn_readfirstlane((int)((slot0 + S.slot_begin) / halfW));
        region = __builtin_amdgcn_readfirstlane(S.region_list ? S.region_list[ridx] : ridx);
        if constexpr (DRAWS == DRAW_PRE) {
            ws = ext_active[i]; wc = ext_partner[i]; z = ext_z[i]; logu = ext_logu[i]; logz = ext_logz[i];
        } else {
            const MoveDraw d = draw_move(S, step, half, region, slot - (long long)ridx * halfW);
            ws = d.ws; wc = d.wc; z = d.z; logu = d.logu;
        }
    }
    const RegionDev R = S.regions[region];
    typename PK::Lds& L = lds[PK::SPLIT ? 0 : wave * PK::SUBS + sub];
    VAMP_STAMP(0);
    stretch_move<F32, DRAWS, MODE, PK>(S, R, L, scr[PK::FF ? wave : 0], dct, px, l, wave, red, tabs[PK::SPLIT ? 0 : wave].a, ws, wc, z, logu,
                                       logz, EXT ? -1ll : slot - S.slot_begin);
}

// ---- device-resident step loop (rows a8 / a10 in the regime the reference lives in) -----------------------------------
// The reference's fits are a few thousand iterations of a 4..33-parameter model (vpfits.py:417-428, vpregion.py:42); as
// ensembles of 32..128 walkers per region a launch per half-step is latency-bound (two launches per step, ~10 us each,
// most of it dispatch and the wavefront's own critical path).  Regions are independent posteriors, so ONE workgroup per
// region can run the region's whole step loop: its compute wavefronts take the movers of a half-step in rounds
// (stretch_move: the arithmetic of k_half_step), the half-step barrier is __syncthreads(), the kept samples are written
// from the kernel, and an extra wavefront draws -- one mover per lane, the same Philox keys -- the NEXT half-step's draws
// into an LDS double buffer while the others move.  The chain equals the launch-per-half-step chain bit for bit
// (tests/test_gpu_parity.py::test_resident_step_loop_equals_launch_per_half_step).  No inter-workgroup communication.
constexpr size_t RES_MAX_LDS = 150 * 1024;  // dynamic LDS of one resident workgroup (a workgroup may take all 160 KB of its CU)
constexpr int RES_MAX_REGIONS = 256;        // automatic policy: at most one resident workgroup per compute unit
struct alignas(16) ResDraws {
    double z[RES_MAX_MOVERS], logu[RES_MAX_MOVERS], logz[RES_MAX_MOVERS];
    int ws[RES_MAX_MOVERS], wc[RES_MAX_MOVERS];
};
// layout of the workgroup's dynamic LDS for nw compute wavefronts (host and device compute it alike)
template <bool F32, int MODE, class PK>
struct ResLayout {
    static constexpr size_t up(size_t v) { return (v + 15) & ~(size_t)15; }
    using Tabs = LineTables<table_doubles<F32, MODE, PK>()>;
    static constexpr size_t dct_bytes = PK::FF ? up(ff_table_doubles<F32>() * sizeof(double)) : 0;
    static constexpr size_t off_draws = dct_bytes;
    static constexpr size_t off_red = off_draws + 2 * sizeof(ResDraws);
    __host__ __device__ static constexpr size_t off_scr(int nw) { return off_red + up((size_t)nw * PARTS * sizeof(double)); }
    __host__ __device__ static constexpr size_t off_tabs(int nw) { return off_scr(nw) + (PK::FF ? up((size_t)nw * sizeof(TileScratch)) : 0); }
    __host__ __device__ static constexpr size_t off_lds(int nw) { return off_tabs(nw) + up((size_t)nw * sizeof(Tabs)); }
    __host__ __device__ static constexpr size_t total(int nw) { return off_lds(nw) + up((size_t)nw * PK::SUBS * sizeof(typename PK::Lds)); }
};
template <bool F32, int MODE, class PK>
__global__ __launch_bounds__(64 * (RES_MAX_WAVES + 1), (min_waves<F32, PK>())) void k_run_resident(
    SamplerDev S, PixPtrs px, unsigned step0, long long n_steps, int thin, double* __restrict__ chain, double* __restrict__ lchain,
    long long total_theta, long long total_walkers) {
    static_assert(!PK::SPLIT || PK::WPB == 1, "a walker's group is one wavefront or a part of one");
    using LY = ResLayout<F32, MODE, PK>;
    extern __shared__ __align__(16) unsigned char res_raw[];
    const int nw = (int)(blockDim.x >> 6) - 1;                   // compute wavefronts; wavefront nw draws
    double* dct = reinterpret_cast<double*>(res_raw);
    ResDraws* draws = reinterpret_cast<ResDraws*>(res_raw + LY::off_draws);
    if constexpr (PK::FF) ff_fill_table<F32>(dct);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane / PK::LPW, l = lane % PK::LPW;
    const bool drawer = wave == nw;
    const int region = __builtin_amdgcn_readfirstlane(S.region_list ? S.region_list[blockIdx.x] : (int)blockIdx.x);
    const RegionDev R = S.regions[region];
    const int halfW = (int)(S.W >> 1);
    const int slots = nw * PK::SUBS;
    const int wv = drawer ? 0 : wave;                             // (the drawing wavefront owns no walker slot)
    double* red = reinterpret_cast<double*>(res_raw + LY::off_red) + wv * PARTS;
    TileScratch& Sx = reinterpret_cast<TileScratch*>(res_raw + LY::off_scr(nw))[PK::FF ? wv : 0];
    double* tab = reinterpret_cast<typename LY::Tabs*>(res_raw + LY::off_tabs(nw))[wv].a;
    typename PK::Lds& L = reinterpret_cast<typename PK::Lds*>(res_raw + LY::off_lds(nw))[wv * PK::SUBS + sub];
    // the draws of one half-step: which walker holds each active slot, its stretch factor, partner and log u
    auto draw_all = [&](unsigned step, int half, ResDraws& D) {
        for (int a = lane; a < halfW; a += 64) {
            const MoveDraw d = draw_move(S, step, half, region, a);
            D.ws[a] = (int)d.ws; D.wc[a] = (int)d.wc; D.z[a] = d.z; D.logu[a] = d.logu; D.logz[a] = log(d.z);
        }
    };
    if (drawer) draw_all(step0, 0, draws[0]);
    __syncthreads();
    long long kept = 0;
    int until_keep = thin;                  // (a countdown: a 64-bit modulo per step is ~100 scalar instructions)
    for (long long it = 0; it < n_steps; ++it) {
        const unsigned step = step0 + (unsigned)it;
        for (int half = 0; half < 2; ++half) {
            const ResDraws& D = draws[half];
            VAMP_STAMP(0);
            if (drawer) {                                          // the next half-step's draws, while the others move
                if (half == 0) draw_all(step, 1, draws[1]);
                else if (it + 1 < n_steps) draw_all(step + 1u, 0, draws[0]);
            } else {
                for (int a0 = 0; a0 < halfW; a0 += slots) {
                    const int a = a0 + wave * PK::SUBS + sub;
                    if (a < halfW)
                        stretch_move<F32, DRAW_PRE, MODE, PK>(S, R, L, Sx, dct, px, l, 0, red, tab, D.ws[a], D.wc[a], D.z[a], D.logu[a],
                                                              D.logz[a], -1ll);
                }
            }
            VAMP_STAMP(7);
            __syncthreads();            // the movers' rows are in place (global, workgroup scope), the next draws are complete
            VAMP_STAMP(8);
        }
        const bool keep_now = --until_keep == 0;
        if (keep_now) until_keep = thin;
        if (keep_now && (chain || lchain)) {
            if (chain) {
                const double* __restrict__ src = S.X + R.theta_off;
                double* __restrict__ dst = chain + kept * total_theta + R.theta_off;
                for (long long e = tid; e < S.W * R.D; e += blockDim.x) dst[e] = src[e];
            }
            if (lchain) {
                const double* __restrict__ src = S.lnp + R.walker_off;
                double* __restrict__ dst = lchain + kept * total_walkers + R.walker_off;
                for (long long e = tid; e < S.W; e += blockDim.x) dst[e] = src[e];
            }
            ++kept;
            __syncthreads();            // (the next half-step moves rows this copy reads)
        }
    }
}

// Walker-sharded runs: rows of the active colour gathered from every device, in slot order
// (recv[i] = slot first_slot + i, D + 1 doubles: position and lnprob), are written to the walkers
// that hold those slots in this (step, half).  16 lanes per row; rows [own_lo, own_hi) are this
// device's own movers, already in place.
__global__ __launch_bounds__(256) void k_scatter_rows(SamplerDev S, const double* __restrict__ recv, unsigned step, int half,
                                                      long long first_slot, long long n_rows, long long own_lo, long long own_hi) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    if (i >= n_rows || (i >= own_lo && i < own_hi)) return;
    const RegionDev& R = S.regions[0];
    const int D = R.D;
    const long long slot = first_slot + i;
    const unsigned hb = (unsigned)(S.split_block >> 1);
    const unsigned chunk = (unsigned)(slot / hb), pos = (unsigned)(slot % hb);
    const long long ws = (long long)chunk * S.split_block +
                         split_perm(S.seed, step, chunk, (unsigned)R.rng_id, pos + (half ? hb : 0u), (unsigned)S.split_block);
    const double* src = recv + i * (long long)(D + 1);
    double* dst = S.X + R.theta_off + ws * D;
    for (int d = l; d < D; d += 16) dst[d] = src[d];
    if (l == 0) S.lnp[R.walker_off + ws] = src[D];
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (void)hipGetLastError();   /* reported here: do not leave it for the next library (RCCL) to find */ \
            return fail(e_ == hipErrorOutOfMemory ? VAMP_ERR_NOMEM : VAMP_ERR_HIP,            \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                   \
        }                                                                                     \
    } while (0)

// device allocation released on every exit path of a host function
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// run the statement with `M` bound to the compile-time parameterisation that matches runtime
// `mode` and `PK` to the packing (small = <16 lanes, 8 lines> per walker, else <64, 16>)
#define VAMP_FOR_MODE_(mode, ...)                                               \
    do {                                                                        \
        if ((mode) == VAMP_GAUSS3) { constexpr int M = VAMP_GAUSS3; __VA_ARGS__; } \
        else if ((mode) == VAMP_VOIGT4) { constexpr int M = VAMP_VOIGT4; __VA_ARGS__; } \
        else { constexpr int M = VAMP_NBZ3; __VA_ARGS__; }                      \
    } while (0)
#define VAMP_FOR_MODE(mode, ...) VAMP_FOR_MODE_(mode, __VA_ARGS__)
// launch shapes (see struct Pack): what one launch class of a context runs
enum Shape { SH_SMALL = 0, SH_MID = 1, SH_WIDE = 2, SH_WIDE_FULL = 3, SH_SPLIT = 4, SH_SPLIT_FULL = 5, SH_SMALL2 = 6, SH_XL = 7 };
#define VAMP_FOR_MODE_PK(mode, shape, ...)                                              \
    do {                                                                        \
        if ((shape) == SH_SMALL) { using PK = PackSmall; VAMP_FOR_MODE_(mode, __VA_ARGS__); } \
        else if ((shape) == SH_SMALL2) { using PK = PackSmall2; VAMP_FOR_MODE_(mode, __VA_ARGS__); } \
        else if ((shape) == SH_MID) { using PK = PackMid; VAMP_FOR_MODE_(mode, __VA_ARGS__); } \
        else if ((shape) == SH_XL) { using PK = PackXL; VAMP_FOR_MODE_(mode, __VA_ARGS__); } \
        else if ((shape) == SH_SPLIT_FULL) { using PK = PackSplitFull; VAMP_FOR_MODE_(mode, __VA_ARGS__); } \
        else if ((shape) == SH_SPLIT) { using PK = PackSplit; VAMP_FOR_MODE_(mode, __VA_ARGS__); } \
        else if ((shape) == SH_WIDE_FULL) { using PK = PackWideFull; VAMP_FOR_MODE_(mode, __VA_ARGS__); } \
        else { using PK = PackWide; VAMP_FOR_MODE_(mode, __VA_ARGS__); }        \
    } while (0)
inline long long shape_walkers_per_block(int sh) {
    return sh == SH_SMALL ? PackSmall::WALKERS_PER_BLOCK : sh == SH_SMALL2 ? PackSmall2::WALKERS_PER_BLOCK : sh == SH_MID ? PackMid::WALKERS_PER_BLOCK
           : sh == SH_XL ? PackXL::WALKERS_PER_BLOCK : (sh == SH_SPLIT || sh == SH_SPLIT_FULL) ? 1 : PackWide::WALKERS_PER_BLOCK;
}
inline long long shape_waves(int sh) { return (long long)(sh == SH_SMALL ? PackSmall::WPB : sh == SH_SMALL2 ? PackSmall2::WPB : sh == SH_MID ? PackMid::WPB
                                                           : sh == SH_XL ? PackXL::WPB : PackWide::WPB); }
inline unsigned shape_threads(int sh) {
    return sh == SH_SMALL ? PackSmall::THREADS : sh == SH_SMALL2 ? PackSmall2::THREADS : sh == SH_MID ? PackMid::THREADS
           : sh == SH_XL ? PackXL::THREADS : (sh == SH_SPLIT || sh == SH_SPLIT_FULL) ? PackSplit::THREADS : PackWide::THREADS;
}

// A launch class: the regions of a context that one kernel shape serves.  Real spectra mix many
// short single-line regions (four walkers per wavefront) with a few long blends (a wavefront per
// walker with Taylor tables); each class is one launch per half-step over its own region list.
using vamp::plan::CK_SMALL; using vamp::plan::CK_MID; using vamp::plan::CK_WIDE; using vamp::plan::CK_SMALL2; using vamp::plan::CK_XL;
struct LaunchClass {
    int kind = CK_WIDE;
    std::vector<int> regions;
    int* list_d = nullptr;       // device copy of `regions`; nullptr when the class is every region in order
};

}  // namespace

struct vamp_ctx {
    int device = 0;
    int dtype = VAMP_F64;
    int wofz_kind = VAMP_WOFZ_ACCURATE;
    bool f32 = false;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // regions
    int n_regions = 0;
    int mode = VAMP_VOIGT4;
    int packing = 0;       // requested: 0 = auto, 16 or 64 lanes per walker, 256 = a 4-wave workgroup per walker
    int min_tiles = 0;     // full 256-pixel tiles of the shortest region
    bool full_tiles = false;   // every region's pixel count is a multiple of 64 * TPIX
    std::vector<LaunchClass> classes;      // partition of the regions by kernel shape (vamp_set_regions)
    std::vector<int> class_of;             // region -> index into classes
    // the partition for SMALL ensembles (<= RES_MAX_MOVERS movers per region): the short-region classes merged into one
    // (vamp::plan::plan_classes, merged) -- a class is a launch per half-step, and there launches are what a half-step costs
    std::vector<LaunchClass> classes_small;
    std::vector<int> class_of_small;
    // draws of packed launches (k_draws), grown on demand
    int *dr_ws = nullptr, *dr_wc = nullptr;
    double *dr_z = nullptr, *dr_lu = nullptr, *dr_lz = nullptr;
    long long dr_cap = 0;
    // launch classes of one half-step run concurrently, each on its own stream (forked from and joined
    // to the ctx stream with events): a class of low-occupancy blends and a class of register-bound
    // short regions fill each other's stalls
    bool concurrent_classes = true;
    std::vector<hipStream_t> cls_stream;
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_join;
    std::vector<RegionDev> regions_h;
    RegionDev* regions_d = nullptr;
    long long n_pix = 0;
    double *x_d = nullptr, *f_d = nullptr, *wt_d = nullptr;
    float *xf_d = nullptr, *ff_d = nullptr, *wtf_d = nullptr;
    // sampler
    bool sampler_ready = false;
    long long W = 0, total_theta = 0, total_walkers = 0;
    int split_block = 0;
    double a = 2.0;
    unsigned long long seed = 0;
    long long step = 0;
    double* X_d = nullptr;
    double* lnp_d = nullptr;
    bool X_ext = false;
    long long* nacc_d = nullptr;
    long long slot_begin = 0, slot_end = 0;      // part 0 (the whole share when shard_parts == 1)
    int shard_rank = 0, shard_world = 1, shard_parts = 1;
    long long part_slots = 0, part_stride = 0;   // slots per part; distance between this rank's parts
    // walker-sharded runs: active-colour exchange (pack -> all-gather -> scatter), see vamp_comm_*
    double* send_d = nullptr;        // [parts][part_slots][D + 1]
    double* recv_d = nullptr;        // [parts][world * part_slots][D + 1]
    std::vector<unsigned> part_step; // (step, half) of the last launch of every part
    std::vector<int> part_half;
    void* comm = nullptr;            // ncclComm_t (RCCL), one per ctx
    int comm_rank = 0, comm_world = 1;
    hipStream_t comm_stream = nullptr;
    std::vector<hipEvent_t> ev_kernel, ev_scatter;   // per part: kernel done / rows scattered
    // exchange timing (vamp_exchange_timing): event pairs around all-gather + scatter on the stream they run on
    std::vector<std::pair<hipEvent_t, hipEvent_t>> xev;
    size_t xev_used = 0;
    double xtiming_ms = 0.0;
    long long xtiming_n = 0;
    // grow-only scratch of vamp_lnprob (the MAP optimiser calls it thousands of times with W = 1)
    double *sc_th = nullptr, *sc_lp = nullptr, *sc_chi = nullptr;
    size_t sc_th_cap = 0, sc_w_cap = 0;
    // small evaluations (the MAP searches: a few points per region, thousands of times): pinned host memory the
    // kernel reads and writes directly -- no staging copies, one launch and one synchronisation per call
    double *pin_th = nullptr, *pin_out = nullptr;
    size_t pin_th_cap = 0, pin_out_cap = 0;
    // the MAP searches on the device (k_map_search): start points, flags, simplices, results; grow-only
    double *map_th_d = nullptr, *map_best_d = nullptr, *map_sim_d = nullptr;
    unsigned char* map_act_d = nullptr;
    long long* map_it_d = nullptr;
    size_t map_th_cap = 0, map_sim_cap = 0, map_r_cap = 0;
    // run-time options (vamp_ctx_set_option)
    int opt_map_device = 1;          // 1: vamp_map_all runs k_map_search; 0: the host-driven search, one launch per iteration
    int opt_resident = 1;            // 1: small ensembles are stepped by the device-resident loop where it pays (resident_eligible);
                                     // 0: never; 2: wherever the kernel can run
    // scratch for the ext hook
    int *ext_act_d = nullptr, *ext_par_d = nullptr;
    double *ext_z_d = nullptr, *ext_lu_d = nullptr;
    long long ext_cap = 0;
    // kernel timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t ev_used = 0;
    double timing_ms = 0.0;
    long long timing_launches = 0;

    PixPtrs pix() const { return PixPtrs{x_d, f_d, wt_d, xf_d, ff_d, wtf_d}; }
};

namespace {

int free_regions(vamp_ctx* c) {
    for (void* p : {(void*)c->regions_d, (void*)c->x_d, (void*)c->f_d, (void*)c->wt_d, (void*)c->xf_d, (void*)c->ff_d,
                    (void*)c->wtf_d})
        if (p) (void)hipFree(p);
    for (std::vector<LaunchClass>* part : {&c->classes, &c->classes_small}) {
        for (LaunchClass& cl : *part)
            if (cl.list_d) (void)hipFree(cl.list_d);
        part->clear();
    }
    c->class_of.clear();
    c->class_of_small.clear();
    c->regions_d = nullptr;
    c->x_d = c->f_d = c->wt_d = nullptr;
    c->xf_d = c->ff_d = c->wtf_d = nullptr;
    c->n_regions = 0;
    c->regions_h.clear();
    return 0;
}

int free_sampler(vamp_ctx* c) {
    if (c->X_d && !c->X_ext) (void)hipFree(c->X_d);
    if (c->lnp_d && !c->X_ext) (void)hipFree(c->lnp_d);
    if (c->nacc_d) (void)hipFree(c->nacc_d);
    if (c->send_d) (void)hipFree(c->send_d);
    if (c->recv_d) (void)hipFree(c->recv_d);
    c->send_d = c->recv_d = nullptr;
    c->X_d = nullptr;
    c->lnp_d = nullptr;
    c->nacc_d = nullptr;
    c->sampler_ready = false;
    return 0;
}

int flush_timing(vamp_ctx* c) {
    if (c->ev_used == 0) return 0;
    HIP_TRY(hipEventSynchronize(c->ev[c->ev_used - 1].second));
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[i].first, c->ev[i].second));
        c->timing_ms += ms;
        c->timing_launches += 1;
    }
    c->ev_used = 0;
    return 0;
}

int flush_exchange_timing(vamp_ctx* c) {
    if (c->xev_used == 0) return 0;
    HIP_TRY(hipEventSynchronize(c->xev[c->xev_used - 1].second));
    for (size_t i = 0; i < c->xev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(c->xev[i].second));
        HIP_TRY(hipEventElapsedTime(&ms, c->xev[i].first, c->xev[i].second));
        c->xtiming_ms += ms;
        c->xtiming_n += 1;
    }
    c->xev_used = 0;
    return 0;
}

// exchange buffers of the shard set by vamp_sampler_set_shard_parts: the movers of every part in slot order,
// position + lnprob per row (+ the per-part events when a communicator will run the exchange)
int ensure_part_events(vamp_ctx* c, int parts);
int alloc_exchange_buffers(vamp_ctx* c) {
    const int D = c->regions_h[0].D;
    HIP_TRY(hipMalloc(&c->send_d, vamp::plan::exchange_send_doubles(c->shard_parts, c->part_slots, D) * sizeof(double)));
    HIP_TRY(hipMalloc(&c->recv_d, vamp::plan::exchange_recv_doubles(c->shard_parts, c->shard_world, c->part_slots, D) * sizeof(double)));
    c->part_step.assign(c->shard_parts, 0u);
    c->part_half.assign(c->shard_parts, 0);
    if (c->comm) return ensure_part_events(c, c->shard_parts);
    return 0;
}

// per-part events of the overlapped exchange (kernel of a part done / its rows scattered)
int ensure_part_events(vamp_ctx* c, int parts) {
    while ((int)c->ev_kernel.size() < parts) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        c->ev_kernel.push_back(a);
        c->ev_scatter.push_back(b);
    }
    return 0;
}

// kernel shape of launch class `cl` for an ensemble with `per_region` movers per region and half-step and `total` movers
// per launch of the class -- UNSHARDED counts (W/2 of a walker-sharded single-region ensemble, whatever this device's share
// or piece of it), so that a shard runs the shape, the same bits, the whole ensemble runs on one device.
// The packed shapes (several walkers per wavefront: the fixed work of a wavefront shared) are the throughput shapes: launches
// that fill the chip (>= PACK_MIN_WALKERS movers), and the small ensembles of model-selection ladders (<= RES_MAX_MOVERS
// movers per region: tens of regions x tens of walkers, where they also feed the device-resident loop -- and single points:
// vamp_lnprob, the MAP search).  In between -- config 2: one region, 4096 walkers -- a launch cannot fill the chip and is
// bound by one wavefront's critical path: one walker per wavefront (14 us per half-step against 23 packed,
// profiles/r04_c_small_ensembles.txt).  All callers of one context ask with the same counts.
// the launch classes an ensemble of `per_region` movers per region and half-step runs in (callers of one context ask with
// the same count: the sampler and the resident loop with W / 2, vamp_lnprob(_all) of W points with W / 2, the MAP search with 1)
inline bool small_ensemble(long long per_region) { return per_region <= RES_MAX_MOVERS; }
const std::vector<LaunchClass>& partition_for(const vamp_ctx* c, long long per_region) {
    return small_ensemble(per_region) ? c->classes_small : c->classes;
}
const std::vector<int>& class_of_for(const vamp_ctx* c, long long per_region) {
    return small_ensemble(per_region) ? c->class_of_small : c->class_of;
}
int class_shape(const vamp_ctx* c, const LaunchClass& cl, long long per_region, long long total, bool packable) {
    if (cl.kind == CK_XL) return SH_XL;
    if (cl.kind == CK_MID) return SH_MID;
    if ((cl.kind == CK_SMALL || cl.kind == CK_SMALL2) && packable &&
        (c->packing == 16 || per_region <= RES_MAX_MOVERS || total >= PACK_MIN_WALKERS)) {
        // one- and two-line regions: eight lanes per walker is the throughput shape (six pixels per lane of a 44-pixel
        // region: 14 800 clocks of sweep per half-step, in-kernel stamps); a small ensemble is bound by exactly that critical
        // path and runs sixteen lanes per walker (three pixels per lane: 8 300; profiles/r04_c_small_ensembles.txt) -- as
        // part of the merged short-region class
        // (small ensembles never see CK_SMALL2: their partition has the short-region classes merged, partition_for)
        return cl.kind == CK_SMALL2 ? SH_SMALL2 : SH_SMALL;
    }
    const bool split = cl.kind == CK_WIDE && (c->packing == 256 || (c->packing == 0 && c->min_tiles >= 2 * PARTS));
    if (split) return c->full_tiles ? SH_SPLIT_FULL : SH_SPLIT;
    return c->full_tiles ? SH_WIDE_FULL : SH_WIDE;
}

int ensure_draw_buffers(vamp_ctx* c, long long n) {
    if (c->dr_cap >= n) return 0;
    for (void* p : {(void*)c->dr_ws, (void*)c->dr_wc, (void*)c->dr_z, (void*)c->dr_lu, (void*)c->dr_lz})
        if (p) (void)hipFree(p);
    c->dr_ws = c->dr_wc = nullptr;
    c->dr_z = c->dr_lu = c->dr_lz = nullptr;
    c->dr_cap = 0;
    HIP_TRY(hipMalloc(&c->dr_ws, n * sizeof(int)));
    HIP_TRY(hipMalloc(&c->dr_wc, n * sizeof(int)));
    HIP_TRY(hipMalloc(&c->dr_z, n * sizeof(double)));
    HIP_TRY(hipMalloc(&c->dr_lu, n * sizeof(double)));
    HIP_TRY(hipMalloc(&c->dr_lz, n * sizeof(double)));
    c->dr_cap = n;
    return 0;
}

// One half-step of this ctx's share (piece `part` of it): one launch per launch class, on the ctx
// stream.  ext: host-supplied draws for `ext_n` movers of `ext_region`.
int launch_half(vamp_ctx* c, int half, bool ext, int ext_region, long long ext_n, int part = 0) {
    SamplerDev S;
    S.regions = c->regions_d;
    S.n_regions = c->n_regions;
    S.W = c->W;
    S.split_block = c->split_block;
    S.a = c->a;
    S.seed = c->seed;
    S.X = c->X_d;
    S.lnp = c->lnp_d;
    S.n_accept = c->nacc_d;
    S.region_list = nullptr;
    S.slot_begin = S.slot_end = 0;
    S.wpr = S.bpr = S.n_cls_regions = 0;
    S.pack = (!ext && c->send_d) ? c->send_d + (long long)part * c->part_slots * (c->regions_h[0].D + 1) : nullptr;
    if (!ext && c->send_d) {
        c->part_step[part] = (unsigned)c->step;
        c->part_half[part] = half;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        if (c->ev_used == c->ev.size()) {
            if (c->ev.size() >= 4096) {
                int rc = flush_timing(c);
                if (rc) return rc;
            } else {
                hipEvent_t a, b;
                HIP_TRY(hipEventCreate(&a));
                HIP_TRY(hipEventCreate(&b));
                c->ev.emplace_back(a, b);
            }
        }
        e0 = c->ev[c->ev_used].first;
        e1 = c->ev[c->ev_used].second;
        c->ev_used++;
        HIP_TRY(hipEventRecord(e0, c->stream));
    }
    const unsigned step = (unsigned)c->step;
    const PixPtrs px = c->pix();
    const long long halfW = c->W / 2;
    const int* ni = nullptr;
    const double* nd = nullptr;
    // several walkers per wavefront need every wavefront inside one region: trivially so with one region, else the
    // wavefronts are dealt per region (SamplerDev::wpr); host-supplied draws run one walker per wavefront
    const bool packable = !ext;
    const std::vector<LaunchClass>& classes = partition_for(c, halfW);
    const std::vector<int>& class_of = class_of_for(c, halfW);
    const size_t ncls = classes.size();
    // the classes of a half-step on forked streams fill each other's stalls when the launches are big (config 3: 4.85 ->
    // 4.20 ms); for the small ensembles of a ladder the event traffic of fork and join costs more than it hides
    // (q1422 ladder, first rung: 73 -> 57 us per half-step without it)
    const bool fork = !ext && ncls > 1 && c->concurrent_classes && c->total_walkers / 2 >= PACK_MIN_WALKERS * 4;
    if (fork) {
        while (c->cls_stream.size() < ncls - 1) {
            hipStream_t st;
            hipEvent_t ev;
            HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            c->cls_stream.push_back(st);
            c->ev_join.push_back(ev);
        }
        if (!c->ev_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
    }
    // draws of the packed classes: one slice of the buffers per class (the classes may overlap in time)
    long long draw_total = 0;
    if (!ext)
        for (const LaunchClass& cl : classes) draw_total += (c->n_regions == 1) ? c->total_walkers : (long long)cl.regions.size() * halfW;
    long long draw_off = 0;
    for (size_t ci = 0; ci < ncls; ++ci) {
        const LaunchClass& cl = classes[ci];
        if (ext && class_of[ext_region] != (int)ci) continue;
        long long n;
        if (ext) {
            n = ext_n;
        } else if (c->n_regions == 1) {          // possibly a shard / a piece of one
            S.slot_begin = c->slot_begin + part * c->part_stride;
            S.slot_end = c->shard_parts > 1 ? S.slot_begin + c->part_slots : c->slot_end;
            n = S.slot_end - S.slot_begin;
        } else {
            S.slot_begin = 0;
            S.slot_end = (long long)cl.regions.size() * halfW;
            n = S.slot_end;
        }
        if (n <= 0) continue;
        hipStream_t st = c->stream;
        if (fork && ci > 0) {
            st = c->cls_stream[ci - 1];
            HIP_TRY(hipStreamWaitEvent(st, c->ev_fork, 0));
        }
        S.region_list = cl.list_d;
        // (a shard or a piece of a single-region ensemble takes the shape of the WHOLE ensemble)
        const int shape = class_shape(c, cl, ext ? n : halfW, (!ext && c->n_regions == 1) ? halfW : n, packable);
        unsigned grid = (unsigned)((n + shape_walkers_per_block(shape) - 1) / shape_walkers_per_block(shape));
        S.wpr = S.bpr = 0;
        S.n_cls_regions = (int)cl.regions.size();
        if (!ext && c->n_regions > 1 && (shape == SH_SMALL || shape == SH_SMALL2)) {
            const vamp::plan::PackedGrid pg = vamp::plan::plan_packed_grid((long long)cl.regions.size(), halfW,
                                                                           shape == SH_SMALL ? PackSmall::SUBS : PackSmall2::SUBS,
                                                                           shape == SH_SMALL ? PackSmall::WPB : PackSmall2::WPB);
            S.wpr = pg.wpr;
            S.bpr = pg.bpr;
            grid = (unsigned)pg.grid;
        } else if (!ext && c->n_regions > 1 && halfW % shape_walkers_per_block(shape) == 0) {
            S.bpr = (int)(halfW / shape_walkers_per_block(shape));        // one walker per wavefront or per workgroup
        }
        const dim3 threads(shape_threads(shape));
        if (ext) {
            if (c->f32)
                VAMP_FOR_MODE_PK(c->mode, shape, hipLaunchKernelGGL((k_half_step<true, DRAW_HOST, M, PK>), dim3(grid), threads, 0, st, S, px, step,
                                                          half, ext_region, ext_n, c->ext_act_d, c->ext_par_d, c->ext_z_d, c->ext_lu_d, nd));
            else
                VAMP_FOR_MODE_PK(c->mode, shape, hipLaunchKernelGGL((k_half_step<false, DRAW_HOST, M, PK>), dim3(grid), threads, 0, st, S, px, step,
                                                          half, ext_region, ext_n, c->ext_act_d, c->ext_par_d, c->ext_z_d, c->ext_lu_d, nd));
        } else if (!small_ensemble(halfW) && (shape == SH_SMALL || shape == SH_SMALL2 || (shape == SH_MID && VAMP_MID_PREDRAW))) {
            // four walkers per wavefront (or a walker per wavefront at ~1.7 wavefronts per SIMD, where
            // ~1000 scalar instructions of draws are exposed latency): draws in their own
            // one-thread-per-mover launch -- for ensembles that fill the chip; a small ensemble's half-step is bound by
            // its launches, and draws in the kernel are one launch less (the same draws: same keys, same functions)
            int rc = ensure_draw_buffers(c, draw_total);
            if (rc) return rc;
            int *d_ws = c->dr_ws + draw_off, *d_wc = c->dr_wc + draw_off;
            double *d_z = c->dr_z + draw_off, *d_lu = c->dr_lu + draw_off, *d_lz = c->dr_lz + draw_off;
            draw_off += n;
            hipLaunchKernelGGL(k_draws, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, step, half, n, d_ws, d_wc, d_z, d_lu, d_lz);
            HIP_TRY(hipGetLastError());
            if (c->f32)
                VAMP_FOR_MODE_PK(c->mode, shape, hipLaunchKernelGGL((k_half_step<true, DRAW_PRE, M, PK>), dim3(grid), threads, 0, st, S, px, step,
                                                          half, 0, 0ll, d_ws, d_wc, d_z, d_lu, d_lz));
            else
                VAMP_FOR_MODE_PK(c->mode, shape, hipLaunchKernelGGL((k_half_step<false, DRAW_PRE, M, PK>), dim3(grid), threads, 0, st, S, px, step,
                                                          half, 0, 0ll, d_ws, d_wc, d_z, d_lu, d_lz));
        } else {
            if (c->f32)
                VAMP_FOR_MODE_PK(c->mode, shape, hipLaunchKernelGGL((k_half_step<true, DRAW_INLINE, M, PK>), dim3(grid), threads, 0, st, S, px, step,
                                                          half, 0, 0ll, ni, ni, nd, nd, nd));
            else
                VAMP_FOR_MODE_PK(c->mode, shape, hipLaunchKernelGGL((k_half_step<false, DRAW_INLINE, M, PK>), dim3(grid), threads, 0, st, S, px, step,
                                                          half, 0, 0ll, ni, ni, nd, nd, nd));
        }
        HIP_TRY(hipGetLastError());
        if (fork && ci > 0) {
            HIP_TRY(hipEventRecord(c->ev_join[ci - 1], st));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join[ci - 1], 0));
        }
    }
    if (c->timing) HIP_TRY(hipEventRecord(e1, c->stream));
    return 0;
}

// ---- the device-resident step loop (k_run_resident): eligibility and launch ----------------------------------------
// compute wavefronts of a resident workgroup of shape PK for `movers` movers per half-step, or 0: does not fit
template <bool F32, int MODE, class PK>
int resident_waves(long long movers) {
    if constexpr (PK::SPLIT && PK::WPB > 1) return 0;
    else {
        int nw = (int)std::min<long long>(RES_MAX_WAVES, (movers + PK::SUBS - 1) / PK::SUBS);
        while (nw > 0 && ResLayout<F32, MODE, PK>::total(nw) > RES_MAX_LDS) --nw;
        return nw;
    }
}
template <bool F32, int MODE, class PK>
void launch_resident(dim3 grid, dim3 threads, int nw, hipStream_t st, const SamplerDev& S, const PixPtrs& px, unsigned step0, long long n_steps,
                     int thin, double* chain_dev, double* lchain_dev, long long total_theta, long long total_walkers) {
    if constexpr (!(PK::SPLIT && PK::WPB > 1)) {    // (resident_waves is 0 for the workgroup-per-walker shapes: never launched)
        using LY = ResLayout<F32, MODE, PK>;
        const size_t lds = LY::total(nw);
        if (lds > 48 * 1024)        // beyond the default dynamic allocation: the kernel is told once per process
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_run_resident<F32, MODE, PK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_MAX_LDS);
        hipLaunchKernelGGL((k_run_resident<F32, MODE, PK>), grid, threads, lds, st, S, px, step0, n_steps, thin, chain_dev, lchain_dev,
                           total_theta, total_walkers);
    }
}
int resident_waves_for(const vamp_ctx* c, int shape, long long movers) {
    int nw = 0;
    if (c->f32) VAMP_FOR_MODE_PK(c->mode, shape, nw = resident_waves<true, M, PK>(movers));
    else VAMP_FOR_MODE_PK(c->mode, shape, nw = resident_waves<false, M, PK>(movers));
    return nw;
}
// movers of one half-step launch of class `cl` on the launch-per-half-step path (what decides its shape)
long long class_movers(const vamp_ctx* c, const LaunchClass& cl) {
    return c->n_regions == 1 ? c->W / 2 : (long long)cl.regions.size() * (c->W / 2);
}
// every launch class of the context can run its regions' step loops inside one launch each
// Policy (opt_resident = 1; measured, profiles/r04_c_small_ensembles.txt).  A resident workgroup removes the dispatch
// gap of a launch per half-step (~5 us of ~16) but serialises its region on ONE compute unit, so it pays only where
// the launch path cannot use the chip anyway: at most one workgroup per compute unit (<= RES_MAX_REGIONS regions),
// every mover of a half-step in ONE round of the workgroup's wavefronts, and only the packed short-region classes
// (a blend or a region of 17+ lines takes a whole wavefront per walker: eight movers per round).  opt_resident = 2
// (tests, A/B) takes every context the kernel can run.
bool resident_eligible(const vamp_ctx* c) {
    if (!c->opt_resident || !c->sampler_ready) return false;
    if (c->shard_world != 1 || c->shard_parts != 1 || c->comm || c->send_d) return false;     // walker-sharded: the exchange is per half-step
    const long long halfW = c->W / 2;
    if (halfW > RES_MAX_MOVERS) return false;
    if (c->opt_resident == 1 && c->n_regions > RES_MAX_REGIONS) return false;
    for (const LaunchClass& cl : partition_for(c, halfW)) {
        const int shape = class_shape(c, cl, halfW, class_movers(c, cl), true);
        const int nw = resident_waves_for(c, shape, halfW);
        if (!vamp::plan::resident_class_ok(cl.kind, halfW, nw, (int)(shape_walkers_per_block(shape) / shape_waves(shape)), c->opt_resident == 1))
            return false;
    }
    return true;
}
// n_steps of every region, one launch per launch class (the classes on forked streams, as in launch_half)
int run_resident(vamp_ctx* c, long long n_steps, int thin, double* chain_dev, double* lchain_dev) {
    SamplerDev S;
    std::memset(&S, 0, sizeof(S));
    S.regions = c->regions_d;
    S.n_regions = c->n_regions;
    S.W = c->W;
    S.split_block = c->split_block;
    S.a = c->a;
    S.seed = c->seed;
    S.X = c->X_d;
    S.lnp = c->lnp_d;
    S.n_accept = c->nacc_d;
    const long long halfW = c->W / 2;
    const std::vector<LaunchClass>& classes = partition_for(c, halfW);
    const size_t ncls = classes.size();
    const bool fork = ncls > 1 && c->concurrent_classes;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        if (c->ev_used == c->ev.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            c->ev.emplace_back(a, b);
        }
        e0 = c->ev[c->ev_used].first;
        e1 = c->ev[c->ev_used].second;
        c->ev_used++;
        HIP_TRY(hipEventRecord(e0, c->stream));
    }
    if (fork) {
        while (c->cls_stream.size() < ncls - 1) {
            hipStream_t st;
            hipEvent_t ev;
            HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            c->cls_stream.push_back(st);
            c->ev_join.push_back(ev);
        }
        if (!c->ev_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
    }
    const PixPtrs px = c->pix();
    for (size_t ci = 0; ci < ncls; ++ci) {
        const LaunchClass& cl = classes[ci];
        hipStream_t st = c->stream;
        if (fork && ci > 0) {
            st = c->cls_stream[ci - 1];
            HIP_TRY(hipStreamWaitEvent(st, c->ev_fork, 0));
        }
        S.region_list = cl.list_d;
        const int shape = class_shape(c, cl, halfW, class_movers(c, cl), true);     // the shape launch_half runs this class in
        const int nw = resident_waves_for(c, shape, halfW);
        const dim3 grid((unsigned)cl.regions.size()), threads(64u * (unsigned)(nw + 1));
        if (c->f32)
            VAMP_FOR_MODE_PK(c->mode, shape, (launch_resident<true, M, PK>(grid, threads, nw, st, S, px, (unsigned)c->step, n_steps, thin, chain_dev,
                                                                          lchain_dev, c->total_theta, c->total_walkers)));
        else
            VAMP_FOR_MODE_PK(c->mode, shape, (launch_resident<false, M, PK>(grid, threads, nw, st, S, px, (unsigned)c->step, n_steps, thin, chain_dev,
                                                                           lchain_dev, c->total_theta, c->total_walkers)));
        HIP_TRY(hipGetLastError());
        if (fork && ci > 0) {
            HIP_TRY(hipEventRecord(c->ev_join[ci - 1], st));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join[ci - 1], 0));
        }
    }
    if (c->timing) HIP_TRY(hipEventRecord(e1, c->stream));
    c->step += n_steps;
    return 0;
}

// ---- RCCL, bound at run time ------------------------------------------------------------------
// Only walker-sharded multi-GPU runs need it, so the library does not link librccl: the few entry
// points are resolved with dlopen on first use (the copy already mapped into the process, e.g. by
// torch, wins; else /opt/rocm/lib).  Prototypes as in rccl/rccl.h (NCCL API).
struct RcclUniqueId { char internal[VAMP_COMM_ID_BYTES]; };
struct RcclApi {
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;        // optional: vamp_comm_info asks the communicator itself
    int (*CommUserRank)(void*, int*) = nullptr;
    bool ok = false;
};
constexpr int RCCL_FLOAT64 = 8;       // ncclFloat64 / ncclDouble

int rccl_api(RcclApi** out) {
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        void* h = nullptr;
        // VAMP_RCCL_LIB: a library with the five NCCL entry points used here, tried first.  The GPU tests
        // point it at a stand-in that moves the bytes through host shared memory (tests/host/rccl_fake.cpp),
        // which lets two ranks share the one GPU of a test box -- RCCL itself refuses that -- and runs the
        // whole in-library exchange path at world > 1.
        if (const char* over = getenv("VAMP_RCCL_LIB")) h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            // RCCL must sit on the SAME HIP / HSA runtime as this library: a process that also imports torch holds
            // torch's private copies (torch/lib/libamdhip64.so, libhsa-runtime64.so, librccl.so) beside the system
            // ones, and an RCCL bound to a runtime nobody initialised fails in ncclCommInitRank ("no ROCm-capable
            // device is detected") -- which copy a bare dlopen("librccl.so") returns depends on the import order.
            // So: first the librccl next to the libamdhip64 this library is linked against, then the usual names.
            std::vector<std::string> names;
            Dl_info info;
            if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
                std::string dir(info.dli_fname);
                const size_t cut = dir.rfind('/');
                if (cut != std::string::npos) {
                    dir.resize(cut);
                    names.push_back(dir + "/librccl.so.1");
                    names.push_back(dir + "/librccl.so");
                }
            }
            for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) names.push_back(n);
            for (const std::string& name : names) {
                h = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL);
                if (h) break;
            }
        }
        if (h) {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
            api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
            api.CommCount = (decltype(api.CommCount))dlsym(h, "ncclCommCount");
            api.CommUserRank = (decltype(api.CommUserRank))dlsym(h, "ncclCommUserRank");
            api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString;
        }
    }
    if (!api.ok) return fail(VAMP_ERR_COMM, "RCCL is not available (librccl.so could not be loaded)");
    *out = &api;
    return VAMP_OK;
}
#define RCCL_TRY(api, expr)                                                                   \
    do {                                                                                      \
        int r_ = (expr);                                                                      \
        if (r_ != 0) return fail(VAMP_ERR_COMM, std::string(#expr) + ": " + (api)->GetErrorString(r_)); \
    } while (0)

// ---- roctx ranges (SURVEY section 5 "Tracing"), bound at run time like RCCL ----------------------
// `rocprofv3 --marker-trace` shows the sampler loop, every half-step (with its exchange), the MAP searches and
// the batched evaluations as named ranges around the kernels.  Without a tool attached a range costs two calls
// into an idle library; VAMP_ROCTX=0 switches them off altogether.
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
RoctxApi* roctx_api() {
    static RoctxApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* sw = getenv("VAMP_ROCTX");
        if (sw && std::atoi(sw) == 0) return nullptr;
        std::vector<std::string> names;
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t cut = dir.rfind('/');
            if (cut != std::string::npos) {
                dir.resize(cut);
                names.push_back(dir + "/librocprofiler-sdk-roctx.so.1");
                names.push_back(dir + "/libroctx64.so.4");
            }
        }
        for (const char* n : {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1",
                              "/opt/rocm/lib/libroctx64.so.4"})
            names.push_back(n);
        for (const std::string& name : names) {
            void* h = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!h) continue;
            api.push = (decltype(api.push))dlsym(h, "roctxRangePushA");
            api.pop = (decltype(api.pop))dlsym(h, "roctxRangePop");
            if (api.push && api.pop) break;
            api.push = nullptr;
            api.pop = nullptr;
        }
    }
    return (api.push && api.pop) ? &api : nullptr;
}
struct RoctxRange {
    RoctxApi* a;
    explicit RoctxRange(const char* name) : a(roctx_api()) { if (a) a->push(name); }
    ~RoctxRange() { if (a) a->pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

// rows of part `part` gathered in c->recv_d -> walker rows (on `st`)
int launch_scatter(vamp_ctx* c, int part, hipStream_t st) {
    SamplerDev S;
    std::memset(&S, 0, sizeof(S));
    S.regions = c->regions_d;
    S.n_regions = c->n_regions;
    S.W = c->W;
    S.split_block = c->split_block;
    S.a = c->a;
    S.seed = c->seed;
    S.X = c->X_d;
    S.lnp = c->lnp_d;
    S.n_accept = c->nacc_d;
    const long long n_rows = (long long)c->shard_world * c->part_slots;
    const long long own_lo = (long long)c->shard_rank * c->part_slots;
    const double* recv = c->recv_d + (long long)part * n_rows * (c->regions_h[0].D + 1);
    const unsigned grid = (unsigned)((n_rows * 16 + 255) / 256);
    hipLaunchKernelGGL(k_scatter_rows, dim3(grid), dim3(256), 0, st, S, recv, c->part_step[part], c->part_half[part],
                       (long long)part * c->part_stride, n_rows, own_lo, own_lo + c->part_slots);
    HIP_TRY(hipGetLastError());
    return 0;
}

// after launch_half(part): all-gather the part's movers over RCCL and scatter them.  With several
// parts the exchange runs on the communication stream, so that it overlaps the next part's kernel.
int exchange_part(vamp_ctx* c, int part) {
    RcclApi* api = nullptr;
    int rc = rccl_api(&api);
    if (rc) return rc;
    const long long row = c->regions_h[0].D + 1;
    const size_t count = (size_t)(c->part_slots * row);
    const double* send = c->send_d + (long long)part * c->part_slots * row;
    double* recv = c->recv_d + (long long)part * c->shard_world * c->part_slots * row;
    const bool overlap = c->shard_parts > 1;
    hipStream_t st = overlap ? c->comm_stream : c->stream;
    if (overlap) {
        rc = ensure_part_events(c, c->shard_parts);     // (whichever of set_shard_parts / comm_init_rank came first)
        if (rc) return rc;
        HIP_TRY(hipEventRecord(c->ev_kernel[part], c->stream));
        HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_kernel[part], 0));
    }
    hipEvent_t x1 = nullptr;
    if (c->timing) {
        if (c->xev_used == c->xev.size()) {
            if (c->xev.size() >= 4096) {
                rc = flush_exchange_timing(c);
                if (rc) return rc;
            } else {
                hipEvent_t a, b;
                HIP_TRY(hipEventCreate(&a));
                HIP_TRY(hipEventCreate(&b));
                c->xev.emplace_back(a, b);
            }
        }
        HIP_TRY(hipEventRecord(c->xev[c->xev_used].first, st));
        x1 = c->xev[c->xev_used].second;
        c->xev_used++;
    }
    {
        const int r_ = api->AllGather(send, recv, count, RCCL_FLOAT64, c->comm, st);
        if (r_ == 0) rc = launch_scatter(c, part, st);
        if (r_ != 0 || rc) {
            if (x1) c->xev_used--;              // its closing event will never be recorded
            if (r_ != 0) return fail(VAMP_ERR_COMM, std::string("ncclAllGather: ") + api->GetErrorString(r_));
            return rc;
        }
    }
    if (x1) HIP_TRY(hipEventRecord(x1, st));
    if (overlap) HIP_TRY(hipEventRecord(c->ev_scatter[part], c->comm_stream));
    return 0;
}

// one half-step of this device's whole share; with a communicator, followed by the exchange
int half_step_all(vamp_ctx* c, int half) {
    RoctxRange range(half ? "vamp half-step 1 (blue moves)" : "vamp half-step 0 (red moves)");
    for (int p = 0; p < c->shard_parts; ++p) {
        int rc = launch_half(c, half, false, 0, 0, p);
        if (rc) return rc;
        if (c->comm && c->send_d) {
            rc = exchange_part(c, p);
            if (rc) return rc;
        }
    }
    if (c->comm && c->send_d && c->shard_parts > 1)
        for (int p = 0; p < c->shard_parts; ++p) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_scatter[p], 0));
    return 0;
}

int free_comm(vamp_ctx* c) {
    if (c->comm) {
        RcclApi* api = nullptr;
        if (rccl_api(&api) == 0) (void)api->CommDestroy(c->comm);
        c->comm = nullptr;
    }
    for (hipEvent_t e : c->ev_kernel) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_scatter) (void)hipEventDestroy(e);
    c->ev_kernel.clear();
    c->ev_scatter.clear();
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    c->comm_stream = nullptr;
    c->comm_rank = 0;
    c->comm_world = 1;
    return 0;
}

}  // namespace

extern "C" {

int vamp_version(void) { return VAMP_ABI_VERSION; }

const char* vamp_last_error(void) { return g_err.c_str(); }

int vamp_device_count(int* n) {
    if (!n) return fail(VAMP_ERR_ARG, "vamp_device_count: n is NULL");
    HIP_TRY(hipGetDeviceCount(n));
    return VAMP_OK;
}

int vamp_ctx_create(vamp_ctx** out, int device, int dtype, int wofz_kind) {
    if (!out) return fail(VAMP_ERR_ARG, "vamp_ctx_create: out is NULL");
    if (!((dtype == VAMP_F64 && wofz_kind == VAMP_WOFZ_ACCURATE) || (dtype == VAMP_F32 && wofz_kind == VAMP_WOFZ_HUMLICEK_W4)))
        return fail(VAMP_ERR_ARG, "vamp_ctx_create: supported pairs are (F64, ACCURATE) and (F32, HUMLICEK_W4)");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(VAMP_ERR_ARG, "vamp_ctx_create: no such device");
    HIP_TRY(hipSetDevice(device));
    vamp_ctx* c = new (std::nothrow) vamp_ctx();
    if (!c) return fail(VAMP_ERR_NOMEM, "vamp_ctx_create: host allocation failed");
    c->device = device;
    c->dtype = dtype;
    c->wofz_kind = wofz_kind;
    c->f32 = (dtype == VAMP_F32);
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(VAMP_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    c->stream = c->own_stream;
    if (const char* e = getenv("VAMP_CLASS_STREAMS")) c->concurrent_classes = std::atoi(e) != 0;
    if (const char* e = getenv("VAMP_RESIDENT")) c->opt_resident = std::max(0, std::min(2, std::atoi(e)));          // (A/B knobs of tools/ and profiles/;
    if (const char* e = getenv("VAMP_MAP_DEVICE")) c->opt_map_device = std::atoi(e) != 0;      //  vamp_ctx_set_option overrides them)
    *out = c;
    return VAMP_OK;
}

int vamp_ctx_destroy(vamp_ctx* c) {
    if (!c) return VAMP_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    free_comm(c);
    free_sampler(c);
    free_regions(c);
    for (void* p : {(void*)c->ext_act_d, (void*)c->ext_par_d, (void*)c->ext_z_d, (void*)c->ext_lu_d, (void*)c->sc_th,
                    (void*)c->sc_lp, (void*)c->sc_chi, (void*)c->dr_ws, (void*)c->dr_wc, (void*)c->dr_z, (void*)c->dr_lu,
                    (void*)c->dr_lz, (void*)c->map_th_d, (void*)c->map_best_d, (void*)c->map_sim_d, (void*)c->map_act_d,
                    (void*)c->map_it_d})
        if (p) (void)hipFree(p);
    if (c->pin_th) (void)hipHostFree(c->pin_th);
    if (c->pin_out) (void)hipHostFree(c->pin_out);
    for (hipStream_t st : c->cls_stream) (void)hipStreamDestroy(st);
    for (hipEvent_t e : c->ev_join) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (auto& p : c->ev) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (auto& p : c->xev) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return VAMP_OK;
}

int vamp_ctx_set_stream(vamp_ctx* c, void* hip_stream) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_ctx_set_stream: ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return VAMP_OK;
}

int vamp_ctx_set_stream_default(vamp_ctx* c) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_ctx_set_stream_default: ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = nullptr;       // HIP's legacy default stream: what callers that never create a stream run on
    return VAMP_OK;
}

int vamp_ctx_set_packing(vamp_ctx* c, int lanes_per_walker) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_ctx_set_packing: ctx is NULL");
    if (lanes_per_walker != 0 && lanes_per_walker != 16 && lanes_per_walker != 64 && lanes_per_walker != 65 && lanes_per_walker != 256)
        return fail(VAMP_ERR_ARG, "vamp_ctx_set_packing: lanes_per_walker must be 0 (auto), 16, 64, 65 (64 + per-walker tables) or 256");
    c->packing = lanes_per_walker;
    return VAMP_OK;
}

int vamp_ctx_set_option(vamp_ctx* c, const char* name, int64_t value) {
    if (!c || !name) return fail(VAMP_ERR_ARG, "vamp_ctx_set_option: NULL argument");
    const std::string key(name);
    if (key == "map_device") c->opt_map_device = value != 0;
    else if (key == "resident") c->opt_resident = value < 0 ? 0 : value > 2 ? 2 : (int)value;
    else if (key == "class_streams") c->concurrent_classes = value != 0;
    else return fail(VAMP_ERR_ARG, "vamp_ctx_set_option: unknown option '" + key + "' (map_device, resident, class_streams)");
    return VAMP_OK;
}

int vamp_ctx_synchronize(vamp_ctx* c) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_ctx_synchronize: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
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
    if (n_regions > 65535) return fail(VAMP_ERR_ARG, "vamp_set_regions: at most 65535 regions per context (one grid row per region)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_sampler(c);
    free_regions(c);
    const int q = (mode == VAMP_VOIGT4) ? 4 : 3;
    std::vector<RegionDev> R(n_regions);
    for (int r = 0; r < n_regions; ++r) {
        const long long P = pix_off[r + 1] - pix_off[r];
        if (P < 2 || P > 0x7fffffff) return fail(VAMP_ERR_ARG, "vamp_set_regions: a region needs >= 2 pixels");
        if (n_comp[r] < 1 || n_comp[r] > KMAX_ALL) return fail(VAMP_ERR_ARG, "vamp_set_regions: n_comp out of range (1..32)");
        RegionDev d;
        std::memset(&d, 0, sizeof(d));
        d.pix_off = pix_off[r];
        d.P = (int)P;
        d.K = n_comp[r];
        d.mode = mode;
        d.q = q;
        d.sample_sd = sample_sd ? 1 : 0;
        d.rng_id = r;
        d.D = q * d.K + d.sample_sd;
        d.d_before = r ? R[r - 1].d_before + R[r - 1].D : 0;
        d.tau_off = r ? R[r - 1].tau_off + (long long)R[r - 1].K * R[r - 1].P : 0;
        d.sim_off = r ? R[r - 1].sim_off + (long long)(R[r - 1].D + 1) * R[r - 1].D : 0;
        const double* xr = x + pix_off[r];
        if (bounds) {
            d.c_lo = bounds[4 * r + 0];
            d.c_hi = bounds[4 * r + 1];
            d.w_max = (mode == VAMP_GAUSS3) ? bounds[4 * r + 2] : bounds[4 * r + 3];
        } else {
            d.c_lo = std::min(xr[0], xr[P - 1]);             // vpfits.py:250 (the reference's grid ascends)
            d.c_hi = std::max(xr[0], xr[P - 1]);
            const double sigma_max = (d.c_hi - d.c_lo) / 2.0;                         // vpfits.py:320
            d.w_max = (mode == VAMP_GAUSS3) ? sigma_max : sigma_max * 2 * std::sqrt(2 * std::log(2.0));   // :326
        }
        if (!(d.c_hi > d.c_lo) || !(d.w_max > 0)) return fail(VAMP_ERR_ARG, "vamp_set_regions: empty prior range");
        d.lp_c = -std::log(d.c_hi - d.c_lo);
        d.lp_w = -std::log(d.w_max);
        if (mode == VAMP_NBZ3) {
            d.l_fixed = nbz[4 * r + 0];
            d.line = nbz[4 * r + 1];
            d.x_origin = nbz[4 * r + 2];
            d.x_scale = nbz[4 * r + 3];
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
        // the tile code takes a tile's first and last pixel as its extent: the grid of a region must be
        // strictly monotonic (either direction; the reference sorts to ascending frequency,
        // vpspectrum.py:274-277) and finite
        double dxmax = 0.0;
        const bool up = xr[1] > xr[0];
        for (long long i = 1; i < P; ++i) {
            const double dx = xr[i] - xr[i - 1];
            if (!std::isfinite(dx) || dx == 0.0 || (dx > 0.0) != up)
                return fail(VAMP_ERR_ARG, "vamp_set_regions: x must be finite and strictly monotonic within a region");
            dxmax = std::max(dxmax, std::fabs(dx));
        }
        d.tile_span = 64.0 * TPIX * dxmax;
        R[r] = d;
    }
    const long long N = pix_off[n_regions];
    std::vector<double> wt(N);
    for (long long i = 0; i < N; ++i) wt[i] = sample_sd ? 1.0 : 1.0 / noise[i];
    HIP_TRY(hipMalloc(&c->x_d, N * sizeof(double)));
    HIP_TRY(hipMalloc(&c->f_d, N * sizeof(double)));
    HIP_TRY(hipMalloc(&c->wt_d, N * sizeof(double)));
    HIP_TRY(hipMemcpy(c->x_d, x, N * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->f_d, flux, N * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->wt_d, wt.data(), N * sizeof(double), hipMemcpyHostToDevice));
    if (c->f32) {
        std::vector<float> t(N);
        HIP_TRY(hipMalloc(&c->xf_d, N * sizeof(float)));
        HIP_TRY(hipMalloc(&c->ff_d, N * sizeof(float)));
        HIP_TRY(hipMalloc(&c->wtf_d, N * sizeof(float)));
        for (long long i = 0; i < N; ++i) t[i] = (float)x[i];
        HIP_TRY(hipMemcpy(c->xf_d, t.data(), N * sizeof(float), hipMemcpyHostToDevice));
        for (long long i = 0; i < N; ++i) t[i] = (float)flux[i];
        HIP_TRY(hipMemcpy(c->ff_d, t.data(), N * sizeof(float), hipMemcpyHostToDevice));
        for (long long i = 0; i < N; ++i) t[i] = (float)wt[i];
        HIP_TRY(hipMemcpy(c->wtf_d, t.data(), N * sizeof(float), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&c->regions_d, n_regions * sizeof(RegionDev)));
    HIP_TRY(hipMemcpy(c->regions_d, R.data(), n_regions * sizeof(RegionDev), hipMemcpyHostToDevice));
    {
        // launch classes (csrc/host_plan.hpp, shared with the host build of this ABI).  Forced packings: one class.
        // Automatic: contexts that look like a real spectrum (<= 8 lines in every region of <= 16, mean region
        // <= 128 px) split into the blends worth a wavefront and a set of Taylor tables per walker (>= 3 lines over
        // 96 .. 512 px: table building costs ~2 near-axis evaluations per line and interval, repaid from ~30 px per
        // line on), the one- and two-line regions (eight walkers per wavefront) and the rest (four); regions of more
        // than 16 lines form their own class (plain shape PackXL); everything else is one wide class.
        vamp::plan::Limits lim;
        static_assert(KMAX == 16 && KMAX_ALL == 32 && PackSmall::KCAP == 8 && PackSmall2::KCAP == 2 && VAMP_MID_MIN_K == 3 &&
                      VAMP_MID_MIN_P == 96 && BLEND_MAX_PIXELS == 512 && 64 * TPIX == 256, "vamp::plan::Limits describes these shapes");
        std::vector<vamp::plan::RegionShape> shp(n_regions);
        for (int r = 0; r < n_regions; ++r) shp[r] = vamp::plan::RegionShape{R[r].P, R[r].K};
        vamp::plan::ClassPlan cp;
        const std::string err = vamp::plan::plan_classes(shp, c->packing, mode == VAMP_GAUSS3, c->f32, VAMP_F32_TABLES != 0, cp, lim);
        if (!err.empty()) return fail(VAMP_ERR_ARG, "vamp_set_regions: " + err);
        c->full_tiles = cp.full_tiles;
        c->min_tiles = cp.min_tiles;
        c->class_of = cp.class_of;
        vamp::plan::ClassPlan cs;          // the partition of small ensembles: short-region classes merged
        (void)vamp::plan::plan_classes(shp, c->packing, mode == VAMP_GAUSS3, c->f32, VAMP_F32_TABLES != 0, cs, lim, true);
        c->class_of_small = cs.class_of;
        for (int which = 0; which < 2; ++which) {
            const vamp::plan::ClassPlan& pl = which ? cs : cp;
            std::vector<LaunchClass>& part = which ? c->classes_small : c->classes;
            for (size_t k = 0; k < pl.kind.size(); ++k) {
                LaunchClass cl;
                cl.kind = pl.kind[k];
                cl.regions = pl.regions[k];
                part.push_back(cl);
            }
#if VAMP_LPT
            // small ensembles: a launch is a few rounds of wavefronts, each a serial chain whose length grows with the region's
            // lines x pixels; the longest chains go FIRST, so that none starts in the last round (regions are independent and the
            // draws are keyed by region and walker: the order changes no result)
            bool reordered = false;
            if (which == 1 || VAMP_LPT == 2)
                for (LaunchClass& cl : part) {
                    std::vector<int> before = cl.regions;
                    std::stable_sort(cl.regions.begin(), cl.regions.end(), [&](int a, int b) {
                        return (long long)R[a].K * R[a].P > (long long)R[b].K * R[b].P;
                    });
                    reordered = reordered || before != cl.regions;
                }
            if (part.size() > 1 || reordered)
#else
            if (part.size() > 1)
#endif
                for (LaunchClass& cl : part) {
                    HIP_TRY(hipMalloc(&cl.list_d, cl.regions.size() * sizeof(int)));
                    HIP_TRY(hipMemcpy(cl.list_d, cl.regions.data(), cl.regions.size() * sizeof(int), hipMemcpyHostToDevice));
                }
        }
    }
    c->regions_h = R;
    c->mode = mode;
    c->n_regions = n_regions;
    c->n_pix = N;
    return VAMP_OK;
}

int vamp_set_region_ids(vamp_ctx* c, const int32_t* ids) {
    if (!c || !ids) return fail(VAMP_ERR_ARG, "vamp_set_region_ids: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_set_region_ids: call vamp_set_regions first");
    for (int r = 0; r < c->n_regions; ++r)
        if (ids[r] < 0) return fail(VAMP_ERR_ARG, "vamp_set_region_ids: ids must be >= 0");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int r = 0; r < c->n_regions; ++r) c->regions_h[r].rng_id = ids[r];
    HIP_TRY(hipMemcpy(c->regions_d, c->regions_h.data(), c->n_regions * sizeof(RegionDev), hipMemcpyHostToDevice));
    return VAMP_OK;
}

int vamp_region_class(vamp_ctx* c, int region, int* kind, int* n_classes) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_region_class: ctx is NULL");
    if (region < 0 || region >= c->n_regions) return fail(VAMP_ERR_ARG, "vamp_region_class: no such region");
    if (kind) *kind = c->classes[c->class_of[region]].kind;
    if (n_classes) *n_classes = (int)c->classes.size();
    return VAMP_OK;
}

int vamp_region_ndim(vamp_ctx* c, int region, int* ndim) {
    if (!c || !ndim) return fail(VAMP_ERR_ARG, "vamp_region_ndim: NULL argument");
    if (region < 0 || region >= c->n_regions) return fail(VAMP_ERR_ARG, "vamp_region_ndim: no such region");
    *ndim = c->regions_h[region].D;
    return VAMP_OK;
}

namespace {
// k_lnprob on device arrays: region >= 0: theta[W, D] of that region -> lnprob[W]; region < 0: every
// region, theta = the regions' [W, D_r] blocks one after the other -> lnprob[n_regions, W]; one launch
// per launch class (blockIdx.y walks the class's region list)
int launch_lnprob(vamp_ctx* c, int region, long long W, const double* th_d, double* lp_d, double* ch_d) {
    const bool all = region < 0;
    const bool packable = true;      // one region per block row: every wave lies inside one region
    // W points per region are W/2 movers of a W-walker ensemble: the sampler's own initial log-posteriors, and a caller's
    // walker checks, run the classes and shapes the ensemble will be stepped in
    const long long per = (W + 1) / 2;
    const std::vector<LaunchClass>& classes = partition_for(c, per);
    const std::vector<int>& class_of = class_of_for(c, per);
    for (size_t ci = 0; ci < classes.size(); ++ci) {
        const LaunchClass& cl = classes[ci];
        if (!all && class_of[region] != (int)ci) continue;
        const int shape = class_shape(c, cl, per, all ? per * (long long)cl.regions.size() : per, packable);
        const long long per_block = shape_walkers_per_block(shape);
        const dim3 grid((unsigned)((W + per_block - 1) / per_block), all ? (unsigned)cl.regions.size() : 1u);
        const dim3 threads(shape_threads(shape));
        if (c->f32)
            VAMP_FOR_MODE_PK(c->mode, shape, hipLaunchKernelGGL((k_lnprob<true, M, PK>), grid, threads, 0, c->stream, c->regions_d, region,
                                                      c->pix(), W, th_d, lp_d, ch_d, (const int*)cl.list_d));
        else
            VAMP_FOR_MODE_PK(c->mode, shape, hipLaunchKernelGGL((k_lnprob<false, M, PK>), grid, threads, 0, c->stream, c->regions_d, region,
                                                      c->pix(), W, th_d, lp_d, ch_d, (const int*)cl.list_d));
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

constexpr size_t PINNED_EVAL_BYTES = 1u << 20;     // parameter blocks up to 1 MB are read by the kernel from pinned host memory
// region >= 0: W parameter vectors of that region; region < 0: W vectors of EVERY region
int lnprob_impl(vamp_ctx* c, int region, int64_t W, const double* theta, double* lnprob, double* chi2) {
    HIP_TRY(hipSetDevice(c->device));
    const bool all = region < 0;
    const long long dsum = all ? c->regions_h.back().d_before + c->regions_h.back().D : c->regions_h[region].D;
    const size_t nth = (size_t)W * dsum;
    const size_t nout = (size_t)W * (all ? c->n_regions : 1);
    if (nth * sizeof(double) <= PINNED_EVAL_BYTES) {
        if (c->pin_th_cap < nth) {
            if (c->pin_th) (void)hipHostFree(c->pin_th);
            c->pin_th = nullptr; c->pin_th_cap = 0;
            const size_t cap = std::max(nth, (size_t)4096);
            HIP_TRY(hipHostMalloc(&c->pin_th, cap * sizeof(double), hipHostMallocDefault));
            c->pin_th_cap = cap;
        }
        if (c->pin_out_cap < 2 * nout) {
            if (c->pin_out) (void)hipHostFree(c->pin_out);
            c->pin_out = nullptr; c->pin_out_cap = 0;
            const size_t cap = std::max(2 * nout, (size_t)1024);
            HIP_TRY(hipHostMalloc(&c->pin_out, cap * sizeof(double), hipHostMallocDefault));
            c->pin_out_cap = cap;
        }
        std::memcpy(c->pin_th, theta, nth * sizeof(double));
        double* lp_h = c->pin_out;
        double* ch_h = chi2 ? c->pin_out + nout : nullptr;
        int rc = launch_lnprob(c, region, W, c->pin_th, lp_h, ch_h);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::memcpy(lnprob, lp_h, nout * sizeof(double));
        if (chi2) std::memcpy(chi2, ch_h, nout * sizeof(double));
        return VAMP_OK;
    }
    if (c->sc_th_cap < nth) {
        if (c->sc_th) (void)hipFree(c->sc_th);
        c->sc_th = nullptr; c->sc_th_cap = 0;
        HIP_TRY(hipMalloc(&c->sc_th, nth * sizeof(double)));
        c->sc_th_cap = nth;
    }
    if (c->sc_w_cap < nout) {
        if (c->sc_lp) (void)hipFree(c->sc_lp);
        if (c->sc_chi) (void)hipFree(c->sc_chi);
        c->sc_lp = c->sc_chi = nullptr; c->sc_w_cap = 0;
        HIP_TRY(hipMalloc(&c->sc_lp, nout * sizeof(double)));
        HIP_TRY(hipMalloc(&c->sc_chi, nout * sizeof(double)));
        c->sc_w_cap = nout;
    }
    double *th_d = c->sc_th, *lp_d = c->sc_lp, *ch_d = chi2 ? c->sc_chi : nullptr;
    HIP_TRY(hipMemcpyAsync(th_d, theta, nth * sizeof(double), hipMemcpyHostToDevice, c->stream));
    // (the shape must not depend on `all`: a point has the same lnprob bits through either entry)
    int rc = launch_lnprob(c, region, W, th_d, lp_d, ch_d);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(lnprob, lp_d, nout * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (chi2) HIP_TRY(hipMemcpyAsync(chi2, ch_d, nout * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VAMP_OK;
}
}  // namespace

int vamp_lnprob(vamp_ctx* c, int region, int64_t W, const double* theta, double* lnprob, double* chi2) {
    if (!c || !theta || !lnprob) return fail(VAMP_ERR_ARG, "vamp_lnprob: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_lnprob: call vamp_set_regions first");
    if (region < 0 || region >= c->n_regions) return fail(VAMP_ERR_ARG, "vamp_lnprob: no such region");
    if (W <= 0) return fail(VAMP_ERR_ARG, "vamp_lnprob: W must be positive");
    return lnprob_impl(c, region, W, theta, lnprob, chi2);
}

int vamp_lnprob_all(vamp_ctx* c, int64_t W, const double* theta, double* lnprob, double* chi2) {
    if (!c || !theta || !lnprob) return fail(VAMP_ERR_ARG, "vamp_lnprob_all: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_lnprob_all: call vamp_set_regions first");
    if (W <= 0) return fail(VAMP_ERR_ARG, "vamp_lnprob_all: W must be positive");
    if (c->n_regions > 65535) return fail(VAMP_ERR_ARG, "vamp_lnprob_all: at most 65535 regions per launch");
    RoctxRange range("vamp_lnprob_all");
    return lnprob_impl(c, -1, W, theta, lnprob, chi2);
}

int vamp_map_all(vamp_ctx* c, const double* theta0, const uint8_t* active, int64_t maxiter, int64_t maxfun, double xtol,
                 double ftol, double* theta_best, double* lnprob_best, double* chi2_best, int64_t* iterations) {
    if (!c || !theta0 || !theta_best || !lnprob_best) return fail(VAMP_ERR_ARG, "vamp_map_all: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_map_all: call vamp_set_regions first");
    if (c->n_regions > 65535) return fail(VAMP_ERR_ARG, "vamp_map_all: at most 65535 regions per launch");
    if (maxiter < 0 || maxfun < 0 || !(xtol >= 0.0) || !(ftol >= 0.0)) return fail(VAMP_ERR_ARG, "vamp_map_all: bad limits");
    RoctxRange range("vamp_map_all");
    if (c->opt_map_device) {
        // every region's whole search in ONE launch per launch class (k_map_search), no per-iteration synchronisation
        HIP_TRY(hipSetDevice(c->device));
        const RegionDev& last = c->regions_h.back();
        const size_t nth = (size_t)(last.d_before + last.D), nsim = (size_t)(last.sim_off + (long long)(last.D + 1) * last.D);
        const size_t nr = (size_t)c->n_regions;
        if (c->map_th_cap < nth) {
            for (void* p : {(void*)c->map_th_d, (void*)c->map_best_d}) if (p) (void)hipFree(p);
            c->map_th_d = c->map_best_d = nullptr; c->map_th_cap = 0;
            HIP_TRY(hipMalloc(&c->map_th_d, nth * sizeof(double)));
            HIP_TRY(hipMalloc(&c->map_best_d, nth * sizeof(double)));
            c->map_th_cap = nth;
        }
        if (c->map_sim_cap < nsim) {
            if (c->map_sim_d) (void)hipFree(c->map_sim_d);
            c->map_sim_d = nullptr; c->map_sim_cap = 0;
            HIP_TRY(hipMalloc(&c->map_sim_d, nsim * sizeof(double)));
            c->map_sim_cap = nsim;
        }
        if (c->map_r_cap < nr) {
            if (c->map_act_d) (void)hipFree(c->map_act_d);
            if (c->map_it_d) (void)hipFree(c->map_it_d);
            c->map_act_d = nullptr; c->map_it_d = nullptr; c->map_r_cap = 0;
            HIP_TRY(hipMalloc(&c->map_act_d, nr));
            HIP_TRY(hipMalloc(&c->map_it_d, nr * sizeof(long long)));
            c->map_r_cap = nr;
        }
        HIP_TRY(hipMemcpyAsync(c->map_th_d, theta0, nth * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (active) HIP_TRY(hipMemcpyAsync(c->map_act_d, active, nr, hipMemcpyHostToDevice, c->stream));
        const unsigned char* act_d = active ? c->map_act_d : nullptr;
        const std::vector<LaunchClass>& classes = partition_for(c, 1);
        for (size_t ci = 0; ci < classes.size(); ++ci) {
            const LaunchClass& cl = classes[ci];
            // the shape vamp_lnprob runs a single point of this class in: the objective has the same bits
            const int shape = class_shape(c, cl, 1, 1, true);
            const dim3 grid((unsigned)cl.regions.size()), threads(shape_threads(shape));
            if (c->f32)
                VAMP_FOR_MODE_PK(c->mode, shape, hipLaunchKernelGGL((k_map_search<true, M, PK>), grid, threads, 0, c->stream, c->regions_d, c->pix(),
                                                          (const int*)cl.list_d, c->map_th_d, act_d, (long long)maxiter, (long long)maxfun, xtol, ftol,
                                                          c->map_sim_d, c->map_best_d, c->map_it_d));
            else
                VAMP_FOR_MODE_PK(c->mode, shape, hipLaunchKernelGGL((k_map_search<false, M, PK>), grid, threads, 0, c->stream, c->regions_d, c->pix(),
                                                          (const int*)cl.list_d, c->map_th_d, act_d, (long long)maxiter, (long long)maxfun, xtol, ftol,
                                                          c->map_sim_d, c->map_best_d, c->map_it_d));
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemcpyAsync(theta_best, c->map_best_d, nth * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        std::vector<long long> its(iterations ? nr : 0);
        if (iterations) HIP_TRY(hipMemcpyAsync(its.data(), c->map_it_d, nr * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (iterations) for (size_t r = 0; r < nr; ++r) iterations[r] = its[r];
        return lnprob_impl(c, -1, 1, theta_best, lnprob_best, chi2_best);
    }
    // the host-driven form (vamp_ctx_set_option "map_device" = 0): csrc/map_search.hpp (scipy fmin's rules; shared with
    // the host build of this ABI), one launch + one synchronisation per iteration of all regions
    std::vector<int> dims(c->n_regions);
    std::vector<long long> offs(c->n_regions);
    for (int r = 0; r < c->n_regions; ++r) {
        dims[r] = c->regions_h[r].D;
        offs[r] = c->regions_h[r].d_before;
    }
    int rc = vamp::nelder_mead_all(c->n_regions, dims.data(), offs.data(), theta0, active, maxiter, maxfun, xtol, ftol, theta_best,
                                   iterations, [&](int W, const double* th, double* lp) { return lnprob_impl(c, -1, W, th, lp, nullptr); });
    if (rc) return rc;
    return lnprob_impl(c, -1, 1, theta_best, lnprob_best, chi2_best);
}

int vamp_model(vamp_ctx* c, int region, const double* theta1, double* tau_comp, double* flux_model) {
    if (!c || !theta1) return fail(VAMP_ERR_ARG, "vamp_model: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_model: call vamp_set_regions first");
    if (region < 0 || region >= c->n_regions) return fail(VAMP_ERR_ARG, "vamp_model: no such region");
    HIP_TRY(hipSetDevice(c->device));
    const RegionDev& R = c->regions_h[region];
    DevBuf th_b, tau_b, fl_b;
    HIP_TRY(hipMalloc(&th_b.p, R.D * sizeof(double)));
    if (tau_comp) HIP_TRY(hipMalloc(&tau_b.p, (size_t)R.K * R.P * sizeof(double)));
    if (flux_model) HIP_TRY(hipMalloc(&fl_b.p, (size_t)R.P * sizeof(double)));
    double *th_d = th_b.as<double>(), *tau_d = tau_b.as<double>(), *fl_d = fl_b.as<double>();
    HIP_TRY(hipMemcpyAsync(th_d, theta1, R.D * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const unsigned grid = (unsigned)((R.P + BLOCK - 1) / BLOCK);
    VAMP_FOR_MODE(c->mode, hipLaunchKernelGGL((k_model<M>), dim3(grid), dim3(BLOCK), 0, c->stream, c->regions_d, region, c->pix(),
                                              th_d, tau_d, fl_d));
    HIP_TRY(hipGetLastError());
    if (tau_comp) HIP_TRY(hipMemcpyAsync(tau_comp, tau_d, (size_t)R.K * R.P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (flux_model) HIP_TRY(hipMemcpyAsync(flux_model, fl_d, (size_t)R.P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VAMP_OK;
}

int vamp_model_all(vamp_ctx* c, const double* theta, double* tau_comp, double* flux_model) {
    if (!c || !theta) return fail(VAMP_ERR_ARG, "vamp_model_all: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_model_all: call vamp_set_regions first");
    HIP_TRY(hipSetDevice(c->device));
    const RegionDev& last = c->regions_h.back();
    const size_t nth = (size_t)(last.d_before + last.D), ntau = (size_t)(last.tau_off + (long long)last.K * last.P);
    int pmax = 0;
    for (const RegionDev& R : c->regions_h) pmax = std::max(pmax, R.P);
    DevBuf th_b, tau_b, fl_b;
    HIP_TRY(hipMalloc(&th_b.p, nth * sizeof(double)));
    if (tau_comp) HIP_TRY(hipMalloc(&tau_b.p, ntau * sizeof(double)));
    if (flux_model) HIP_TRY(hipMalloc(&fl_b.p, (size_t)c->n_pix * sizeof(double)));
    double *th_d = th_b.as<double>(), *tau_d = tau_b.as<double>(), *fl_d = fl_b.as<double>();
    HIP_TRY(hipMemcpyAsync(th_d, theta, nth * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const dim3 grid((unsigned)((pmax + BLOCK - 1) / BLOCK), (unsigned)c->n_regions);
    VAMP_FOR_MODE(c->mode, hipLaunchKernelGGL((k_model<M>), grid, dim3(BLOCK), 0, c->stream, c->regions_d, -1, c->pix(), th_d, tau_d, fl_d));
    HIP_TRY(hipGetLastError());
    if (tau_comp) HIP_TRY(hipMemcpyAsync(tau_comp, tau_d, ntau * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (flux_model) HIP_TRY(hipMemcpyAsync(flux_model, fl_d, (size_t)c->n_pix * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VAMP_OK;
}

int vamp_line_records(vamp_ctx* c, int region, const double* theta1, double* rec, double* lnprior) {
    if (!c || !theta1 || !rec || !lnprior) return fail(VAMP_ERR_ARG, "vamp_line_records: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_line_records: call vamp_set_regions first");
    if (region < 0 || region >= c->n_regions) return fail(VAMP_ERR_ARG, "vamp_line_records: no such region");
    HIP_TRY(hipSetDevice(c->device));
    const RegionDev& R = c->regions_h[region];
    DevBuf th_b, rec_b;
    HIP_TRY(hipMalloc(&th_b.p, R.D * sizeof(double)));
    HIP_TRY(hipMalloc(&rec_b.p, (5 * R.K + 1) * sizeof(double)));
    double *th_d = th_b.as<double>(), *rec_d = rec_b.as<double>();
    HIP_TRY(hipMemcpyAsync(th_d, theta1, R.D * sizeof(double), hipMemcpyHostToDevice, c->stream));
    VAMP_FOR_MODE(c->mode, hipLaunchKernelGGL((k_line_records<M>), dim3(1), dim3(64), 0, c->stream, c->regions_d, region, th_d,
                                              rec_d, rec_d + 5 * R.K));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rec, rec_d, 5 * R.K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(lnprior, rec_d + 5 * R.K, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VAMP_OK;
}

int vamp_wofz_re(vamp_ctx* c, int64_t n, const double* x, const double* y, double* re_w) {
    if (!c || !x || !y || !re_w || n <= 0) return fail(VAMP_ERR_ARG, "vamp_wofz_re: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf x_b, y_b, o_b;
    HIP_TRY(hipMalloc(&x_b.p, n * sizeof(double)));
    HIP_TRY(hipMalloc(&y_b.p, n * sizeof(double)));
    HIP_TRY(hipMalloc(&o_b.p, n * sizeof(double)));
    double *x_d = x_b.as<double>(), *y_d = y_b.as<double>(), *o_d = o_b.as<double>();
    HIP_TRY(hipMemcpyAsync(x_d, x, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(y_d, y, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    if (c->f32) hipLaunchKernelGGL((k_wofz<true>), dim3(grid), dim3(BLOCK), 0, c->stream, (long long)n, x_d, y_d, o_d);
    else hipLaunchKernelGGL((k_wofz<false>), dim3(grid), dim3(BLOCK), 0, c->stream, (long long)n, x_d, y_d, o_d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(re_w, o_d, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VAMP_OK;
}

int vamp_sampler_bind_state(vamp_ctx* c, void* X_dev, void* lnp_dev) {
    if (!c || !X_dev || !lnp_dev) return fail(VAMP_ERR_ARG, "vamp_sampler_bind_state: NULL argument");
    free_sampler(c);
    c->X_d = (double*)X_dev;
    c->lnp_d = (double*)lnp_dev;
    c->X_ext = true;
    return VAMP_OK;
}

int vamp_sampler_init(vamp_ctx* c, int64_t W, const double* theta0, uint64_t seed, double a, int32_t split_block) {
    if (!c || !theta0) return fail(VAMP_ERR_ARG, "vamp_sampler_init: NULL argument");
    if (c->n_regions == 0) return fail(VAMP_ERR_STATE, "vamp_sampler_init: call vamp_set_regions first");
    if (W < 2 || (W & 1)) return fail(VAMP_ERR_ARG, "vamp_sampler_init: W must be even and >= 2");
    if (split_block < 2 || (split_block & 1) || W % split_block) return fail(VAMP_ERR_ARG, "vamp_sampler_init: split_block must be even and divide W");
    if (!(a > 1.0)) return fail(VAMP_ERR_ARG, "vamp_sampler_init: a must be > 1");
    HIP_TRY(hipSetDevice(c->device));
    RoctxRange range("vamp_sampler_init");
    HIP_TRY(hipStreamSynchronize(c->stream));
    long long tt = 0;
    for (int r = 0; r < c->n_regions; ++r) {
        c->regions_h[r].theta_off = tt;
        c->regions_h[r].walker_off = (long long)r * W;
        tt += (long long)W * c->regions_h[r].D;
    }
    HIP_TRY(hipMemcpy(c->regions_d, c->regions_h.data(), c->n_regions * sizeof(RegionDev), hipMemcpyHostToDevice));
    const bool ext = c->X_ext && c->X_d;
    if (!ext) free_sampler(c);
    if (c->nacc_d) { (void)hipFree(c->nacc_d); c->nacc_d = nullptr; }
    c->W = W;
    c->total_theta = tt;
    c->total_walkers = (long long)c->n_regions * W;
    c->split_block = split_block;
    c->a = a;
    c->seed = seed;
    c->step = 0;
    if (!ext) {
        c->X_ext = false;
        HIP_TRY(hipMalloc(&c->X_d, tt * sizeof(double)));
        HIP_TRY(hipMalloc(&c->lnp_d, c->total_walkers * sizeof(double)));
    }
    HIP_TRY(hipMalloc(&c->nacc_d, c->total_walkers * sizeof(long long)));
    HIP_TRY(hipMemsetAsync(c->nacc_d, 0, c->total_walkers * sizeof(long long), c->stream));
    HIP_TRY(hipMemcpyAsync(c->X_d, theta0, tt * sizeof(double), hipMemcpyHostToDevice, c->stream));
    // initial log-posteriors of every walker: the state has the layout of vamp_lnprob_all's arguments
    // (block r at W * d_before(r), lnprob of region r at r * W), so this is one launch per launch class
    {
        int rc = launch_lnprob(c, -1, W, c->X_d, c->lnp_d, nullptr);
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->shard_rank = 0;
    c->shard_world = 1;
    c->shard_parts = 1;
    c->part_slots = c->part_stride = 0;
    c->slot_begin = 0;
    c->slot_end = c->total_walkers / 2;
    c->sampler_ready = true;
    return VAMP_OK;
}

int vamp_sampler_set_shard_parts(vamp_ctx* c, int rank, int world, int parts, int64_t* own_begin, int64_t* own_end) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_sampler_set_shard: ctx is NULL");
    if (!c->sampler_ready) return fail(VAMP_ERR_STATE, "vamp_sampler_set_shard: call vamp_sampler_init first");
    if (world < 1 || rank < 0 || rank >= world) return fail(VAMP_ERR_ARG, "vamp_sampler_set_shard: bad rank/world");
    if (parts < 1 || parts > 64) return fail(VAMP_ERR_ARG, "vamp_sampler_set_shard: parts must be in 1..64");
    if (c->n_regions != 1) return fail(VAMP_ERR_ARG, "vamp_sampler_set_shard: walker sharding is for single-region contexts (shard regions across devices otherwise)");
    if (c->comm && (world != c->comm_world || rank != c->comm_rank))
        return fail(VAMP_ERR_ARG, "vamp_sampler_set_shard: rank/world differ from the communicator's (vamp_comm_init_rank)");
    // the ensemble is cut into `parts` equal row ranges and each of those into `world` shards (csrc/host_plan.hpp)
    vamp::plan::ShardPlan sp;
    {
        const std::string err = vamp::plan::plan_shard(c->W, c->split_block, rank, world, parts, sp);
        if (!err.empty()) return fail(VAMP_ERR_ARG, "vamp_sampler_set_shard: " + err);
    }
    c->shard_rank = rank;
    c->shard_world = world;
    c->shard_parts = parts;
    c->part_slots = sp.part_slots;
    c->part_stride = sp.part_stride;
    c->slot_begin = sp.slot_begin;
    c->slot_end = c->slot_begin + c->part_slots;      // of part 0
    // exchange buffers: the movers of every part in slot order, position + lnprob per row
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->send_d) (void)hipFree(c->send_d);
    if (c->recv_d) (void)hipFree(c->recv_d);
    c->send_d = c->recv_d = nullptr;
    if (world > 1 || c->comm) {
        int rc = alloc_exchange_buffers(c);
        if (rc) return rc;
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
    if (!c->sampler_ready) return fail(VAMP_ERR_STATE, "vamp_sampler_state_ptrs: call vamp_sampler_init first");
    if (X_dev) *X_dev = c->X_d;
    if (lnp_dev) *lnp_dev = c->lnp_d;
    if (total_theta) *total_theta = c->total_theta;
    if (total_walkers) *total_walkers = c->total_walkers;
    return VAMP_OK;
}

int vamp_sampler_half_step(vamp_ctx* c, int half) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step: ctx is NULL");
    if (!c->sampler_ready) return fail(VAMP_ERR_STATE, "vamp_sampler_half_step: call vamp_sampler_init first");
    if (half != 0 && half != 1) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step: half must be 0 or 1");
    HIP_TRY(hipSetDevice(c->device));
    int rc = half_step_all(c, half);
    if (rc) return rc;
    if (half == 1) c->step += 1;
    return VAMP_OK;
}

int vamp_sampler_half_step_part(vamp_ctx* c, int half, int part) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_part: ctx is NULL");
    if (!c->sampler_ready) return fail(VAMP_ERR_STATE, "vamp_sampler_half_step_part: call vamp_sampler_init first");
    if (half != 0 && half != 1) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_part: half must be 0 or 1");
    if (part < 0 || part >= c->shard_parts) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_part: no such part");
    if (c->comm && c->send_d)
        return fail(VAMP_ERR_STATE, "vamp_sampler_half_step_part: with a communicator the exchange is part of vamp_sampler_half_step / vamp_sampler_run");
    HIP_TRY(hipSetDevice(c->device));
    int rc = launch_half(c, half, false, 0, 0, part);
    if (rc) return rc;
    if (half == 1 && part == c->shard_parts - 1) c->step += 1;
    return VAMP_OK;
}

int vamp_sampler_half_step_ext(vamp_ctx* c, int region, int64_t n, const int32_t* active_idx, const int32_t* partner_idx,
                               const double* zz, const double* logu) {
    if (!c || !active_idx || !partner_idx || !zz || !logu) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_ext: NULL argument");
    if (!c->sampler_ready) return fail(VAMP_ERR_STATE, "vamp_sampler_half_step_ext: call vamp_sampler_init first");
    if (region < 0 || region >= c->n_regions) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_ext: no such region");
    if (n <= 0 || n > c->W) return fail(VAMP_ERR_ARG, "vamp_sampler_half_step_ext: bad n");
    // validate on the host: a wild index would be an out-of-bounds device access
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
    HIP_TRY(hipSetDevice(c->device));
    if (c->ext_cap < n) {
        for (void* p : {(void*)c->ext_act_d, (void*)c->ext_par_d, (void*)c->ext_z_d, (void*)c->ext_lu_d})
            if (p) (void)hipFree(p);
        c->ext_cap = 0;
        HIP_TRY(hipMalloc(&c->ext_act_d, n * sizeof(int)));
        HIP_TRY(hipMalloc(&c->ext_par_d, n * sizeof(int)));
        HIP_TRY(hipMalloc(&c->ext_z_d, n * sizeof(double)));
        HIP_TRY(hipMalloc(&c->ext_lu_d, n * sizeof(double)));
        c->ext_cap = n;
    }
    HIP_TRY(hipMemcpyAsync(c->ext_act_d, active_idx, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ext_par_d, partner_idx, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ext_z_d, zz, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ext_lu_d, logu, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int rc = launch_half(c, 0, true, region, n);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VAMP_OK;
}

int vamp_sampler_run_dev(vamp_ctx* c, int64_t n_steps, int thin, double* chain_dev, double* lnprob_chain_dev, double* seconds) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_sampler_run_dev: ctx is NULL");
    if (!c->sampler_ready) return fail(VAMP_ERR_STATE, "vamp_sampler_run_dev: call vamp_sampler_init first");
    if (n_steps < 0 || thin < 1) return fail(VAMP_ERR_ARG, "vamp_sampler_run_dev: n_steps >= 0 and thin >= 1 required");
    if (c->shard_world != 1 && !(c->comm && c->send_d))
        return fail(VAMP_ERR_STATE, "vamp_sampler_run_dev: a sharded context without a communicator is stepped by the host "
                                    "(half_step_part + pack_get / scatter_put)");
    HIP_TRY(hipSetDevice(c->device));
    RoctxRange range("vamp_sampler_run");
    const long long n_keep = n_steps / thin;
    HIP_TRY(hipStreamSynchronize(c->stream));
    const auto t0 = std::chrono::steady_clock::now();
    long long kept = 0;
    if (n_steps > 0 && resident_eligible(c)) {
        // small ensembles: every region's whole step loop in ONE launch per launch class (k_run_resident)
        RoctxRange res_range("vamp resident step loop");
        int rc = run_resident(c, n_steps, thin, n_keep ? chain_dev : nullptr, n_keep ? lnprob_chain_dev : nullptr);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return VAMP_OK;
    }
    for (long long it = 0; it < n_steps; ++it) {
        for (int half = 0; half < 2; ++half) {
            int rc = half_step_all(c, half);
            if (rc) return rc;
        }
        c->step += 1;
        if ((it + 1) % thin == 0 && kept < n_keep) {
            if (chain_dev)
                HIP_TRY(hipMemcpyAsync(chain_dev + kept * c->total_theta, c->X_d, c->total_theta * sizeof(double),
                                       hipMemcpyDeviceToDevice, c->stream));
            if (lnprob_chain_dev)
                HIP_TRY(hipMemcpyAsync(lnprob_chain_dev + kept * c->total_walkers, c->lnp_d, c->total_walkers * sizeof(double),
                                       hipMemcpyDeviceToDevice, c->stream));
            ++kept;
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    const auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    return VAMP_OK;
}

int vamp_sampler_run(vamp_ctx* c, int64_t n_steps, int thin, double* chain, double* lnprob_chain, int64_t* n_accept,
                     double* seconds) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_sampler_run: ctx is NULL");
    if (!c->sampler_ready) return fail(VAMP_ERR_STATE, "vamp_sampler_run: call vamp_sampler_init first");
    if (n_steps < 0 || thin < 1) return fail(VAMP_ERR_ARG, "vamp_sampler_run: n_steps >= 0 and thin >= 1 required");
    HIP_TRY(hipSetDevice(c->device));
    const long long n_keep = n_steps / thin;
    DevBuf chain_b, lchain_b;
    if (chain && n_keep) HIP_TRY(hipMalloc(&chain_b.p, (size_t)n_keep * c->total_theta * sizeof(double)));
    if (lnprob_chain && n_keep) HIP_TRY(hipMalloc(&lchain_b.p, (size_t)n_keep * c->total_walkers * sizeof(double)));
    double *chain_d = chain_b.as<double>(), *lchain_d = lchain_b.as<double>();
    int rc = vamp_sampler_run_dev(c, n_steps, thin, chain_d, lchain_d, seconds);
    if (rc) return rc;
    if (chain_d) {
        HIP_TRY(hipMemcpy(chain, chain_d, (size_t)n_keep * c->total_theta * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (lchain_d) {
        HIP_TRY(hipMemcpy(lnprob_chain, lchain_d, (size_t)n_keep * c->total_walkers * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (n_accept) HIP_TRY(hipMemcpy(n_accept, c->nacc_d, c->total_walkers * sizeof(long long), hipMemcpyDeviceToHost));
    return VAMP_OK;
}

// ---- walker-sharded multi-GPU runs ------------------------------------------------------------
int vamp_comm_unique_id(char* id) {
    if (!id) return fail(VAMP_ERR_ARG, "vamp_comm_unique_id: id is NULL");
    RcclApi* api = nullptr;
    int rc = rccl_api(&api);
    if (rc) return rc;
    RcclUniqueId u;
    RCCL_TRY(api, api->GetUniqueId(&u));
    std::memcpy(id, u.internal, VAMP_COMM_ID_BYTES);
    return VAMP_OK;
}

int vamp_comm_init_rank(vamp_ctx* c, const char* id, int rank, int world) {
    if (!c || !id) return fail(VAMP_ERR_ARG, "vamp_comm_init_rank: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(VAMP_ERR_ARG, "vamp_comm_init_rank: bad rank/world");
    if (c->comm) return fail(VAMP_ERR_STATE, "vamp_comm_init_rank: the context already has a communicator");
    // either order of vamp_sampler_set_shard_parts and vamp_comm_init_rank is accepted, but they must agree
    if (c->sampler_ready && (c->shard_world != 1 || c->send_d) && (c->shard_world != world || c->shard_rank != rank))
        return fail(VAMP_ERR_ARG, "vamp_comm_init_rank: rank/world differ from the shard already set (vamp_sampler_set_shard_parts)");
    RcclApi* api = nullptr;
    int rc = rccl_api(&api);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    // RCCL checks the runtime's sticky last error during initialisation: one left behind by an earlier call of
    // anyone in this process (a failed allocation, a rejected argument) would surface as "unhandled cuda error"
    (void)hipGetLastError();
    RcclUniqueId u;
    std::memcpy(u.internal, id, VAMP_COMM_ID_BYTES);
    void* comm = nullptr;
    RCCL_TRY(api, api->CommInitRank(&comm, world, u, rank));
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_world = world;
    HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    // the shard came first: its exchange buffers (a single-rank shard has none yet) and per-part events are due now
    if (c->sampler_ready && c->n_regions == 1 && c->part_slots > 0) {
        if (!c->send_d) rc = alloc_exchange_buffers(c);
        else rc = ensure_part_events(c, c->shard_parts);
        if (rc) return rc;
    }
    return VAMP_OK;
}

int vamp_comm_info(vamp_ctx* c, int* rank, int* world, int* queried) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_comm_info: ctx is NULL");
    if (!c->comm) return fail(VAMP_ERR_STATE, "vamp_comm_info: the context has no communicator (vamp_comm_init_rank)");
    int r = c->comm_rank, w = c->comm_world, q = 0;
    RcclApi* api = nullptr;
    if (rccl_api(&api) == 0 && api->CommCount && api->CommUserRank) {
        int w2 = 0, r2 = 0;
        RCCL_TRY(api, api->CommCount(c->comm, &w2));
        RCCL_TRY(api, api->CommUserRank(c->comm, &r2));
        r = r2; w = w2; q = 1;
    }
    if (rank) *rank = r;
    if (world) *world = w;
    if (queried) *queried = q;
    return VAMP_OK;
}

int vamp_comm_library(char* path, int64_t capacity) {
    if (!path || capacity < 2) return fail(VAMP_ERR_ARG, "vamp_comm_library: no room for a path");
    RcclApi* api = nullptr;
    int rc = rccl_api(&api);
    if (rc) return rc;
    Dl_info info;
    const char* name = (dladdr(reinterpret_cast<void*>(api->AllGather), &info) && info.dli_fname) ? info.dli_fname : "";
    std::snprintf(path, (size_t)capacity, "%s", name);
    return VAMP_OK;
}

int vamp_comm_destroy(vamp_ctx* c) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_comm_destroy: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->comm_stream) HIP_TRY(hipStreamSynchronize(c->comm_stream));
    free_comm(c);
    return VAMP_OK;
}

int vamp_sampler_pack_get(vamp_ctx* c, int part, double* rows) {
    if (!c || !rows) return fail(VAMP_ERR_ARG, "vamp_sampler_pack_get: NULL argument");
    if (!c->sampler_ready || !c->send_d) return fail(VAMP_ERR_STATE, "vamp_sampler_pack_get: no sharded sampler (vamp_sampler_set_shard_parts with world > 1)");
    if (part < 0 || part >= c->shard_parts) return fail(VAMP_ERR_ARG, "vamp_sampler_pack_get: no such part");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t n = (size_t)c->part_slots * (c->regions_h[0].D + 1);
    HIP_TRY(hipMemcpy(rows, c->send_d + (size_t)part * n, n * sizeof(double), hipMemcpyDeviceToHost));
    return VAMP_OK;
}

int vamp_sampler_scatter_put(vamp_ctx* c, int part, const double* rows_all) {
    if (!c || !rows_all) return fail(VAMP_ERR_ARG, "vamp_sampler_scatter_put: NULL argument");
    if (!c->sampler_ready || !c->recv_d) return fail(VAMP_ERR_STATE, "vamp_sampler_scatter_put: no sharded sampler (vamp_sampler_set_shard_parts with world > 1)");
    if (part < 0 || part >= c->shard_parts) return fail(VAMP_ERR_ARG, "vamp_sampler_scatter_put: no such part");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)c->shard_world * c->part_slots * (c->regions_h[0].D + 1);
    HIP_TRY(hipMemcpyAsync(c->recv_d + (size_t)part * n, rows_all, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int rc = launch_scatter(c, part, c->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VAMP_OK;
}

int vamp_sampler_get_state(vamp_ctx* c, double* theta, double* lnprob, int64_t* n_accept, int64_t* step) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_sampler_get_state: ctx is NULL");
    if (!c->sampler_ready) return fail(VAMP_ERR_STATE, "vamp_sampler_get_state: call vamp_sampler_init first");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (theta) HIP_TRY(hipMemcpy(theta, c->X_d, c->total_theta * sizeof(double), hipMemcpyDeviceToHost));
    if (lnprob) HIP_TRY(hipMemcpy(lnprob, c->lnp_d, c->total_walkers * sizeof(double), hipMemcpyDeviceToHost));
    if (n_accept) HIP_TRY(hipMemcpy(n_accept, c->nacc_d, c->total_walkers * sizeof(long long), hipMemcpyDeviceToHost));
    if (step) *step = c->step;
    return VAMP_OK;
}

int vamp_sampler_set_state(vamp_ctx* c, const double* theta, const double* lnprob, int64_t step) {
    if (!c || !theta || !lnprob) return fail(VAMP_ERR_ARG, "vamp_sampler_set_state: NULL argument");
    if (!c->sampler_ready) return fail(VAMP_ERR_STATE, "vamp_sampler_set_state: call vamp_sampler_init first");
    if (step < 0) return fail(VAMP_ERR_ARG, "vamp_sampler_set_state: step must be >= 0");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(c->X_d, theta, c->total_theta * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->lnp_d, lnprob, c->total_walkers * sizeof(double), hipMemcpyHostToDevice));
    c->step = step;
    return VAMP_OK;
}

int vamp_exchange_timing(vamp_ctx* c, double* total_ms, int64_t* exchanges) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_exchange_timing: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    int rc = flush_exchange_timing(c);
    if (rc) return rc;
    if (total_ms) *total_ms = c->xtiming_ms;
    if (exchanges) *exchanges = c->xtiming_n;
    c->xtiming_ms = 0.0;
    c->xtiming_n = 0;
    return VAMP_OK;
}

#ifdef VAMP_STAMPS
// timing-only builds: copy out and reset the (tag, clock) stamps; returns their number
int vamp_debug_stamps(unsigned long long* out, int cap) {
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_stamp_n), sizeof(n)) != hipSuccess) return -1;
    if ((int)n > cap) n = (unsigned)cap;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), (size_t)n * 2 * sizeof(unsigned long long)) != hipSuccess) return -1;
    const unsigned int zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_n), &zero, sizeof(zero)) != hipSuccess) return -1;
    return (int)n;
}
#endif

int vamp_kernel_timing(vamp_ctx* c, int enable, double* total_ms, int64_t* launches) {
    if (!c) return fail(VAMP_ERR_ARG, "vamp_kernel_timing: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    int rc = flush_timing(c);
    if (rc) return rc;
    rc = flush_exchange_timing(c);
    if (rc) return rc;
    if (total_ms) *total_ms = c->timing_ms;
    if (launches) *launches = c->timing_launches;
    c->timing_ms = 0.0;
    c->timing_launches = 0;
    c->timing = enable != 0;
    return VAMP_OK;
}

}  // extern "C"
