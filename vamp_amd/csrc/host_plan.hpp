// host_plan.hpp -- the pure-host arithmetic behind the launches of libvamp_hip.so: which launch class a region
// falls into, how an ensemble is cut into shards and pieces, how large the exchange buffers are, how many
// workgroups a packed launch takes, which contexts the device-resident step loop takes.  No HIP in here: the
// same header is compiled into oracle/libvamp_cpu.so (the host implementation of the C ABI), whose
// AddressSanitizer + UndefinedBehaviorSanitizer build runs it under tests/test_sanitizers.py -- GPU sanitizers
// are not available on this pool, this is how the product's own host code gets sanitizer coverage.
//
// Reference: none of this exists there (one process, one region at a time: vpspectrum.py:273-348); the regions'
// sizes it plans for are the reference's (vpspectrum.py:287-294: more than 15 / 22.5 lines are "difficult").
#pragma once

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#if defined(__HIPCC__)
#define VAMP_PLAN_HD __host__ __device__ inline
#else
#define VAMP_PLAN_HD inline
#endif

namespace vamp {
namespace plan {

// ---- launch classes -----------------------------------------------------------------------------------
enum ClassKind { CK_SMALL = 0, CK_MID = 1, CK_WIDE = 2, CK_SMALL2 = 3, CK_XL = 4 };

struct Limits {                 // what the kernel shapes hold (static_asserted against struct Pack in vamp_hip.hip)
    int kmax = 16;              // lines per region of the fast shapes
    int kmax_all = 32;          // VAMP_MAX_COMPONENTS
    int small_kcap = 8;         // Pack<16, 8>: lines per walker of the packed shape
    int small2_kcap = 2;        // Pack<8, 2>
    int mid_min_k = 3;          // blends: >= 3 lines ...
    int mid_min_p = 96;         // ... over >= 96 pixels ...
    int blend_max_p = 512;      // ... and at most 512 (the optical depths of a lane's <= 8 pixels stay in registers)
    int tile = 256;             // pixels of a far-field tile (64 lanes x 4 pixels)
    double mean_p_max = 128.0;  // a context "looks like a spectrum" when its mean region is at most this long
};

struct RegionShape { int P, K; };

struct ClassPlan {
    std::vector<int> kind;                   // per class: ClassKind, in launch order (blends first: the longest launch)
    std::vector<std::vector<int>> regions;   // per class: its regions, ascending
    std::vector<int> class_of;               // region -> class
    bool full_tiles = true;                  // every region of <= kmax lines is a whole number of tiles
    int min_tiles = 0;                       // full tiles of the shortest such region
    bool spectrum_like = false;
};

// packing: 0 automatic, 16 / 64 / 65 / 256 forced (vamp_ctx_set_packing).  gauss: Gaussian components (no tables).
// tables_f32: fp32 contexts have single-precision Taylor rows for their blends.  Returns "" or an error message.
// merged: the partition for SMALL ENSEMBLES (<= 128 movers per region: model-selection ladders, single points, the MAP
// search).  There a half-step is bound by its launches, not by its arithmetic -- every class is a launch of its own,
// ~15 us each on one stream, where the whole half-step of 1218 short regions is ~25 us of work -- so all regions of
// <= 8 lines that are not blends form ONE class (four walkers per wavefront serve any of them).
inline std::string plan_classes(const std::vector<RegionShape>& R, int packing, bool gauss, bool f32, bool tables_f32, ClassPlan& out,
                                const Limits& lim = Limits(), bool merged = false) {
    const int n = (int)R.size();
    int kmax = 0, n_std = 0;
    long long pix_std = 0;
    out = ClassPlan();
    out.min_tiles = 0x7fffffff;
    for (int r = 0; r < n; ++r) {
        if (R[r].K < 1 || R[r].K > lim.kmax_all) return "n_comp out of range";
        if (R[r].P < 2) return "a region needs >= 2 pixels";
        if (R[r].K > lim.kmax) continue;                   // regions of 17+ lines: their own class, everything below is about the others
        kmax = std::max(kmax, R[r].K);
        n_std += 1;
        pix_std += R[r].P;
        out.full_tiles = out.full_tiles && (R[r].P % lim.tile == 0);
        out.min_tiles = std::min(out.min_tiles, R[r].P / lim.tile);
    }
    if (n_std == 0) out.min_tiles = 0;
    const double mean_p = n_std ? (double)pix_std / n_std : 0.0;
    if ((packing == 16 || packing == 65) && (kmax > lim.small_kcap || n_std < n))
        return "packings 16 and 65 support at most 8 components per region";
    // A context "looks like a spectrum" when its regions of <= kmax lines are short on average: judged over those only (a
    // region of 17+ lines has its own class whatever the others run), and WITHOUT a limit on their line counts: a model-
    // selection ladder grows a few regions past 8 lines, and those few take the one-walker-per-wavefront class on their
    // own instead of dragging hundreds of one- and two-line regions there (round 4: the second rung of the q1422 ladder,
    // 1218 regions, one of them with 9 lines: 94 us per half-step, all of it in the wide class).
    out.spectrum_like = packing == 0 && n_std > 0 && mean_p <= lim.mean_p_max;
    std::vector<int> cls[5];
    const int kind0 = packing == 16 ? CK_SMALL : packing == 65 ? CK_MID : out.spectrum_like ? CK_SMALL : CK_WIDE;
    const int kinds[5] = {kind0, CK_MID, CK_SMALL2, CK_XL, CK_WIDE};
    for (int r = 0; r < n; ++r) {
        int k = 0;
        if (R[r].K > lim.kmax) {
            k = 3;
        } else if (out.spectrum_like) {
            if (R[r].K > lim.small_kcap) k = 4;           // 9 .. 16 lines: one walker per wavefront
            else if (R[r].K >= lim.mid_min_k && R[r].P >= lim.mid_min_p && R[r].P <= lim.blend_max_p && !gauss && (!f32 || tables_f32)) k = 1;
            else if (R[r].K <= lim.small2_kcap && !merged) k = 2;
        }
        cls[k].push_back(r);
    }
    out.class_of.assign(n, 0);
    for (int k : {1, 2, 0, 4, 3})
        if (!cls[k].empty()) {
            for (int r : cls[k]) out.class_of[r] = (int)out.kind.size();
            out.kind.push_back(kinds[k]);
            out.regions.push_back(cls[k]);
        }
    return "";
}

// ---- walker sharding (vamp_sampler_set_shard_parts) -------------------------------------------------------
// The ensemble (W walkers in chunks of split_block, half of every chunk moving per half-step) is cut into `parts`
// equal slot ranges and each of those into `world` shards: part p of rank r = chunks [p * chunks/parts + r * cpp, + cpp).
struct ShardPlan {
    long long part_slots = 0;      // active slots of one piece of one rank
    long long part_stride = 0;     // distance between this rank's pieces, in slots
    long long slot_begin = 0;      // of piece 0
    std::vector<long long> own_begin, own_end;    // rows of the walkers this rank owns, per piece (whole split chunks)
};
inline std::string plan_shard(long long W, int split_block, int rank, int world, int parts, ShardPlan& out) {
    if (world < 1 || rank < 0 || rank >= world) return "bad rank/world";
    if (parts < 1 || parts > 64) return "parts must be in 1..64";
    if (split_block < 2 || (split_block & 1) || W % split_block) return "split_block must be even and divide W";
    const long long chunks = W / split_block;
    if (chunks % ((long long)world * parts)) return "W/split_block must be a multiple of world * parts";
    const long long cpp = chunks / ((long long)world * parts), hb = split_block / 2;
    out.part_slots = cpp * hb;
    out.part_stride = (chunks / parts) * hb;
    out.slot_begin = (long long)rank * cpp * hb;
    out.own_begin.assign(parts, 0);
    out.own_end.assign(parts, 0);
    for (int p = 0; p < parts; ++p) {
        const long long first = (long long)p * (chunks / parts) + (long long)rank * cpp;
        out.own_begin[p] = first * split_block;
        out.own_end[p] = (first + cpp) * split_block;
    }
    return "";
}
// exchange buffers: the movers of every piece in slot order, position + lnprob per row
inline size_t exchange_send_doubles(int parts, long long part_slots, int D) { return (size_t)parts * (size_t)part_slots * (size_t)(D + 1); }
inline size_t exchange_recv_doubles(int parts, int world, long long part_slots, int D) {
    return (size_t)parts * (size_t)world * (size_t)part_slots * (size_t)(D + 1);
}

// ---- packed launches of several regions: wavefronts are dealt per region -----------------------------------
struct PackedGrid { int wpr = 0; long long grid = 0; int bpr = 0; };     // wavefronts / workgroups per region, workgroups in all
inline PackedGrid plan_packed_grid(long long n_regions, long long half_w, int subs, int waves_per_block) {
    PackedGrid g;
    g.wpr = (int)((half_w + subs - 1) / subs);
    g.grid = (n_regions * g.wpr + waves_per_block - 1) / waves_per_block;
    g.bpr = g.wpr % waves_per_block == 0 ? g.wpr / waves_per_block : 0;     // workgroups align with regions: region -> XCD mapping
    return g;
}
// the region -> XCD mapping of k_half_step: physical workgroup b (on XCD b % 8) -> logical workgroup
VAMP_PLAN_HD long long xcd_map(long long b, long long n_regions, long long bpr) {
    const long long mapped = (n_regions & ~7ll) * bpr;
    if (bpr <= 0 || b >= mapped) return b;
    const long long x = b & 7, j = b >> 3, jr = j / bpr;
    return (x + 8 * jr) * bpr + (j - jr * bpr);
}

// ---- device-resident step loop: automatic policy ---------------------------------------------------------------
// (measured, profiles/r04_c_small_ensembles.txt) at most one workgroup per compute unit, every mover of a half-step in
// one round of the workgroup's wavefronts, only the packed short-region classes
struct ResidentLimits { int max_movers = 128; int max_regions = 256; };
inline bool resident_class_ok(int kind, long long half_w, int compute_waves, int walkers_per_wave, bool automatic) {
    if (compute_waves <= 0) return false;
    if (!automatic) return true;
    if (!(kind == CK_SMALL || kind == CK_SMALL2)) return false;
    return (long long)compute_waves * walkers_per_wave >= half_w;
}

}  // namespace plan
}  // namespace vamp
