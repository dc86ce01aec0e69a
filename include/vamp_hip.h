/* vamp_hip.h -- C ABI of libvamp_hip.so, the MI355X (gfx950) implementation of VAMP's MCMC hot
 * path: per-walker log-posterior (Voigt/Gaussian optical depth -> flux = exp(-tau) -> Gaussian
 * chi^2 + priors) and the affine-invariant stretch move.
 *
 * The reference (sarahappleby/VAMP, pure Python) has no FFI/plugin interface; the boundary this
 * library sits behind is the Python surface of class VPfit (vamp_1.0/vpfits.py:33).  Each entry
 * point below names the reference code it replaces.  The Python side binds these with ctypes
 * (vamp_amd/_lib.py); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative vamp_status; the message is available
 *     from vamp_last_error() (thread-local storage owned by the library, valid until the next
 *     failing call on that thread).  Nothing aborts or throws across the ABI.
 *   - all host buffers are caller-owned, contiguous, C order; floating point is double and
 *     indices are int32/int64 regardless of the device arithmetic type.  The library copies in
 *     on set/init and out on run/get.  A vamp_ctx owns its device memory, stream and RNG state.
 *   - a vamp_ctx is not thread-safe; use one per device.  ctypes releases the GIL during calls.
 *   - non-finite parameters give that walker lnprob = -inf (vpfits.py:241-242 convention), not
 *     an error.
 */
#ifndef VAMP_HIP_H
#define VAMP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vamp_ctx vamp_ctx;

enum vamp_status {
    VAMP_OK = 0,
    VAMP_ERR_ARG = -1,      /* bad argument */
    VAMP_ERR_HIP = -2,      /* HIP runtime error */
    VAMP_ERR_COMM = -3,     /* RCCL error (vamp_comm_*, the exchange inside a sharded half-step) */
    VAMP_ERR_NOMEM = -4,    /* out of device or host memory */
    VAMP_ERR_STATE = -5     /* not initialised / wrong call order */
};

/* parameterisation of one component */
enum vamp_mode {
    VAMP_GAUSS3 = 0,  /* (amplitude, centroid, sigma)            vpfits.py:219-262 */
    VAMP_VOIGT4 = 1,  /* (amplitude, centroid, L_fwhm, G_fwhm)   vpfits.py:265-307 */
    VAMP_NBZ3 = 2     /* (N, b, z) through physics.py:6-27,116-134; Lorentzian FWHM fixed */
};

enum vamp_dtype { VAMP_F64 = 0, VAMP_F32 = 1 };          /* per-pixel arithmetic type */
enum vamp_wofz { VAMP_WOFZ_ACCURATE = 0, VAMP_WOFZ_HUMLICEK_W4 = 1 };

#define VAMP_MAX_COMPONENTS 32
#define VAMP_ABI_VERSION 4
#define VAMP_COMM_ID_BYTES 128   /* = NCCL_UNIQUE_ID_BYTES */

/* library identification */
int vamp_version(void);
const char* vamp_last_error(void);
int vamp_device_count(int* n);

/* context = one device + one stream.  dtype/wofz_kind choose the per-pixel arithmetic:
 * (VAMP_F64, VAMP_WOFZ_ACCURATE) is the parity path; (VAMP_F32, VAMP_WOFZ_HUMLICEK_W4) the
 * throughput path of BASELINE.json config 5.  Walker state is always double. */
int vamp_ctx_create(vamp_ctx** out, int device, int dtype, int wofz_kind);
int vamp_ctx_destroy(vamp_ctx* ctx);
/* run on a caller-provided hipStream_t instead of the ctx's own (non-blocking) stream;
 * hip_stream = NULL returns to the ctx's own stream.  A caller whose other work runs on HIP's
 * legacy default stream (handle 0: e.g. torch without an explicit stream) adopts THAT stream with
 * vamp_ctx_set_stream_default -- the ctx's own stream does not synchronise with it. */
int vamp_ctx_set_stream(vamp_ctx* ctx, void* hip_stream);
int vamp_ctx_set_stream_default(vamp_ctx* ctx);
int vamp_ctx_synchronize(vamp_ctx* ctx);
/* Run-time switches of a context (all default to 1); an unknown name is VAMP_ERR_ARG.
 *   "map_device"     1: vamp_map_all runs every region's whole Nelder-Mead search in one launch (one workgroup
 *                    per region); 0: the host drives the same search, one launch + one synchronisation per
 *                    iteration of all regions.  Same rules, same arithmetic: identical results.
 *   "resident"       1: vamp_sampler_run(_dev) steps small ensembles with the whole step loop inside ONE launch per
 *                    launch class (one workgroup per region) where that pays: contexts of at most 256 short regions
 *                    (<= 8 components, the packed shapes) whose movers of a half-step (W / 2 <= 128) fit one round of
 *                    the workgroup; 2: wherever the kernel can run (every shape but the workgroup-per-walker one);
 *                    0: always one launch per half-step.  Same draws, same arithmetic: identical chains.
 *   "class_streams"  1: the launch classes of a half-step run concurrently on forked streams; 0: one after the
 *                    other (what the per-class profiles use).  Also set by VAMP_CLASS_STREAMS in the environment. */
int vamp_ctx_set_option(vamp_ctx* ctx, const char* name, int64_t value);
/* Lanes that serve one walker:
 *   64   one walker per wavefront;
 *   256  one walker per 4-wavefront workgroup: each wavefront sweeps every 4th 256-pixel tile and, in
 *        fp64, the line cores are read from per-line Taylor tables shared by the workgroup (long regions);
 *   16   four walkers per wavefront, <= 8 components per region (the short regions of real
 *        spectra; draws come from a one-thread-per-mover launch; automatic packing uses eight
 *        walkers per wavefront, 8 lanes each, for regions of one or two components when the launch
 *        fills the chip or the ensemble has 33 .. 128 movers per region);
 *   65   64 lanes + the walker's own Taylor tables (four lines' at a time), <= 8 components, no far
 *        field (the blended regions of real spectra: a few lines over a few hundred pixels; regions
 *        of more than 512 pixels fall back to per-pixel evaluation without tables);
 *   Regions of 17 .. VAMP_MAX_COMPONENTS components always run one walker per wavefront with every line
 *   evaluated per pixel (no far-field interpolant, no Taylor tables: those shapes hold 16 lines) -- the
 *   reference plans for such regions (vpspectrum.py:287-294: more than max_single_region_components = 15
 *   lines means fewer attempts and a laxer chi^2 limit, not a refusal), so they work; they are not fast.
 *   0    choose: contexts that look like a real spectrum (the regions of <= 16 components are <= 128 pixels
 *        long on average) are split into launch classes -- regions with 3..8 components over 96..512 pixels
 *        run as 65, regions of 9..16 components as 64, the rest as 16 when the launch fills the chip
 *        (>= 16 384 movers) or the ensemble is small (<= 128 movers per region: model-selection ladders, single
 *        points, the MAP search, the device-resident loop) and as 64 in between; otherwise 256 when every region
 *        has >= 2048 pixels, else 64.
 * The choice never depends on how an ensemble is sharded, so a shard runs the arithmetic of the
 * whole ensemble.  Takes effect at the next vamp_set_regions.  fp64: all shapes agree to rounding.
 * fp32: 256 evaluates the line cores through single-precision Taylor rows instead of Humlicek's
 * regions III / IV, so it agrees with the other shapes to W4's own error (~1e-4 relative in H). */
int vamp_ctx_set_packing(vamp_ctx* ctx, int lanes_per_walker);

/* Upload the data of n_regions independent absorption regions (replaces
 * VPfit.initialise_model's capture of frequency/flux/noise, vpfits.py:310-349, and the prior
 * bounds of vpfits.py:249-252,292-297,320,326).
 *   pix_off[n_regions+1]  CSR offsets into x/flux/noise
 *   x                     abscissa, finite and strictly monotonic inside a region (either direction;
 *                         anything else is VAMP_ERR_ARG); same units as the width params
 *   n_comp[n_regions]     components per region (<= VAMP_MAX_COMPONENTS)
 *   mode                  vamp_mode, one for all regions
 *   sample_sd             1 = reference likelihood with free precision sd~U(0,1) as the last
 *                         dimension (vpfits.py:39,341); 0 = known per-pixel noise, -chi^2/2
 *   include_norm          add -1/2 sum log(2 pi sigma_i^2) (vamp_2.0/vamp_src/fit/fit.py:156)
 *   bounds[n_regions*4]   {c_lo, c_hi, sigma_max, fwhm_max} per region, or NULL to derive them
 *                         from x as vpfits.py:250,320,326 do
 *   nbz[n_regions*4]      {l_fixed, line[A], x_origin[Hz], x_scale[Hz]} for VAMP_NBZ3, else NULL */
int vamp_set_regions(vamp_ctx* ctx, int n_regions, const int64_t* pix_off, const double* x,
                     const double* flux, const double* noise, const int32_t* n_comp, int mode,
                     int sample_sd, int include_norm, const double* bounds, const double* nbz);

/* Identity of every region in the sampler's draw keys (default: its index in this context).  A
 * spectrum whose regions are spread over several contexts / devices (independent posteriors: no
 * collective, SURVEY 8e "Multi-region") passes each region's index in the WHOLE spectrum, and every
 * region then follows the chain it follows in the single-context batch. */
int vamp_set_region_ids(vamp_ctx* ctx, const int32_t* ids);

/* Launch class of a region: 0 short (four walkers per wavefront), 1 blend (a wavefront and per-walker Taylor tables
 * per walker), 2 wide (one walker per wavefront or per 4-wavefront workgroup), 3 short with one or two components
 * (eight walkers per wavefront), 4 more than 16 components -- and the number of classes of the context (one launch
 * per class and half-step).  This is the partition of large ensembles; for small ones (<= 128 movers per region)
 * classes 0 and 3 are one class, four walkers per wavefront: a launch less per half-step.  Diagnostics: which kernels a context will run; see vamp_ctx_set_packing.  Either
 * output may be NULL. */
int vamp_region_class(vamp_ctx* ctx, int region, int* kind, int* n_classes);
/* number of sampled dimensions of a region (q*K, +1 with sample_sd) */
int vamp_region_ndim(vamp_ctx* ctx, int region, int* ndim);

/* Log-posterior of W parameter vectors theta[W, D] for one region: replaces one PyMC
 * evaluation of the model graph (profile closures vpfits.py:254-260/299-305, total :334-336,
 * obs :341, priors :239-252/283-297).  chi2 may be NULL; with sample_sd it receives the
 * unweighted sum of squared residuals. */
int vamp_lnprob(vamp_ctx* ctx, int region, int64_t W, const double* theta, double* lnprob,
                double* chi2);
/* The same for EVERY region in one launch: theta holds the regions' [W, D_r] blocks one after the
 * other, lnprob (and chi2, may be NULL) are [n_regions, W].  Serves the simultaneous MAP searches
 * of all regions of a spectrum (vpfits.py:352-358, 426 per region in the reference). */
int vamp_lnprob_all(vamp_ctx* ctx, int64_t W, const double* theta, double* lnprob, double* chi2);
/* Maximum a posteriori search for every region at once: replaces mc.MAP(model).fit(iterlim, tol)
 * (vpfits.py:352-358, 426), i.e. scipy's Nelder-Mead `fmin` on -logp, one region after the other
 * in the reference.  Same simplex rules, start simplex and stopping test as fmin (xtol, ftol,
 * maxiter iterations, maxfun evaluations per region; maxfun = 0 means scipy's default of 200 per
 * dimension, which is what PyMC's MAP.fit leaves it at), so each region follows the path fmin would
 * take alone, but an iteration of all regions is one launch.  theta0 / theta_best: the regions'
 * D_r-vectors one after the other; active[n_regions] (may be NULL = all) selects the regions to
 * search, the others are returned unchanged.  A region whose search ends worse than its start
 * keeps the start.  lnprob_best[n_regions] and chi2_best (may be NULL) describe the returned
 * points; iterations (may be NULL) receives the simplex updates carried out per region (fmin
 * reports one more: it counts from 1, and like fmin the search stops at maxiter - 1 updates). */
int vamp_map_all(vamp_ctx* ctx, const double* theta0, const uint8_t* active, int64_t maxiter,
                 int64_t maxfun, double xtol, double ftol, double* theta_best, double* lnprob_best,
                 double* chi2_best, int64_t* iterations);

/* Per-component optical depths tau_comp[K, P] and model flux[P] for one parameter vector:
 * the `component_k` and `profile` deterministics (vpfits.py:254-260, 299-305, 334-336) whose
 * .value the callers read (vpspectrum.py:335,352-363).  Either output may be NULL. */
int vamp_model(vamp_ctx* ctx, int region, const double* theta1, double* tau_comp,
               double* flux_model);
/* The same for EVERY region in one launch (the fits of all regions of a spectrum end together:
 * vpspectrum.py:351-365 reads their values one after the other): theta = the regions' D_r-vectors
 * one after the other, tau_comp = their [K_r, P_r] blocks one after the other, flux_model laid out
 * like the pixels.  Either output may be NULL. */
int vamp_model_all(vamp_ctx* ctx, const double* theta, double* tau_comp, double* flux_model);

/* The per-line records the kernels stage in LDS for one parameter vector, rec[K*5] =
 * {centroid, x-scale, damping y, tau scale, pole factor} per component, and the log-prior: test
 * hook for the parameter maps of physics.py:6-27,116-134 / vpfits.py:79-88 and the priors of
 * vpfits.py:239-252,283-297. */
int vamp_line_records(vamp_ctx* ctx, int region, const double* theta1, double* rec, double* lnprior);

/* Device evaluation of Re w(x + i y) with the ctx's wofz_kind (test hook for the in-register
 * Faddeeva evaluator; the profile of vpfits.py:57-76 is built on it). */
int vamp_wofz_re(vamp_ctx* ctx, int64_t n, const double* x, const double* y, double* re_w);

/* ---- ensemble sampler: replaces VPfit.mcmc_fit / the MCMC calls of find_bic
 *      (vpfits.py:361-395, 420-425) with the stretch move of SURVEY Appendix B. ----
 * Every region of the ctx gets W walkers (W even, W % split_block == 0, split_block even).
 * theta0 is the concatenation over regions of [W, D_r] blocks.  Red/blue membership is a keyed
 * permutation inside chunks of split_block consecutive walkers, re-drawn every step; all draws
 * are Philox4x32-10 keyed by (seed, step, half, global walker id), so a trajectory does not
 * depend on how walkers are sharded over devices. */
int vamp_sampler_init(vamp_ctx* ctx, int64_t W, const double* theta0, uint64_t seed, double a,
                      int32_t split_block);
/* ---- walker-sharded runs of ONE region over several devices (one process per device) ----
 * The reference has no multi-device path (do_vamp.py:64-96 is an unfinished process pool over
 * files); this is the north star's "walkers shard over the GPUs of a node, one all-gather of
 * walker positions per stretch half-step".  Every device keeps the whole state X[W, D]; in a
 * half-step a device moves only its share of the active colour, and the rows those walkers end
 * the half-step with (the ACTIVE colour only: (W/2/world) x (D + 1) doubles per device, position +
 * lnprob) are all-gathered and scattered into every device's state before the other colour moves.
 *
 * vamp_comm_unique_id     rank 0 creates the 128-byte RCCL id and hands it to the other ranks by
 *                         any host channel (torch.distributed, MPI, a file ...)
 * vamp_comm_init_rank     collective over the `world` processes: one RCCL communicator per ctx,
 *                         on the ctx's device; the exchange then runs INSIDE
 *                         vamp_sampler_half_step / vamp_sampler_run(_dev) on the ctx's streams --
 *                         no host synchronisation, no torch in the data path.  (SURVEY 8b sketched
 *                         vamp_comm_init(ctxs, n_dev) for one process driving all devices; the
 *                         launch contract here is one process per GPU, hence the per-rank form.)
 * RCCL is loaded with dlopen on first use; single-device runs never touch it. */
int vamp_comm_unique_id(char id[VAMP_COMM_ID_BYTES]);
int vamp_comm_init_rank(vamp_ctx* ctx, const char id[VAMP_COMM_ID_BYTES], int rank, int world);
int vamp_comm_destroy(vamp_ctx* ctx);
/* What the communicator itself says it is: rank and size from ncclCommUserRank / ncclCommCount
 * (*queried = 1), or -- with a library that lacks those two entry points -- the values it was
 * created with (*queried = 0).  bench.py prints them as `rccl_ranks`, so that a scaling line proves
 * how many ranks the exchange really spanned.  VAMP_ERR_STATE without a communicator.  Any output
 * may be NULL.  vamp_comm_init_rank and vamp_sampler_set_shard_parts may come in either order;
 * the second one returns VAMP_ERR_ARG if its rank / world differ from the first one's. */
int vamp_comm_info(vamp_ctx* ctx, int* rank, int* world, int* queried);
/* File name of the shared library whose ncclAllGather carries the exchange (dladdr of the bound entry point),
 * copied into path[capacity]: librccl of the ROCm runtime this library is bound to -- or whatever VAMP_RCCL_LIB
 * named (the GPU tests' shared-memory stand-in).  bench.py prints it as `exchange.rccl_library`, so that a
 * scaling line says which library moved its bytes.  VAMP_ERR_COMM when no RCCL can be loaded. */
int vamp_comm_library(char* path, int64_t capacity);
/* Restrict this ctx to shard `rank` of `world` equal shards of every half-step's active slots;
 * own_begin / own_end receive the rows of the walkers it owns (whole split chunks).  n_accept is
 * maintained for owned rows only; positions and lnprob are complete on every rank. */
int vamp_sampler_set_shard(vamp_ctx* ctx, int rank, int world, int64_t* own_begin, int64_t* own_end);
/* The same with this rank's share cut into `parts` pieces that are stepped and exchanged one
 * after the other, so that the exchange of piece p (on the ctx's communication stream) overlaps
 * the kernel of piece p + 1: safe because a half-step kernel reads only rows of the frozen colour
 * and writes only its own piece.  The ensemble is first cut into `parts` equal slot ranges, each
 * of those into `world` shards.  own_begin / own_end: arrays of `parts` row bounds.  Needs
 * W / split_block to be a multiple of world * parts. */
int vamp_sampler_set_shard_parts(vamp_ctx* ctx, int rank, int world, int parts, int64_t* own_begin,
                                 int64_t* own_end);
/* Host-staged form of the exchange, for transports other than RCCL (the gloo tests; two ranks
 * sharing one device): after vamp_sampler_half_step_part(half, part), pack_get copies out this
 * rank's movers of that piece, rows[part_slots, D + 1]; the caller all-gathers them in rank order
 * and scatter_put writes rows_all[world * part_slots, D + 1] into the state (same kernels as the
 * RCCL path on both sides of the wire). */
int vamp_sampler_pack_get(vamp_ctx* ctx, int part, double* rows);
int vamp_sampler_scatter_put(vamp_ctx* ctx, int part, const double* rows_all);
/* Use caller-owned device memory for the walker state (X[total_theta] and lnp[total_walkers],
 * both double), e.g. torch tensors that take part in collectives.  Call before sampler_init. */
int vamp_sampler_bind_state(vamp_ctx* ctx, void* X_dev, void* lnp_dev);
int vamp_sampler_state_ptrs(vamp_ctx* ctx, void** X_dev, void** lnp_dev, int64_t* total_theta,
                            int64_t* total_walkers);
/* one half-step (half = 0 red moves, 1 blue moves), asynchronous on the ctx stream; the step
 * counter advances after half 1.  With a communicator it includes the exchange. */
int vamp_sampler_half_step(vamp_ctx* ctx, int half);
/* piece `part` of this rank's share only (vamp_sampler_set_shard_parts), WITHOUT exchange; the
 * step counter advances after the last piece of half 1.  vamp_sampler_half_step runs all pieces. */
int vamp_sampler_half_step_part(vamp_ctx* ctx, int half, int part);
/* same with every draw supplied by the host (deterministic-parity hook, single region):
 * active_idx[n], partner_idx[n] walker ids, zz[n] stretch factors, logu[n] = log(u2) */
int vamp_sampler_half_step_ext(vamp_ctx* ctx, int region, int64_t n, const int32_t* active_idx,
                               const int32_t* partner_idx, const double* zz, const double* logu);
/* n_steps full steps, keeping every thin-th sample.  chain[n_keep, total_theta],
 * lnprob_chain[n_keep, total_walkers], n_accept[total_walkers] (accepted proposals so far) may
 * be NULL.  seconds receives the wall time of the sampling loop (device-synchronised). */
int vamp_sampler_run(vamp_ctx* ctx, int64_t n_steps, int thin, double* chain, double* lnprob_chain,
                     int64_t* n_accept, double* seconds);
/* the same with the chain kept on the device: chain_dev / lnprob_chain_dev are DEVICE pointers
 * (caller-owned, same shapes, may be NULL).  Works on sharded contexts with a communicator: every
 * rank then records the complete state. */
int vamp_sampler_run_dev(vamp_ctx* ctx, int64_t n_steps, int thin, double* chain_dev,
                         double* lnprob_chain_dev, double* seconds);
int vamp_sampler_get_state(vamp_ctx* ctx, double* theta, double* lnprob, int64_t* n_accept,
                           int64_t* step);
int vamp_sampler_set_state(vamp_ctx* ctx, const double* theta, const double* lnprob, int64_t step);
/* HIP-event time (ms) and launch count of the half-step kernel since the last reset; used by
 * bench.py for the roofline line */
int vamp_kernel_timing(vamp_ctx* ctx, int enable, double* total_ms, int64_t* launches);
/* HIP-event time (ms) and count of the exchanges (ncclAllGather + scatter of one piece, on the stream
 * they ran on) recorded while vamp_kernel_timing was enabled, since the last reset; resets.  With
 * pieces > 1 the exchanges run on the communication stream beside the next piece's kernel, so
 * kernel time + exchange time may exceed the wall time: the difference is what the overlap hides. */
int vamp_exchange_timing(vamp_ctx* ctx, double* total_ms, int64_t* exchanges);

#ifdef __cplusplus
}
#endif
#endif /* VAMP_HIP_H */
