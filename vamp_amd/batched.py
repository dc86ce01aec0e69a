"""Model selection for MANY regions at once (BASELINE.json config 3 end to end).

The reference fits the regions of a spectrum one after the other (vpspectrum.py:273-348), each by
the BIC ladder of ``VPregion.region_fit`` (vpregion.py:42-91) with three {MCMC, MAP} repeats per
rung (vpfits.py:417-428).  Regions are independent posteriors and so are the three repeats of a
rung, so here one rung of EVERY unfinished region -- all three repeats of it -- is ONE ragged context
of 3 R regions (repeat ``rep`` of region ``r`` is region ``rep * R + r``; its draws are keyed by that
index, so every repeat has a chain of its own): one ``sampler_init`` / ``run`` / ``map_all`` /
``model_all`` per rung instead of three of each, a third of the launches on a path that is
launch-bound.  The per-region decisions (BIC fell? reduced chi^2 under the limit?) are the
reference's, taken on the host from each region's own numbers.

The host side of a rung works on the ragged batch with array operations: the start points, the
walkers' start ball and their prior fall-backs are drawn for all regions of equal size together from
one generator per rung, and the only objects built per (region, repeat) are the numbers the ladder
compares.  A full ``VPfit`` -- ``.total.value``, ``estimated_variables``, ``mcmc.stats()``,
``map.BIC`` ... everything a caller reads from a reference fit object -- is built only for the fits
a ladder KEEPS (``RungResult.fit()``), from the last repeat's chain as in the reference
(vpfits.py:417-428 leaves the last model on ``self``).

What the batched path does not do: the DIC / BPIC of a chain (every kept sample scored once more)
are not part of the reference's model selection; ``find_bic_batched(score_chains=True)`` computes
them, the ladder leaves them ``None``.
"""
from __future__ import annotations

import os
import time

import numpy as np

from . import hip_backend as hb
from .vpfits import FWHM_PER_SIGMA, VPfit, _EnsembleMCMC, _MAP

MAX_COMPONENTS = 32      # VAMP_MAX_COMPONENTS of include/vamp_hip.h (the reference sets no limit, vpspectrum.py:287-294)
REPEATS = 3              # vpfits.py:417


class _DeferredModel:
    """Stands in for the context while a batched fit is set up: the start point's component / total
    values are never read (the MAP optimum replaces them), so no per-region ``vamp_model`` launch."""

    def __init__(self, n_pix):
        self.n_pix = n_pix

    def model(self, theta, region=0):
        return None, None


def _bound_fit(ctx, region_index, nu, flux, n, voigt, nwalkers, seed):
    fit = VPfit(seed=seed, dtype=getattr(ctx, "dtype", None))
    fit.nwalkers = nwalkers
    fit._region, fit._shared_ctx = region_index, True
    fit._ctx = _DeferredModel(flux.size)
    fit.initialise_model(nu, flux, n, voigt=voigt)
    fit._ctx = ctx
    return fit


class RungResult:
    """find_bic's outcome for one region: ``bic_array`` / ``red_chi_array`` of the three repeats, and what a
    ``VPfit`` of the last repeat needs, built on demand by ``fit()``."""

    __slots__ = ("bic_array", "red_chi_array", "n", "_ctx", "_index", "_region", "_voigt", "_W", "_seed", "_chain", "_lnp",
                 "_nacc", "_steps", "_keep", "_seconds", "_best", "_lnp_best", "_ssum_best", "_model", "_scored", "_fit")

    def detach(self):
        """own copies of the chain slices (they are views of the whole rung's chain, 3 R regions wide)"""
        self._chain, self._lnp = np.ascontiguousarray(self._chain), np.ascontiguousarray(self._lnp)
        return self

    def fit(self):
        if self._fit is None:
            nu, flux, _ = self._region
            f = _bound_fit(self._ctx, self._index, nu, flux, self.n, self._voigt, self._W, self._seed)
            f.map, f.mcmc = _MAP(f), _EnsembleMCMC(f)
            f._ingest_chain(self._chain, self._lnp, self._nacc, self._steps, self._keep, self._seconds,
                            scored=self._scored if self._scored is not None else "skip", set_values=False)
            f._map_finish(self._best, self._lnp_best, self._ssum_best, f.map, model=self._model)
            f.bic_array, f.red_chi_array = list(self.bic_array), list(self.red_chi_array)
            self._fit = f
        return self._fit


def _draw_prior_group(rng, x0, x1, n, q, voigt, sd, W):
    """[G, W, D] prior draws for G regions of n lines each (vpfits.py:239-252, 283-297: amplitude ~ x e^-x,
    centroid ~ U(x0, x1), widths ~ U(0, sigma_max | fwhm_max), sd ~ U(0, 1)); the twin of VPfit._draw_prior"""
    G = x0.size
    wmax = (x1 - x0) / 2.0 * (FWHM_PER_SIGMA if voigt else 1.0)
    th = np.empty((G, W, q * n + (1 if sd else 0)))
    lines = th[:, :, :q * n].reshape(G, W, n, q)
    lines[..., 0] = rng.gamma(2.0, 1.0, (G, W, n))
    lines[..., 1] = x0[:, None, None] + (x1 - x0)[:, None, None] * rng.random((G, W, n))
    for j in range(2, q):
        lines[..., j] = wmax[:, None, None] * rng.random((G, W, n))
    if sd:
        th[:, :, -1] = rng.random((G, W))
    return th


def find_bic_batched(ctx, regions, ns, voigt=False, nwalkers=64, iterations=3000, thin=15, burn=300, seed=0, freedoms=None,
                     score_chains=False):
    """vpfits.find_bic (vpfits.py:398-429) for a list of regions [(nu, flux, noise), ...] with ns[r]
    components each: the three repeats of every region in ONE context.  Returns one ``RungResult`` per region."""
    R = len(regions)
    J = REPEATS * R
    q = 4 if voigt else 3
    ns = [int(n) for n in ns]
    W = max(int(nwalkers), 2 * (q * max(ns) + 1) + 2)
    W += W % 2
    thin = max(1, thin)
    keep = max(thin, iterations - burn)
    rng = np.random.default_rng([int(seed) & 0xFFFFFFFF, 0x5EED])
    # region-centred pixel coordinates (vamp_amd.vpfits: SURVEY section 7 "fp32 coordinate cancellation")
    xs1, P = [], np.empty(R, dtype=np.int64)
    for r, (nu, flux, noise) in enumerate(regions):
        mid, dnu = 0.5 * (nu[0] + nu[-1]), (nu[-1] - nu[0]) / (nu.size - 1)
        xs1.append((nu - mid) / dnu)
        P[r] = flux.size
    fs1 = [reg[1] for reg in regions]
    ones1 = [np.ones_like(f) for f in fs1]
    # the reference builds VPfit() without noise: free precision sd ~ U(0,1) (vpregion.py:59, vpfits.py:39)
    ctx.set_regions(xs1 * REPEATS, fs1 * REPEATS, ones1 * REPEATS, ns * REPEATS, mode=hb.MODE_VOIGT4 if voigt else hb.MODE_GAUSS3,
                    sample_sd=True)
    x_lo = np.array([x[0] for x in xs1] * REPEATS)
    x_hi = np.array([x[-1] for x in xs1] * REPEATS)
    n_of = np.array(ns * REPEATS)
    # start points (amplitude 0.5, vpfits.py:240, the rest from the priors like PyMC), the walkers' ball around them and
    # one prior draw per walker to fall back on -- VPfit.initialise_model / _draw_walkers for every (region, repeat) of
    # equal size together
    centre, X0, fallback = [None] * J, [None] * J, [None] * J
    for n in sorted(set(ns)):
        idx = np.nonzero(n_of == n)[0]
        c = _draw_prior_group(rng, x_lo[idx], x_hi[idx], n, q, voigt, True, 1)[:, 0, :]
        c[:, 0:q * n:q] = 0.5
        span = np.abs(_draw_prior_group(rng, x_lo[idx], x_hi[idx], n, q, voigt, True, W) - c[:, None, :])
        ball = c[:, None, :] + 1e-2 * span * rng.standard_normal(span.shape)
        prior = _draw_prior_group(rng, x_lo[idx], x_hi[idx], n, q, voigt, True, W)
        for g, j in enumerate(idx):
            centre[j], X0[j], fallback[j] = c[g], ball[g], prior[g]
    lnp0 = ctx.lnprob_all(X0)                                     # one launch for every walker of every region
    for j in range(J):
        bad = ~np.isfinite(lnp0[j])
        if bad.any():
            X0[j][bad] = fallback[j][bad]                        # a poor start cannot trap the ensemble
        X0[j][0] = centre[j]
    ctx.sampler_init(X0, seed=(int(seed) * 2654435761 + 0x9E37) & (2 ** 64 - 1), a=2.0, split_block=hb.default_split_block(W))
    t_run = time.perf_counter()
    if burn > 0:
        ctx.run(burn, store_chain=False)
    chain2d, lnp2d, nacc1d, seconds = ctx.run_flat(keep, thin=thin)
    t_run = time.perf_counter() - t_run
    n_keep = chain2d.shape[0]
    offs = np.concatenate([[0], np.cumsum([W * d for d in ctx.ndims])])
    chains = [chain2d[:, offs[j]:offs[j + 1]].reshape(n_keep, W, ctx.ndims[j]) for j in range(J)]
    lnp3 = lnp2d.reshape(n_keep, J, W)
    # MAP polish of every (region, repeat) together, started from its best posterior sample (VPfit._map_start)
    flat_best = lnp3.transpose(1, 0, 2).reshape(J, n_keep * W).argmax(axis=1)
    starts = [chains[j][flat_best[j] // W, flat_best[j] % W] for j in range(J)]
    t_map = time.perf_counter()
    best, lnp_best, ssum_best, its = ctx.map_all(starts, iterlim=iterations, tol=1e-3)
    t_map = time.perf_counter() - t_map
    if os.environ.get("VAMP_FIT_TIMING"):       # developer knob: where a rung's time goes
        kinds = ctx.region_classes()[0] if hasattr(ctx, "region_classes") else []
        print("vamp_rung regions=%d (x%d repeats) W=%d steps=%d run=%.1f ms (%.1f us per half-step) map=%.1f ms (max %d iterations) classes=%s"
              % (R, REPEATS, W, burn + keep, t_run * 1e3, t_run * 1e6 / (2 * (burn + keep)), t_map * 1e3, int(np.max(its)),
                 {k: kinds.count(k) for k in sorted(set(kinds))}), flush=True)
    taus, fluxes = ctx.model_all(best)
    # PyMC 2.3's information criteria at the optimum (vamp_amd.vpfits._MAP): lnL of the observed flux with the free precision
    sd = np.array([b[-1] for b in best])
    Pj = np.tile(P, REPEATS)
    with np.errstate(all="ignore"):
        t = 1.0 / sd ** 2
        lnL = np.where(np.isfinite(lnp_best), Pj * 0.5 * np.log(t / (2.0 * np.pi)) - 0.5 * t * ssum_best, -np.inf)
    kfree = np.array(ctx.ndims, dtype=np.float64)
    bic = kfree * np.log(Pj) - 2.0 * lnL
    scored = None
    if score_chains:       # DIC / BPIC of the chains: every kept sample and the mean points, one launch each
        flats = [ch.reshape(-1, ch.shape[2]) for ch in chains]
        lnp_s, ss_s = ctx.lnprob_all(flats, return_chi2=True)
        lnp_m, ss_m = ctx.lnprob_all([fl.mean(0)[None, :] for fl in flats], return_chi2=True)
        scored = [((lnp_s[j], ss_s[j]), (lnp_m[j, 0], ss_m[j, 0])) for j in range(J)]
    out = []
    for r, (nu, flux, noise) in enumerate(regions):
        freedom = freedoms[r] if freedoms is not None else flux.size - 3 * ns[r]
        res = RungResult()
        res.n = ns[r]
        res.bic_array = [float(bic[rep * R + r]) for rep in range(REPEATS)]
        res.red_chi_array = [VPfit.ReducedChisquared(flux, fluxes[rep * R + r], noise, freedom) for rep in range(REPEATS)]
        j = (REPEATS - 1) * R + r                                 # the last repeat is the model the reference leaves on `self`
        res._ctx, res._index, res._region, res._voigt, res._W = ctx, j, regions[r], voigt, W
        res._seed = (int(seed) * 1000003 + 7919 * r) & 0x7FFFFFFFFFFFFFFF
        res._chain, res._lnp, res._nacc = chains[j], lnp3[:, j, :], nacc1d[j * W:(j + 1) * W]
        res._steps, res._keep, res._seconds = burn + keep, keep, seconds
        res._best, res._lnp_best, res._ssum_best, res._model = best[j], lnp_best[j], ssum_best[j], (taus[j], fluxes[j])
        res._scored = None if scored is None else scored[j]
        res._fit = None
        out.append(res)
    return out


class BatchedRegionLadder:
    """``VPregion.region_fit`` for a list of ``VPregion`` objects, all rungs of all regions batched.
    After ``run()`` every region has ``.fit`` and ``.n`` as after the reference's ``region_fit``."""

    def __init__(self, regions, nwalkers=64, iterations=3000, thin=15, burn=300, seed=0, device=0, verbose=True, ctx=None,
                 dtype=None):
        self.regions = list(regions)
        self.nwalkers, self.iterations, self.thin, self.burn = nwalkers, iterations, thin, burn
        self.seed, self.verbose = seed, verbose
        self.ctx = ctx if ctx is not None else hb.HipContext(device=device, dtype=dtype)

    def _rung(self, idx, rung):
        regs = [self.regions[i] for i in idx]
        data = [(r.frequency_array, r.flux_array, r.noise_array) for r in regs]
        return find_bic_batched(self.ctx, data, [r.n for r in regs], voigt=regs[0].voigt, nwalkers=self.nwalkers,
                                iterations=self.iterations, thin=self.thin, burn=self.burn,
                                seed=self.seed + 1000003 * rung, freedoms=[r.freedom for r in regs])

    def run(self):
        say = print if self.verbose else (lambda *a, **k: None)
        active = list(range(len(self.regions)))
        kept, kept_bic = {}, {}
        say("Batched BIC ladder over {} regions.".format(len(active)))
        for i, f in zip(active, self._rung(active, 0)):       # first rung: judged by its LAST BIC (vpregion.py:63)
            kept[i], kept_bic[i] = f.detach(), f.bic_array[-1]
        rung = 1
        while active:
            nxt = []
            for i in active:
                if self.regions[i].n >= MAX_COMPONENTS:
                    continue
                self.regions[i].n += 1
                nxt.append(i)
            active = nxt
            if not active:
                break
            say("rung {}: {} regions still adding components".format(rung, len(active)))
            still = []
            for i, f in zip(active, self._rung(active, rung)):
                reg = self.regions[i]
                bic = float(np.average(f.bic_array))
                if not (kept_bic[i] > bic):
                    reg.n -= 1                                 # BIC rose: keep the previous rung
                    continue
                kept[i], kept_bic[i] = f.detach(), bic
                if np.average(f.red_chi_array) < reg.chi_limit:
                    continue                                   # good enough: stop at this n
                still.append(i)
            active = still
            rung += 1
        for i, reg in enumerate(self.regions):
            reg.fit = kept[i].fit()                            # the one full fit object per region: the rung it keeps
            reg.n = len(reg.fit.estimated_profiles)
        return self.regions
