"""Model selection for MANY regions at once (BASELINE.json config 3 end to end).

The reference fits the regions of a spectrum one after the other (vpspectrum.py:273-348), each by
the BIC ladder of ``VPregion.region_fit`` (vpregion.py:42-91) with three {MCMC, MAP} repeats per
rung (vpfits.py:417-428).  Regions are independent posteriors, so here one rung of EVERY unfinished
region is sampled together: the regions are uploaded as one ragged batch, each with its own number
of components, and a half-step of all their ensembles is ONE kernel launch
(``vamp_sampler_half_step`` over ``n_regions * W/2`` walkers; four walkers per wavefront for short
regions).  The per-region decisions (BIC fell? reduced chi^2 under the limit?) are the
reference's, taken on the host from each region's own chain.

Each region is represented by a ``VPfit`` bound to (shared context, region index), so everything a
caller reads from a reference fit object -- ``.total.value``, ``estimated_variables``,
``mcmc.stats()``, ``map.BIC`` ... -- is there.
"""
from __future__ import annotations

from copy import copy

import numpy as np

from . import hip_backend as hb
from .vpfits import VPfit, _EnsembleMCMC, _MAP

MAX_COMPONENTS = 32      # VAMP_MAX_COMPONENTS of include/vamp_hip.h (the reference sets no limit, vpspectrum.py:287-294)


class _DeferredModel:
    """Stands in for the context while a batched fit is set up: the start point's component / total
    values are never read (the ensemble run replaces them), so no per-region ``vamp_model`` launch."""

    def __init__(self, n_pix):
        self.n_pix = n_pix

    def model(self, theta, region=0):
        return None, None


def _bound_fit(ctx, region_index, nu, flux, n, voigt, nwalkers, seed):
    fit = VPfit(seed=seed)
    fit.nwalkers = nwalkers
    fit._region, fit._shared_ctx = region_index, True
    fit._ctx = _DeferredModel(flux.size)
    fit._set_values_deferred = True
    fit.initialise_model(nu, flux, n, voigt=voigt)
    fit._ctx = ctx
    return fit


def find_bic_batched(ctx, regions, ns, voigt=False, nwalkers=64, iterations=3000, thin=15, burn=300, seed=0, freedoms=None):
    """vpfits.find_bic (vpfits.py:398-429) for a list of regions [(nu, flux, noise), ...] with ns[r]
    components each.  Returns one VPfit per region carrying ``bic_array`` / ``red_chi_array`` of
    the three repeats and the state of the last one."""
    R = len(regions)
    fits = [None] * R
    bics = [[] for _ in range(R)]
    chis = [[] for _ in range(R)]
    W = max(int(nwalkers), 2 * ((4 if voigt else 3) * max(ns) + 1) + 2)
    W += W % 2
    thin = max(1, thin)
    keep = max(thin, iterations - burn)
    for rep in range(3):
        xs, fs = [], []
        for (nu, flux, noise) in regions:
            mid, dnu = 0.5 * (nu[0] + nu[-1]), (nu[-1] - nu[0]) / (nu.size - 1)
            xs.append((nu - mid) / dnu)
            fs.append(flux)
        # the reference builds VPfit() without noise: free precision sd ~ U(0,1) (vpregion.py:59, vpfits.py:39)
        ctx.set_regions(xs, fs, [np.ones_like(f) for f in fs], list(ns), mode=hb.MODE_VOIGT4 if voigt else hb.MODE_GAUSS3,
                        sample_sd=True)
        cur = [_bound_fit(ctx, r, regions[r][0], regions[r][1], ns[r], voigt, W, seed + 7919 * rep + 104729 * r)
               for r in range(R)]
        for f in cur:
            f.map, f.mcmc = _MAP(f), _EnsembleMCMC(f)
        drawn = [f._draw_walkers() for f in cur]
        lnp0 = ctx.lnprob_all([d[0] for d in drawn])              # one launch instead of one per region
        X0 = [f._finish_walkers(d[0], d[1], lnp0[r]) for r, (f, d) in enumerate(zip(cur, drawn))]
        ctx.sampler_init(X0, seed=(seed * 2654435761 + rep) & (2 ** 64 - 1), a=2.0, split_block=hb.default_split_block(W))
        if burn > 0:
            ctx.run(burn, store_chain=False)
        res = ctx.run(keep, thin=thin)
        chains = res["chain"] if R > 1 else [res["chain"]]
        lnps = res["lnprob"] if R > 1 else [res["lnprob"]]
        naccs = res["n_accept"] if R > 1 else [res["n_accept"]]
        # every kept sample of every region scored in ONE launch (DIC / BPIC), the mean points in another
        flats = [chains[r].reshape(-1, cur[r]._ndim) for r in range(R)]
        lnp_s, ss_s = ctx.lnprob_all(flats, return_chi2=True)
        lnp_m, ss_m = ctx.lnprob_all([fl.mean(0)[None, :] for fl in flats], return_chi2=True)
        for r, f in enumerate(cur):
            f._ingest_chain(chains[r], lnps[r], naccs[r], burn + keep, keep, res["seconds"],
                            scored=((lnp_s[r], ss_s[r]), (lnp_m[r, 0], ss_m[r, 0])), set_values=False)
        # the MAP polish of every region together: one launch per Nelder-Mead iteration; then the
        # component and total values of all optima from one launch (vamp_model_all)
        best, lnp_best, ssum_best, _ = ctx.map_all([f._map_start() for f in cur], iterlim=iterations, tol=1e-3)
        taus, fluxes = ctx.model_all(best)
        for r, f in enumerate(cur):
            f._map_finish(best[r], lnp_best[r], ssum_best[r], f.map, model=(taus[r], fluxes[r]))
            nu, flux, noise = regions[r]
            freedom = freedoms[r] if freedoms is not None else flux.size - 3 * ns[r]
            bics[r].append(f.map.BIC)
            chis[r].append(f.ReducedChisquared(flux, f.total.value, noise, freedom))
        fits = cur
    for r, f in enumerate(fits):
        f.bic_array, f.red_chi_array = bics[r], chis[r]
    return fits


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
            kept[i], kept_bic[i] = f, f.bic_array[-1]
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
                kept[i], kept_bic[i] = copy(f), bic
                if np.average(f.red_chi_array) < reg.chi_limit:
                    continue                                   # good enough: stop at this n
                still.append(i)
            active = still
            rung += 1
        for i, reg in enumerate(self.regions):
            reg.fit = kept[i]
            reg.n = len(kept[i].estimated_profiles)
        return self.regions
