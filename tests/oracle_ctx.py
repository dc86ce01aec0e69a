"""Oracle-backed stand-in with the surface of ``vamp_amd.HipContext`` -- TEST INFRASTRUCTURE.

Lets the tests drive the product's host logic (``VPfit.find_bic``, ``chain_covariance``,
``VPregion.region_fit``, the walker-sharded driver) a second time with every posterior evaluation,
sampler step and MAP search carried out by the CPU oracle (oracle/vamp_oracle.py: numpy +
scipy.special.wofz, the restated stretch move with the counter-based draws, and scipy's own
``fmin`` for the MAP search that PyMC's ``MAP.fit`` runs, vpfits.py:352-358).  It lives under
tests/ and is injected by the tests only (``fit._ctx = OracleContext()``); nothing in vamp_amd/
can reach it.
"""
import time

import numpy as np
from scipy.optimize import fmin

from oracle import vamp_oracle as vo


class OracleContext:
    def __init__(self, device=0, dtype=0, wofz_kind=None):
        self.n_regions = 0
        self.ndims = []
        self.W = 0

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        pass

    def set_packing(self, lanes_per_walker):
        pass

    # -- data --------------------------------------------------------------------------------
    def set_regions(self, xs, fluxes, noises, n_comp, mode=vo.MODE_VOIGT4, sample_sd=False, include_norm=False,
                    bounds=None, nbz=None):
        if isinstance(xs, np.ndarray) and xs.ndim == 1:
            xs, fluxes, noises = [xs], [fluxes], [noises]
        R = len(xs)
        if np.isscalar(n_comp):
            n_comp = [int(n_comp)] * R
        self.regions = []
        for r in range(R):
            kw = {}
            if bounds is not None:
                b = np.asarray(bounds, dtype=np.float64).reshape(R, 4)[r]
                kw.update(c_lo=b[0], c_hi=b[1], sigma_max=b[2], fwhm_max=b[3])
            if nbz is not None:
                z = np.asarray(nbz, dtype=np.float64).reshape(R, 4)[r]
                kw.update(l_fixed=z[0], line=z[1], x_origin=z[2], x_scale=z[3])
            self.regions.append(vo.Region(x=xs[r], flux=fluxes[r], noise=noises[r], n_comp=int(n_comp[r]), mode=mode,
                                          sample_sd=bool(sample_sd), include_norm=bool(include_norm), **kw))
        self.n_regions = R
        self.mode = mode
        self.ndims = [reg.ndim for reg in self.regions]
        self.n_pix = [len(a) for a in xs]
        self.n_comp = [int(k) for k in n_comp]

    # -- evaluation --------------------------------------------------------------------------
    def lnprob(self, theta, region=0, return_chi2=False):
        theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
        out, chi = vo.log_prob_batch(self.regions[region], theta, return_chi2=True)
        return (out, chi) if return_chi2 else out

    def lnprob_all(self, thetas, return_chi2=False):
        res = [self.lnprob(t, region=r, return_chi2=True) for r, t in enumerate(thetas)]
        out, chi = np.stack([a for a, _ in res]), np.stack([b for _, b in res])
        return (out, chi) if return_chi2 else out

    def model(self, theta1, region=0):
        reg = self.regions[region]
        tau = vo.component_taus(reg, np.asarray(theta1, dtype=np.float64).ravel())
        with np.errstate(all="ignore"):
            return tau, vo.tau2flux(sum(list(tau)))

    def map_all(self, starts, iterlim=1000, tol=1e-3, active=None, xtol=1e-4, maxfun=0):
        """scipy.optimize.fmin on -lnprob (1e300 where it is not finite), region by region: what
        PyMC 2's ``MAP.fit(method='fmin', iterlim, tol)`` runs (fmin(..., maxiter=iterlim, ftol=tol),
        xtol and maxfun at scipy's defaults 1e-4 and 200 N -- from knowledge, pymc absent)."""
        best, lnp, chi, its = [], np.empty(self.n_regions), np.empty(self.n_regions), np.zeros(self.n_regions, dtype=np.int64)
        for r in range(self.n_regions):
            x0 = np.asarray(starts[r], dtype=np.float64).ravel()
            if active is not None and not active[r]:
                xb = x0.copy()
            else:
                def neg(t, r=r):
                    v = self.lnprob(t, region=r)[0]
                    return -v if np.isfinite(v) else 1e300
                xopt, fopt, it, calls, flag = fmin(neg, x0, xtol=xtol, ftol=tol, maxiter=iterlim, disp=False, full_output=True)
                its[r] = it - 1
                xb = xopt if fopt <= neg(x0) else x0.copy()
            l1, c1 = self.lnprob(xb, region=r, return_chi2=True)
            best.append(np.array(xb))
            lnp[r], chi[r] = l1[0], c1[0]
        return best, lnp, chi, its

    # -- sampler -----------------------------------------------------------------------------
    def sampler_init(self, theta0, seed=0, a=2.0, split_block=None):
        blocks = [theta0] if isinstance(theta0, np.ndarray) else list(theta0)
        self.W = blocks[0].shape[0]
        self.X = [np.array(b, dtype=np.float64) for b in blocks]
        self.fns = [(lambda q, reg=reg: vo.log_prob_batch(reg, q)) for reg in self.regions]
        self.lnp = [fn(X) for fn, X in zip(self.fns, self.X)]
        self.nacc = [np.zeros(self.W, dtype=np.int64) for _ in blocks]
        self.seed, self.a, self.split_block, self.step = int(seed), float(a), int(split_block), 0

    def run(self, n_steps, thin=1, store_chain=True):
        t0 = time.perf_counter()
        chains, lchains = [], []
        for r in range(self.n_regions):
            ch, lc, na = vo.run_sampler(self.fns[r], self.X[r], self.lnp[r], n_steps, seed=self.seed, block=self.split_block,
                                        a=self.a, step0=self.step, thin=1, region=r, walker_off=r * self.W)
            if n_steps:
                self.X[r], self.lnp[r] = ch[-1].copy(), lc[-1].copy()
            self.nacc[r] = self.nacc[r] + na
            keep = slice(thin - 1, (n_steps // thin) * thin, thin)
            chains.append(ch[keep] if n_steps else np.empty((0,) + self.X[r].shape))
            lchains.append(lc[keep] if n_steps else np.empty((0, self.W)))
        self.step += n_steps
        one = self.n_regions == 1
        res = {"seconds": time.perf_counter() - t0, "n_accept": self.nacc[0].copy() if one else [n.copy() for n in self.nacc]}
        if store_chain:
            res["chain"] = chains[0] if one else chains
            res["lnprob"] = lchains[0] if one else lchains
        return res

    def run_flat(self, n_steps, thin=1, store_chain=True):
        """HipContext.run_flat: the kept samples in the library's layout"""
        res = self.run(n_steps, thin=thin, store_chain=store_chain)
        nacc = np.concatenate([np.atleast_1d(n) for n in ([res["n_accept"]] if self.n_regions == 1 else res["n_accept"])])
        if not store_chain:
            return None, None, nacc, res["seconds"]
        ch = [res["chain"]] if self.n_regions == 1 else res["chain"]
        lc = [res["lnprob"]] if self.n_regions == 1 else res["lnprob"]
        n_keep = ch[0].shape[0]
        return (np.concatenate([c.reshape(n_keep, -1) for c in ch], axis=1), np.concatenate(lc, axis=1), nacc, res["seconds"])

    def model_all(self, thetas):
        res = [self.model(t, region=r) for r, t in enumerate(thetas)]
        return [np.array(list(a)) for a, _ in res], [b for _, b in res]

    def set_option(self, name, value):
        pass

    def get_state(self):
        if self.n_regions == 1:
            return self.X[0].copy(), self.lnp[0].copy(), self.nacc[0].copy(), self.step
        return [x.copy() for x in self.X], [l.copy() for l in self.lnp], [n.copy() for n in self.nacc], self.step
