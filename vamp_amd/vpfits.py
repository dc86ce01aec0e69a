"""``VPfit`` -- the reference's fit object (vamp_1.0/vpfits.py:33) on the MI355X hot path.

Same constructor, method names, argument meaning and public attributes as the reference class, so
``VPregion`` / ``VPspectrum`` / notebook code written against it runs unchanged; what differs is
what sits underneath:

  reference                                            here
  -------------------------------------------------    ------------------------------------------
  PyMC 2 model graph evaluated per proposal            HIP kernel: one wavefront per walker
  (vpfits.py:239-260, 283-305, 334-341)                (csrc/vamp_hip.hip, k_lnprob / k_half_step)
  one-at-a-time Metropolis, mc.MCMC.sample             affine-invariant stretch move, W walkers
  (vpfits.py:379-393, 420-425)                         (vamp_sampler_*), counter-based RNG
  mc.MAP.fit = scipy fmin on -logp                     fmin's Nelder-Mead on -logp in the library, every
  (vpfits.py:357-358, 419-426)                         evaluation a device call, started from the
                                                       best walker
  PyMC nodes (.value), mcmc.trace / stats, map.BIC     light stand-ins with the same attributes

Sampler knobs keep their names: ``iterations`` = ensemble steps, ``burn``/``burnin`` = steps
discarded, ``thin``/``thinning`` = keep every n-th step; each kept step contributes ``nwalkers``
samples per parameter.  ``.value`` of every node is left at the MAP optimum after ``map.fit`` /
``find_bic``, as PyMC leaves them (vpfits.py:426).

Coordinates: the device works in region-centred units x = (nu - nu_mid)/dnu (dnu = mean pixel
spacing), which keeps |x| small in fp64 and makes fp32 possible; every value handed back
(traces, .value, sigma_max, ...) is in the caller's units (Hz).

There is no CPU fallback: constructing the model needs libvamp_hip.so and a GPU.
"""
from __future__ import annotations

import datetime
from collections.abc import Mapping

import numpy as np

from . import hip_backend as hb
from .physics import *  # noqa: F401,F403  (the reference module does `from physics import *`)
from .physics import Tau2flux

FWHM_PER_SIGMA = 2.0 * np.sqrt(2.0 * np.log(2.0))


class _Node:
    """Stand-in for a PyMC node: callers only read ``.value`` (and ``__name__``)."""

    def __init__(self, name, value=None):
        self.__name__ = name
        self.value = value

    def __repr__(self):
        return f"<node {self.__name__} = {self.value!r}>"


class _Trace:
    def __init__(self, samples):
        self._s = samples

    def __getitem__(self, idx):
        return self._s[idx]

    def __len__(self):
        return len(self._s)

    def gettrace(self, burn=0, thin=1):
        return self._s[burn::thin]


class _EnsembleMCMC:
    """What callers use of ``mc.MCMC``: trace(name)[:], stats()[name]['standard deviation'],
    DIC / BPIC (PyMC 2 definitions: deviance D = -2 log L; DIC = 2 mean(D) - D(mean theta),
    BPIC = 3 mean(D) - 2 D(mean theta))."""

    def __init__(self, fit):
        self._fit = fit
        self._flat = None          # [samples, D] in the caller's units; traces are columns, cut on demand
        self._names = []
        self._derived = {}         # name -> (source column name, function): est_sigma_k in Voigt mode
        self.DIC = None
        self.BPIC = None
        self.acceptance_fraction = None
        self.walker_steps_per_second = None

    def sample(self, iter, burn=0, thin=1, progress_bar=False, **_ignored):
        self._fit._run_sampler(int(iter), int(burn), int(thin))

    def _samples(self, name):
        if name in self._derived:
            src, fn = self._derived[name]
            return fn(self._samples(src))
        return self._flat[:, self._names.index(name)].copy()

    @property
    def _traces(self):
        """name -> samples for every trace (materialised on demand: a batched model-selection ladder
        builds thousands of fits whose traces are never read)"""
        return {nm: self._samples(nm) for nm in list(self._names) + list(self._derived)}

    def trace(self, name):
        if name not in self._names and name not in self._derived:
            raise KeyError(name)
        return _Trace(self._samples(name))

    def stats(self):
        """name -> {'n', 'standard deviation', 'mean', 'quantiles', 'mc error'} as PyMC's ``MCMC.stats()``; an entry is
        computed when it is read (the harvest of a spectrum reads two numbers per line, vpspectrum.py:400-412)."""
        return _Stats(self)


class _Stats(Mapping):
    """``mcmc.stats()``: a read-only mapping over every trace name whose entries are computed on first access"""

    def __init__(self, mcmc):
        self._mcmc = mcmc
        self._names = list(mcmc._names) + list(mcmc._derived)
        self._done = {}

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)

    def __getitem__(self, name):
        if name not in self._done:
            if name not in self._names:
                raise KeyError(name)
            s = self._mcmc._samples(name)
            q = np.percentile(s, [2.5, 25, 50, 75, 97.5])
            sd = float(np.std(s))
            self._done[name] = {"n": s.size, "standard deviation": sd, "mean": float(np.mean(s)),
                                "quantiles": {2.5: q[0], 25: q[1], 50: q[2], 75: q[3], 97.5: q[4]},
                                "mc error": sd / np.sqrt(max(1, s.size))}
        return self._done[name]


class _MAP:
    """What callers use of ``mc.MAP``: fit(iterlim, tol), BIC, AIC, lnL, logp_at_max (PyMC 2.3:
    BIC = k ln(n_data) - 2 lnL, AIC = 2 k - 2 lnL, k = number of free scalars, lnL = log-likelihood
    of the observed data at the optimum)."""

    def __init__(self, fit):
        self._fit = fit
        self.BIC = self.AIC = self.lnL = self.logp_at_max = None
        self.len = fit._ndim
        self.data_len = fit._flux.size

    def fit(self, iterlim=1000, tol=1e-3, **_ignored):
        self._fit._run_map(int(iterlim), float(tol), self)


class VPfit():

    # ensemble size used by mcmc_fit / find_bic; may be changed per instance
    nwalkers = 64

    def __init__(self, noise=None, device=0, dtype=None, seed=None):
        """noise=None reproduces the reference's free precision ``sd ~ U(0,1)`` (vpfits.py:39);
        an array of per-pixel sigmas gives the known-noise chi^2 likelihood."""
        self.noise = None if noise is None else np.asarray(noise, dtype=np.float64)
        self.std_deviation = None if noise is None else 1.0 / self.noise ** 2
        self.verbose = False
        self.device = device
        self.dtype = hb.resolve_dtype(dtype)      # None: $VAMP_DTYPE, else fp64 (fp32 = Humlicek W4, BASELINE config 5)
        self._seed = np.random.SeedSequence(seed).generate_state(1, dtype=np.uint64)[0] if seed is not None \
            else np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0]
        self._ctx = None
        self._region = 0          # index of this fit's region inside the context (batched fits share one)
        self._shared_ctx = False

    # ---- statics (vpfits.py:43-131), host numpy as in the reference -------------------------
    @staticmethod
    def GaussFunction(x, amplitude, centroid, sigma):
        """Gaussian tau-profile (vpfits.py:43-54)"""
        return amplitude * np.exp(-0.5 * ((x - centroid) / sigma) ** 2)

    @staticmethod
    def VoigtFunction(x, centroid, amplitude, L_fwhm, G_fwhm):
        """Voigt tau-profile (vpfits.py:57-76), evaluated by the HIP Faddeeva code."""
        x = np.asarray(x, dtype=np.float64)
        order = np.argsort(x)
        xs = x[order]
        mid, dx = 0.5 * (xs[0] + xs[-1]), (xs[-1] - xs[0]) / max(1, xs.size - 1)
        dx = dx if dx > 0 else 1.0
        with hb.HipContext() as ctx:
            ctx.set_regions((xs - mid) / dx, np.ones_like(xs), np.ones_like(xs), 1, mode=hb.MODE_VOIGT4)
            tau, _ = ctx.model(np.array([amplitude, (centroid - mid) / dx, L_fwhm / dx, G_fwhm / dx]))
        out = np.empty_like(x)
        out[order] = tau[0]
        return out

    @staticmethod
    def GaussianWidth(G_fwhm):
        """sigma of the Gaussian part of a Voigt profile from its FWHM (vpfits.py:79-88)"""
        return G_fwhm / (2. * np.sqrt(2. * np.log(2.)))

    @staticmethod
    def Chisquared(observed, expected, noise):
        """chi^2 (vpfits.py:109-118)"""
        return sum(((observed - expected) / noise) ** 2)

    @staticmethod
    def ReducedChisquared(observed, expected, noise, freedom):
        """chi^2 / dof (vpfits.py:121-131)"""
        return VPfit.Chisquared(observed, expected, noise) / freedom

    # ---- model ------------------------------------------------------------------------------
    def initialise_model(self, frequency, flux, n, local_minima=[], voigt=False):
        """Build the posterior for ``n`` components (vpfits.py:310-349): upload the region once,
        create the node stand-ins.  Prior bounds as in the reference: centroid ~ U(nu[0], nu[-1]),
        sigma ~ U(0, sigma_max), L/G ~ U(0, fwhm_max), amplitude ~ x e^-x, sd ~ U(0,1)."""
        frequency = np.asarray(frequency, dtype=np.float64)
        flux = np.asarray(flux, dtype=np.float64)
        self.sigma_max = (frequency[-1] - frequency[0]) / 2.
        self._voigt = bool(voigt)
        self._n = int(n)
        self._freq, self._flux = frequency, flux
        self._mid = 0.5 * (frequency[0] + frequency[-1])
        self._dnu = (frequency[-1] - frequency[0]) / (frequency.size - 1)
        self._x = (frequency - self._mid) / self._dnu
        self._sample_sd = self.noise is None
        if voigt:
            if self.verbose:
                print("Initialising Voigt profile components.")
            self.fwhm_max = self.sigma_max * 2 * np.sqrt(2 * np.log(2.))
        elif self.verbose:
            print("Initialising Gaussian profile components.")
        self._mode = hb.MODE_VOIGT4 if voigt else hb.MODE_GAUSS3
        self._q = 4 if voigt else 3
        self._ndim = self._q * self._n + (1 if self._sample_sd else 0)
        if not self._shared_ctx:      # a batched fit's region is uploaded by vamp_amd.batched
            noise = np.ones_like(flux) if self._sample_sd else self.noise
            if self._ctx is None:
                self._ctx = hb.HipContext(device=self.device, dtype=self.dtype)
            self._ctx.set_regions(self._x, flux, noise, self._n, mode=self._mode, sample_sd=self._sample_sd)

        # parameter names and unit scales (device units -> caller units)
        names, keys, scale, shift = [], [], [], []
        for k in range(self._n):
            if voigt:
                spec = (("xexp_%d", "amplitude", 1.0, 0.0), ("est_centroid_%d", "centroid", self._dnu, self._mid),
                        ("est_L_%d", "L_fwhm", self._dnu, 0.0), ("est_G_%d", "G_fwhm", self._dnu, 0.0))
            else:
                spec = (("xexp_%d", "amplitude", 1.0, 0.0), ("est_centroid_%d", "centroid", self._dnu, self._mid),
                        ("est_sigma_%d", "sigma", self._dnu, 0.0))
            for nm, key, sc, sh in spec:
                names.append(nm % k); keys.append((k, key)); scale.append(sc); shift.append(sh)
        if self._sample_sd:
            names.append("sd"); keys.append((None, "sd")); scale.append(1.0); shift.append(0.0)
        self._names, self._keys = names, keys
        self._scale, self._shift = np.array(scale), np.array(shift)

        self.estimated_variables = {}
        self.estimated_profiles = []
        for k in range(self._n):
            self.estimated_variables[k] = {}
        for nm, (k, key) in zip(names, keys):
            node = _Node(nm)
            if k is None:
                self._sd_node = node
            else:
                self.estimated_variables[k][key] = node
        for k in range(self._n):
            self.estimated_profiles.append(_Node("component_%d" % k))
        self.total = _Node("profile")
        self.profile = _Node("obs", flux)
        self.model = [self.estimated_variables[k][key] for k in self.estimated_variables for key in self.estimated_variables[k]]

        # initial point: amplitude 0.5 (vpfits.py:240), the rest drawn from their priors like PyMC
        rng = np.random.default_rng(int(self._seed) & 0xFFFFFFFF)
        self._theta_dev = self._draw_prior(rng, 1)[0]
        self._theta_dev[0::self._q][:self._n] = 0.5
        self._chain_dev = None
        self._lnp_chain = None
        self._set_values(self._theta_dev)

    # -- helpers ---------------------------------------------------------------------------
    def _draw_prior(self, rng, W, local_minima=()):
        x0, x1 = self._x[0], self._x[-1]
        wmax = (x1 - x0) / 2. * (FWHM_PER_SIGMA if self._voigt else 1.0)
        th = np.empty((W, self._ndim))
        for k in range(self._n):
            o = self._q * k
            th[:, o] = rng.gamma(2.0, 1.0, W)                      # density x e^-x
            th[:, o + 1] = rng.uniform(x0, x1, W)
            for j in range(2, self._q):
                th[:, o + j] = rng.uniform(0, wmax, W)
        if self._sample_sd:
            th[:, -1] = rng.uniform(0, 1, W)
        return th

    def _to_caller(self, theta_dev):
        return theta_dev * self._scale + self._shift

    def _set_values(self, theta_dev, model=None):
        """Put one parameter vector on the nodes: .value of every variable, component and total.
        ``model`` = (tau[K, P], flux[P]) when the caller already has them (batched fits evaluate
        the models of all regions in one launch), else one ``vamp_model`` call."""
        theta_dev = np.asarray(theta_dev, dtype=np.float64)
        self._theta_dev = theta_dev.copy()
        vals = self._to_caller(theta_dev)
        for nm, (k, key), v in zip(self._names, self._keys, vals):
            if k is None:
                self._sd_node.value = float(v)
            else:
                self.estimated_variables[k][key].value = float(v)
        tau, flux = model if model is not None else self._ctx.model(theta_dev, region=self._region)
        if tau is None:          # a batched fit being set up: values follow from the first batched evaluation
            return
        for k in range(self._n):
            self.estimated_profiles[k].value = tau[k].copy()
        self.total.value = flux

    def _loglike(self, theta_dev):
        """log-likelihood of the observed flux (device chi^2 / SSR, host epilogue)"""
        lnp, s = self._ctx.lnprob(theta_dev, region=self._region, return_chi2=True)
        return self._loglike_from_sum(theta_dev, lnp, s)

    def _loglike_from_sum(self, theta_dev, lnp, s):
        th = np.atleast_2d(theta_dev)
        if self._sample_sd:
            t = 1.0 / th[:, -1] ** 2
            ll = self._flux.size * 0.5 * np.log(t / (2 * np.pi)) - 0.5 * t * s
        else:
            ll = -0.5 * s - 0.5 * np.sum(np.log(2 * np.pi * self.noise ** 2))
        return np.where(np.isfinite(lnp), ll, -np.inf)

    # -- MAP ---------------------------------------------------------------------------------
    def map_estimate(self, iterations=2000):
        """Maximum a posteriori estimate (vpfits.py:352-358)."""
        self.map = _MAP(self)
        self.map.fit(iterlim=iterations, tol=1e-3)

    def _map_start(self):
        """Start of the MAP search: the best posterior sample of the ensemble if there is one
        (polish), else the current parameter values."""
        if self._lnp_chain is not None:
            i = np.unravel_index(np.argmax(self._lnp_chain), self._lnp_chain.shape)
            return self._chain_dev[i[0], i[1]]
        return self._theta_dev

    def _map_finish(self, best, lnp_best, ssum_best, mp, model=None):
        """Adopt the optimum and fill the MAP object (PyMC 2.3 definitions, see _MAP)."""
        self._set_values(best, model=model)
        mp.logp_at_max = float(lnp_best)
        mp.lnL = float(np.ravel(self._loglike_from_sum(best, lnp_best, ssum_best))[0])
        k, n = self._ndim, self._flux.size
        mp.len, mp.data_len = k, n
        mp.BIC = k * np.log(n) - 2.0 * mp.lnL
        mp.AIC = 2.0 * k - 2.0 * mp.lnL

    def _run_map(self, iterlim, tol, mp):
        """Nelder-Mead on -logp in the library (vamp_map_all: scipy fmin's rules, every candidate
        point of an iteration in one launch); only this fit's region of a shared context moves."""
        ctx, reg = self._ctx, self._region
        starts = [np.zeros(d) for d in ctx.ndims]
        starts[reg] = np.asarray(self._map_start(), dtype=np.float64)
        active = np.zeros(ctx.n_regions, dtype=np.uint8)
        active[reg] = 1
        best, lnp, ssum, _ = ctx.map_all(starts, iterlim=iterlim, tol=tol, active=active)
        self._map_finish(best[reg], lnp[reg], ssum[reg], mp)

    # -- MCMC --------------------------------------------------------------------------------
    def mcmc_fit(self, iterations=15000, burnin=100, thinning=15, step_method=None):
        """Sample the posterior (vpfits.py:361-395) with the stretch-move ensemble.  ``step_method``
        is accepted for signature compatibility and ignored (there is one move)."""
        try:
            getattr(self, 'map')
        except AttributeError:
            print("\nWARNING: MAP estimate not provided. \nIt is recommended to compute this "
                  "in advance of running the MCMC so as to start the sampling with good initial values.")
        self.mcmc = _EnsembleMCMC(self)
        starttime = datetime.datetime.now()
        self.mcmc.sample(iter=iterations, burn=burnin, thin=thinning)
        self.fit_time = str(datetime.datetime.now() - starttime)
        print("\nTook:", self.fit_time, " to finish.")

    def _draw_walkers(self):
        """[W, D] start of the ensemble before the prior check: a ball around the current point (the
        MAP, when map_estimate ran first) and one prior draw per walker to fall back on."""
        W = int(self.nwalkers)
        W = max(W, 2 * self._ndim + 2)
        W += W % 2
        rng = np.random.default_rng((int(self._seed) >> 16) & 0xFFFFFFFF)
        centre = self._theta_dev
        span = np.abs(self._draw_prior(rng, W) - centre)
        X0 = centre + 1e-2 * span * rng.standard_normal((W, self._ndim))
        prior = self._draw_prior(rng, W)
        return X0, prior

    def _finish_walkers(self, X0, prior, lnp):
        """walkers of the ball that fall outside the prior are replaced by their prior draw, so that
        a poor start cannot trap the ensemble; walker 0 is the current point itself"""
        bad = ~np.isfinite(lnp)
        X0[bad] = prior[bad]
        X0[0] = self._theta_dev
        return X0

    def _initial_walkers(self):
        X0, prior = self._draw_walkers()
        return self._finish_walkers(X0, prior, self._ctx.lnprob(X0, region=self._region))

    def _run_sampler(self, iterations, burn, thin):
        X0 = self._initial_walkers()
        W = X0.shape[0]
        self._ctx.sampler_init(X0, seed=int(self._seed), a=2.0, split_block=hb.default_split_block(W))
        if burn > 0:
            self._ctx.run(burn, store_chain=False)
        thin = max(1, thin)
        keep = max(thin, iterations - burn)            # always keep at least one sample
        res = self._ctx.run(keep, thin=thin)
        self._ingest_chain(res["chain"], res["lnprob"], res["n_accept"], burn + keep, keep, res["seconds"])

    def _ingest_chain(self, chain, lnpc, n_accept, steps, keep, seconds, scored=None, set_values=True):
        """chain [n_keep, W, D] / lnprob [n_keep, W] of THIS fit's region -> traces, acceptance,
        DIC / BPIC, node values.  ``scored`` = (lnprob, sum) of every kept sample and of the mean
        point when the caller scored them already (batched fits: all regions in one launch), or
        "skip": leave DIC / BPIC at None."""
        W = chain.shape[1]
        self._chain_dev, self._lnp_chain = chain, lnpc
        flat_dev = chain.reshape(-1, self._ndim)
        flat = self._to_caller(flat_dev)
        mc_ = self.mcmc
        mc_._flat, mc_._names, mc_._derived = flat, list(self._names), {}
        if self._voigt:      # the reference's callers ask for est_sigma_k in Voigt mode too (vpspectrum.py:400)
            for k in range(self._n):
                mc_._derived["est_sigma_%d" % k] = ("est_G_%d" % k, self.GaussianWidth)
        mc_.acceptance_fraction = float(np.mean(n_accept)) / max(1, steps)
        mc_.walker_steps_per_second = W * keep / seconds if seconds > 0 else float("nan")
        # information criteria from the chain (every kept sample scored on the device)
        if isinstance(scored, str) and scored == "skip":      # batched model selection: DIC / BPIC are not computed
            mc_.DIC = mc_.BPIC = None
            if set_values:
                i = np.unravel_index(np.argmax(lnpc), lnpc.shape)
                self._set_values(chain[i[0], i[1]])
            return
        mean_theta = flat_dev.mean(0)
        if scored is None:
            ll = self._loglike(flat_dev)
            ll_mean = self._loglike(mean_theta)[0]
        else:
            (lnp_s, ss_s), (lnp_m, ss_m) = scored
            ll = self._loglike_from_sum(flat_dev, lnp_s, ss_s)
            ll_mean = self._loglike_from_sum(mean_theta, np.atleast_1d(lnp_m), np.atleast_1d(ss_m))[0]
        dev_mean = float(np.mean(-2.0 * ll[np.isfinite(ll)]))
        dev_at_mean = float(-2.0 * ll_mean)
        mc_.DIC = 2 * dev_mean - dev_at_mean
        mc_.BPIC = 3 * dev_mean - 2 * dev_at_mean
        # leave the nodes at the best posterior sample (PyMC leaves them at the last sample)
        if set_values:
            i = np.unravel_index(np.argmax(lnpc), lnpc.shape)
            self._set_values(chain[i[0], i[1]])

    def find_bic(self, frequency_array, flux_array, n, noise_array, freedom, voigt=False,
                 iterations=3000, thin=15, burn=300, thorough=False):
        """Three independent {model, MCMC, MAP} repeats; collects map.BIC and the reduced chi^2 of
        the MAP model (vpfits.py:398-429).  The reference's form of the call (VPfit() without noise, thorough=False:
        vpregion.py:59) runs the three repeats as three regions of one context (_find_bic_folded); a fit with known
        noise, or thorough=True, runs them one after the other as written there."""
        if not thorough and self.noise is None:
            return self._find_bic_folded(frequency_array, flux_array, n, noise_array, freedom, voigt, iterations, thin, burn)
        self.bic_array = []
        self.red_chi_array = []
        for i in range(3):
            self._seed = (int(self._seed) * 6364136223846793005 + 1442695040888963407 + i) & (2 ** 64 - 1)
            self.initialise_model(frequency_array, flux_array, n, voigt=voigt)
            self.map = _MAP(self)
            self.mcmc = _EnsembleMCMC(self)
            if thorough:
                self.map.fit(iterlim=iterations, tol=1e-3)
                self.mcmc.sample(iter=iterations, burn=burn, thin=thin, progress_bar=False)
                self.map.fit(iterlim=iterations, tol=1e-3)
            self.mcmc.sample(iter=iterations, burn=burn, thin=thin, progress_bar=False)
            self.map.fit(iterlim=iterations, tol=1e-3)
            self.bic_array.append(self.map.BIC)
            self.red_chi_array.append(self.ReducedChisquared(flux_array, self.total.value, noise_array, freedom))
        return

    def _find_bic_folded(self, frequency_array, flux_array, n, noise_array, freedom, voigt, iterations, thin, burn):
        """The three repeats of ``find_bic`` as THREE regions of one context (vamp_amd.batched.find_bic_batched with one
        region): one ensemble run, one MAP launch and one model launch instead of three of each -- a third of the device
        calls of a path that is bound by them.  Every repeat has a chain of its own (its region index keys the draws);
        ``self`` ends as the last repeat's fit, as in the reference (vpfits.py:417-428), bound to a context that holds
        this one region again, so every method keeps working on it."""
        from .batched import find_bic_batched            # (batched imports this module)
        self._seed = (int(self._seed) * 6364136223846793005 + 1442695040888963407) & (2 ** 64 - 1)
        if self._ctx is None:
            self._ctx = hb.HipContext(device=self.device, dtype=self.dtype)
        ctx, mine = self._ctx, {k: self.__dict__[k] for k in ("noise", "std_deviation", "verbose", "device", "dtype", "_seed", "nwalkers")
                                if k in self.__dict__}
        res = find_bic_batched(ctx, [(np.asarray(frequency_array, dtype=np.float64), np.asarray(flux_array, dtype=np.float64), noise_array)],
                               [int(n)], voigt=voigt, nwalkers=self.nwalkers, iterations=iterations, thin=thin, burn=burn,
                               seed=int(self._seed) & 0x7FFFFFFFFFFFFFFF, freedoms=[freedom], score_chains=True)[0]
        last = res.detach().fit()
        self.__dict__.update(last.__dict__)
        self.__dict__.update(mine)
        self.map._fit = self.mcmc._fit = self
        self.bic_array, self.red_chi_array = list(res.bic_array), list(res.red_chi_array)
        # the context held the three repeats; this fit is a fit of ONE region again
        self._region, self._shared_ctx = 0, False
        ctx.set_regions(self._x, self._flux, np.ones_like(self._flux), self._n, mode=self._mode, sample_sd=True)

    def chain_covariance(self, n, voigt=False):
        """Per-component 3x3 covariance of (amplitude, sigma, centroid) samples (vpfits.py:432-456)."""
        cov = np.zeros((n, 3, 3))
        for i in range(n):
            amp_samples = self.mcmc.trace('xexp_' + str(i))[:]
            if not voigt:
                sigma_samples = self.mcmc.trace('est_sigma_' + str(i))[:]
            else:
                gfwhm_samples = self.mcmc.trace('est_G_' + str(i))[:]
                sigma_samples = self.GaussianWidth(gfwhm_samples)
            c_samples = self.mcmc.trace('est_centroid_' + str(i))[:]
            cov[i] = np.cov(np.array((amp_samples, sigma_samples, c_samples)))
        return cov

    def plot(self, wavelength_array, flux_array, clouds=None, n=1, onesigmaerror=0.02,
             start_pix=None, end_pix=None, filename=None):
        """Diagnostic figure with the reference's signature (vpfits.py:134): residuals in units of
        ``onesigmaerror``, the fitted components, the total model.  Plotting is outside the hot
        path; this is a minimal stand-in so that callers of ``plot`` keep working."""
        import matplotlib
        if filename:
            matplotlib.use("Agg")
        from matplotlib import pyplot
        lam = np.asarray(wavelength_array)
        lo = start_pix or 0
        hi = end_pix or lam.size
        fig, panels = pyplot.subplots(3, 1, sharex=True, figsize=(10, 10), gridspec_kw={"hspace": 0})
        resid, comps, total = panels
        resid.plot(lam, (np.asarray(flux_array) - self.total.value) / onesigmaerror)
        for level in (-3, -1, 1, 3):
            resid.axhline(level, color="red", linestyle="-" if abs(level) == 1 else "--")
        comps.plot(lam, flux_array, color="black", linewidth=1.0)
        truth = [] if clouds is None else [np.asarray(clouds.iloc[i]["tau"])[lo:hi] for i in range(len(clouds))]
        fitted = [self.estimated_profiles[k].value for k in range(n)]
        for taus, colour, label, width in ((truth, "red", "Actual", 1.5), (fitted, "green", "Fit", None)):
            for i, tau in enumerate(taus):
                comps.plot(lam, Tau2flux(tau), color=colour, label=label if i == 0 else None, lw=width)
        total.plot(lam, flux_array, label="Measured")
        total.plot(lam, self.total.value, color="green", linewidth=2.0, label="Fit")
        for ax, ylabel in zip(panels, ("Residuals", "Normalised Flux", "Normalised Flux")):
            ax.set_ylabel(ylabel)
        comps.legend()
        total.legend()
        total.set_xlabel(r"$ \lambda (\AA)$")
        if getattr(self, "fit_time", None):
            resid.set_title("Fit time: " + self.fit_time)
        if not filename:
            pyplot.show()
            return
        fig.savefig(filename)
        pyplot.close(fig)

    # copy.copy(fit) is used by VPregion (vpregion.py:65,72): share the device context
    def __copy__(self):
        new = self.__class__.__new__(self.__class__)
        new.__dict__.update(self.__dict__)
        return new
