"""``VPregion`` -- model selection for one absorption region (reference vamp_1.0/vpregion.py:8-91)
on top of the HIP-backed ``VPfit``.

Same constructor, attributes (``n``, ``freedom``, ``fit``, ``num_pixels`` ...) and the same
add-a-component-while-BIC-falls loop.  Host control flow only: every posterior evaluation and every
sampler step it triggers runs in libvamp_hip.so.  ``fit_region_pygadds`` (vpregion.py:94-144) is
dead Python-2 code in the reference and is not reproduced.
"""
from copy import copy

import numpy as np
from scipy.ndimage import gaussian_filter
from scipy.signal import argrelextrema

from .vpfits import VPfit

MAX_COMPONENTS = 32      # VAMP_MAX_COMPONENTS of include/vamp_hip.h (the reference sets no limit, vpspectrum.py:287-294)


class VPregion():

    def __init__(self, frequency_array, flux_array, noise_array, voigt=False, chi_limit=1.5, nwalkers=None, seed=None,
                 dtype=None, device=0):
        self.frequency_array = frequency_array
        self.flux_array = flux_array
        self.noise_array = noise_array
        self.voigt = voigt
        self.chi_limit = chi_limit
        self.num_pixels = len(flux_array)
        self.nwalkers = nwalkers
        self._seed = seed
        self.dtype, self.device = dtype, device    # per-pixel arithmetic of the fits (None: $VAMP_DTYPE, else fp64) and GPU
        self._attempt = 0         # region_fit calls so far: a retry must not replay the previous attempt's draws
        self.estimate_n()
        self.set_freedom()

    def estimate_n(self):
        """Initial guess of the number of lines: local minima of the flux smoothed with a
        sigma = 3 px Gaussian; fewer than 4 minima -> 1 (vpregion.py:21-35)."""
        self.n = argrelextrema(gaussian_filter(self.flux_array, 3), np.less)[0].shape[0]
        if self.n < 4:
            self.n = 1

    def set_freedom(self):
        """degrees of freedom = pixels - 3 n (3 n for Voigt too, vpregion.py:37-39)"""
        self.freedom = self.num_pixels - 3 * self.n

    def _fit_n(self, n, iterations, thin, burn):
        fit = VPfit(seed=None if self._seed is None else self._seed + 1000 * n + 1000003 * (self._attempt - 1),
                    dtype=self.dtype, device=self.device)
        if self.nwalkers is not None:
            fit.nwalkers = self.nwalkers
        fit.find_bic(self.frequency_array, self.flux_array, n, self.noise_array, self.freedom,
                     voigt=self.voigt, iterations=iterations, thin=thin, burn=burn)
        return fit

    def region_fit(self, verbose=True, iterations=3000, thin=15, burn=300):
        """BIC ladder of vpregion.py:42-91: fit n, then n+1, n+2, ... while the mean BIC of the
        three repeats keeps falling; stop early once the mean reduced chi^2 is under ``chi_limit``.
        The first rung is judged by the LAST of its three BICs, as in the reference (:63)."""
        say = print if verbose else (lambda *a, **k: None)
        self._attempt += 1
        say("Setting initial number of lines to: {}".format(self.n))
        kept = self._fit_n(self.n, iterations, thin, burn)
        kept_bic = kept.bic_array[-1]
        while True:
            if self.n >= MAX_COMPONENTS:
                say("Reached the {}-component limit of the device kernels.".format(MAX_COMPONENTS))
                break
            self.n += 1
            say("Trying n={} lines.".format(self.n))
            trial = self._fit_n(self.n, iterations, thin, burn)
            trial_bic = float(np.average(trial.bic_array))
            if not (kept_bic > trial_bic):
                # the reference decrements n only inside `if verbose` (:82-86); the fit it keeps is
                # the previous rung, so n is restored unconditionally here
                self.n -= 1
                say("BIC rose from {:.2f} to {:.2f}: keeping n={}.".format(kept_bic, trial_bic, self.n))
                self._release(trial)
                break
            say("BIC fell from {:.2f} to {:.2f}.".format(kept_bic, trial_bic))
            self._release(kept)
            kept, kept_bic = copy(trial), trial_bic
            if np.average(trial.red_chi_array) < self.chi_limit:
                say("Reduced chi squared below {}: final n={}.".format(self.chi_limit, self.n))
                break
        self.fit = kept

    @staticmethod
    def _release(fit):
        """Free the device context of a fit the ladder drops.  (The reference calls gc.collect() after every ladder,
        vpregion.py:90, to shed PyMC's models; a fit object and its MAP / MCMC stand-ins form a reference cycle here too,
        but its only heavy member is the context, closed here -- a full collection costs ~50 ms per ladder in a process
        that has torch loaded: 3.7 of the 6.8 s of a 20-region sequential fit.)"""
        ctx = getattr(fit, "_ctx", None)
        if ctx is not None and not getattr(fit, "_shared_ctx", False) and hasattr(ctx, "close"):
            ctx.close()
