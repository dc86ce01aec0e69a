"""CPU oracle for the VAMP MCMC hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product path (``vamp_amd``) never imports anything from ``oracle/``.

It restates, in numpy + ``scipy.special.wofz``, the arithmetic of the reference hot path
(``/root/reference/vamp_1.0``; citations are file:line in that tree):

  * Gaussian tau-profile                ``vpfits.py:43-54``
  * Voigt tau-profile                   ``vpfits.py:57-76`` (astropy ``Voigt1D`` in its wofz form,
                                        equal to the commented formula at ``vpfits.py:72-73``
                                        times ``amplitude_L*pi*fwhm_L/2``)
  * ``flux = exp(-sum_k tau_k)``        ``physics.py:98-105``, ``vpfits.py:334-336``
  * chi^2 / reduced chi^2               ``vpfits.py:109-131``, ``vpregion.py:37-39``
  * observed Normal likelihood          ``vpfits.py:39,341``
  * priors (xexp / Uniform)             ``vpfits.py:239-252,283-297,320,326``
  * unit maps (N, b, z) <-> fit params  ``physics.py:3-27,116-134``, ``vpfits.py:79-88``

The sampler is the north star's substitution for PyMC 2's Metropolis (``vpfits.py:361-395``): the
affine-invariant stretch move of Goodman & Weare (2010) with emcee-v3 ``StretchMove`` /
``RedBlueMove`` semantics (a = 2, red/blue halves, membership reshuffled every step).  emcee is
not a dependency of the reference and is not installed, so the move is restated from the
published algorithm; all random draws are injectable so that CPU and GPU follow one trajectory.

Parity pinning: the numpy-only reference statics (``GaussFunction``, ``Chisquared``,
``ReducedChisquared``, ``GaussianWidth``) and ``physics.py`` are imported from the reference in
``tests/golden/make_golden.py`` and their outputs are committed as fixtures; this oracle is
checked against them in ``tests/test_oracle.py``.  Voigt values are pinned by
``scipy.special.wofz`` (the north star's named oracle) and spot-checked against mpmath.  The
reference holds no numeric known-answer for Voigt values, BIC values or sampler statistics (astropy,
pymc absent and unpinned), so for those quantities parity is "unpinned" beyond scipy/mpmath.  The
DEFINITION of the MAP information criteria is pinned by the outputs the reference's
vpfits_intro.ipynb still holds (BIC - AIC = k (ln n - 2) with k counting `sd`, n = pixels:
tests/test_notebook_pins.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
from scipy.special import wofz

# physics.py:3-4 -- constants reproduced exactly (c = 2.98e8 is the reference's value, sic)
C_LIGHT = 2.98e8
SIGMA0 = 0.0263

MODE_GAUSS3 = 0   # theta_k = (amplitude, centroid, sigma)            vpfits.py:219-262
MODE_VOIGT4 = 1   # theta_k = (amplitude, centroid, L_fwhm, G_fwhm)   vpfits.py:265-307
MODE_NBZ3 = 2     # theta_k = (N, b, z) mapped through physics.py:6-27,116-134 (north-star form)
Q_OF_MODE = {MODE_GAUSS3: 3, MODE_VOIGT4: 4, MODE_NBZ3: 3}

SQRT_LN2 = math.sqrt(math.log(2.0))
FWHM_PER_SIGMA = 2.0 * math.sqrt(2.0 * math.log(2.0))   # vpfits.py:88,326


# --------------------------------------------------------------------------------------
# L1 profile + statistics restatements
# --------------------------------------------------------------------------------------
def gauss_function(x, amplitude, centroid, sigma):
    """vpfits.py:54"""
    return amplitude * np.exp(-0.5 * ((x - centroid) / sigma) ** 2)


def voigt_function(x, centroid, amplitude, L_fwhm, G_fwhm):
    """vpfits.py:57-76: astropy Voigt1D(x_0, amplitude_L, fwhm_L, fwhm_G) in its wofz form,
    amplitude_L * fwhm_L * sqrt(pi ln2) / fwhm_G * Re w(z), z = (2(x-x0) + i fwhm_L) sqrt(ln2)/fwhm_G.
    """
    z = (2.0 * (x - centroid) + 1j * L_fwhm) * SQRT_LN2 / G_fwhm
    return amplitude * L_fwhm * math.sqrt(math.pi) * SQRT_LN2 / G_fwhm * wofz(z).real


def voigt_function_commented(x, centroid, amplitude, L_fwhm, G_fwhm):
    """The formula left in comments at vpfits.py:72-73 (alpha, gamma = HWHMs), scaled by
    amplitude*pi*fwhm_L/2 so that its peak Lorentzian amplitude matches Voigt1D's."""
    alpha = 0.5 * G_fwhm
    gamma = 0.5 * L_fwhm
    sigma = alpha / np.sqrt(2 * np.log(2))
    v = np.real(wofz((x - centroid + 1j * gamma) / sigma / np.sqrt(2))) / sigma / np.sqrt(2 * np.pi)
    return v * amplitude * np.pi * L_fwhm / 2.0


def gaussian_width(G_fwhm):
    """vpfits.py:88"""
    return G_fwhm / (2.0 * np.sqrt(2.0 * np.log(2.0)))


def tau2flux(tau):
    """physics.py:105"""
    return np.exp(-tau)


def chisquared(observed, expected, noise):
    """vpfits.py:118 (python ``sum`` = sequential accumulation)"""
    return sum(((observed - expected) / noise) ** 2)


def reduced_chisquared(observed, expected, noise, freedom):
    """vpfits.py:131"""
    return chisquared(observed, expected, noise) / freedom


def wave2freq(wavelength):
    """physics.py:126"""
    return C_LIGHT / (wavelength * 1.0e-10)


def freq2wave(frequency):
    """physics.py:120"""
    return (C_LIGHT / frequency) / 1.0e-10


def wave2red(wave, rest_wave):
    """physics.py:134"""
    return (wave - rest_wave) / rest_wave


def column_density(amplitude, sigma):
    """physics.py:15"""
    return amplitude * sigma * np.sqrt(2 * np.pi) / SIGMA0


def doppler_parameter(sigma, line):
    """physics.py:26-27 (line in Angstrom)"""
    line = line * 1.0e-10
    return (line * sigma * 2.355 / np.sqrt(2)) * 1.0e-3


def nbz_to_native(N, b, z, line):
    """Inverse of physics.py:15 (ColumnDensity), :27 (DopplerParameter), :120,:134
    (Freq2wave, Wave2red): (N, b, z) -> (amplitude, centroid[Hz], sigma[Hz])."""
    sigma = b * 1.0e3 * math.sqrt(2.0) / (2.355 * (line * 1.0e-10))
    amplitude = N * SIGMA0 / (sigma * math.sqrt(2.0 * math.pi))
    centroid = C_LIGHT / (line * (1.0 + z) * 1.0e-10)
    return amplitude, centroid, sigma


def native_to_nbz(amplitude, centroid, sigma, line):
    N = column_density(amplitude, sigma)
    b = doppler_parameter(sigma, line)
    z = wave2red(freq2wave(centroid), line)
    return N, b, z


# --------------------------------------------------------------------------------------
# One absorption region = one posterior
# --------------------------------------------------------------------------------------
@dataclass
class Region:
    """Inputs of one log-posterior (``VPfit.initialise_model``, vpfits.py:310-349).

    ``x`` is the (ascending) abscissa the profiles are evaluated on.  Any affine re-centring of
    the frequency axis is the caller's business: all width/centroid parameters live in the same
    units as ``x``.  For ``MODE_NBZ3`` the physical frequency is ``nu = x_origin + x_scale*x``.
    """
    x: np.ndarray
    flux: np.ndarray
    noise: np.ndarray
    n_comp: int
    mode: int = MODE_VOIGT4
    sample_sd: bool = False       # True = reference likelihood with free precision (vpfits.py:39)
    include_norm: bool = False    # add -1/2 sum log(2 pi sigma_i^2) (vamp_2.0/vamp_src/fit/fit.py:156,171)
    l_fixed: float = 0.0          # NBZ3: fixed Lorentzian FWHM, in units of x
    line: float = 1215.67         # NBZ3: rest wavelength [Angstrom]
    x_origin: float = 0.0         # NBZ3: nu = x_origin + x_scale * x   [Hz]
    x_scale: float = 1.0
    # prior bounds (vpfits.py:249-252, 292-297, 320, 326); filled by __post_init__ when None
    c_lo: float | None = None
    c_hi: float | None = None
    sigma_max: float | None = None
    fwhm_max: float | None = None

    def __post_init__(self):
        self.x = np.ascontiguousarray(self.x, dtype=np.float64)
        self.flux = np.ascontiguousarray(self.flux, dtype=np.float64)
        self.noise = np.ascontiguousarray(self.noise, dtype=np.float64)
        if self.c_lo is None:
            self.c_lo = float(self.x[0])           # vpfits.py:250
        if self.c_hi is None:
            self.c_hi = float(self.x[-1])
        if self.sigma_max is None:
            self.sigma_max = (float(self.x[-1]) - float(self.x[0])) / 2.0   # vpfits.py:320
        if self.fwhm_max is None:
            self.fwhm_max = self.sigma_max * 2 * np.sqrt(2 * np.log(2.0))   # vpfits.py:326

    @property
    def q(self):
        return Q_OF_MODE[self.mode]

    @property
    def ndim(self):
        return self.q * self.n_comp + (1 if self.sample_sd else 0)

    @property
    def norm_const(self):
        if not self.include_norm or self.sample_sd:
            return 0.0
        return -0.5 * float(np.sum(np.log(2.0 * np.pi * self.noise ** 2)))


def _xexp_logp(v):
    """vpfits.py:239-244 / 283-288, literally: -inf for v<0 else log(v*exp(-v))."""
    if v < 0 or not np.isfinite(v):
        return -np.inf
    with np.errstate(divide="ignore"):
        return float(np.log(v * np.exp(-v)))


def _uniform_logp(v, lo, hi):
    """PyMC 2 ``uniform_like``: -log(hi-lo) inside [lo, hi], -inf outside (from knowledge;
    pymc absent)."""
    if not (v >= lo and v <= hi):
        return -np.inf
    return -math.log(hi - lo)


def native_components(region: Region, theta):
    """theta[D] -> list of per-component native tuples used by the profile functions."""
    q = region.q
    comps = []
    for k in range(region.n_comp):
        t = theta[q * k:q * k + q]
        if region.mode == MODE_GAUSS3:
            comps.append((float(t[0]), float(t[1]), float(t[2])))
        elif region.mode == MODE_VOIGT4:
            comps.append((float(t[0]), float(t[1]), float(t[2]), float(t[3])))
        else:
            amp, nu_c, sig = nbz_to_native(float(t[0]), float(t[1]), float(t[2]), region.line)
            c = (nu_c - region.x_origin) / region.x_scale
            G = (sig / region.x_scale) * FWHM_PER_SIGMA
            comps.append((amp, c, region.l_fixed, G))
    return comps


def log_prior(region: Region, theta):
    """Sum of the priors of vpfits.py:239-252 (Gaussian) / 283-297 (Voigt) and, when sampled,
    ``sd ~ U(0,1)`` (vpfits.py:39).  NBZ3 applies the Voigt priors to the mapped native
    parameters (amplitude, centroid, G_fwhm); no Jacobian (north-star construct, documented in
    DESIGN.md)."""
    lp = 0.0
    with np.errstate(all="ignore"):
        comps = native_components(region, theta)
    for comp in comps:
        if region.mode == MODE_GAUSS3:
            a, c, s = comp
            lp += _xexp_logp(a)
            lp += _uniform_logp(c, region.c_lo, region.c_hi)
            lp += _uniform_logp(s, 0.0, region.sigma_max)
        elif region.mode == MODE_VOIGT4:
            a, c, L, G = comp
            lp += _xexp_logp(a)
            lp += _uniform_logp(c, region.c_lo, region.c_hi)
            lp += _uniform_logp(L, 0.0, region.fwhm_max)
            lp += _uniform_logp(G, 0.0, region.fwhm_max)
        else:
            a, c, L, G = comp
            lp += _xexp_logp(a)
            lp += _uniform_logp(c, region.c_lo, region.c_hi)
            lp += _uniform_logp(G, 0.0, region.fwhm_max)
    if region.sample_sd:
        lp += _uniform_logp(float(theta[-1]), 0.0, 1.0)
    return lp


def component_taus(region: Region, theta):
    """[K, P] optical depths (the ``component_k`` deterministics, vpfits.py:254-260 / 299-305)."""
    out = np.empty((region.n_comp, region.x.size))
    with np.errstate(all="ignore"):
        for k, comp in enumerate(native_components(region, theta)):
            if region.mode == MODE_GAUSS3:
                a, c, s = comp
                out[k] = gauss_function(region.x, a, c, s)
            else:
                a, c, L, G = comp
                out[k] = voigt_function(region.x, c, a, L, G)
    return out


def model_flux(region: Region, theta):
    """``total`` deterministic, vpfits.py:334-336: Tau2flux(sum(profiles)); python ``sum`` adds
    the K arrays in order starting from 0."""
    taus = component_taus(region, theta)
    with np.errstate(all="ignore"):
        return tau2flux(sum(list(taus)))


def log_like(region: Region, theta, return_chi2=False):
    """Known noise: -1/2 chi^2 (+ norm);  free precision (vpfits.py:39,341):
    sum_i [ 1/2 log(t/2pi) - 1/2 t (f_i - m_i)^2 ],  t = 1/sd^2."""
    m = model_flux(region, theta)
    with np.errstate(all="ignore"):
        if region.sample_sd:
            sd = float(theta[-1])
            t = 1.0 / sd ** 2
            s = float(np.sum((region.flux - m) ** 2))
            ll = region.x.size * 0.5 * math.log(t / (2.0 * math.pi)) - 0.5 * t * s if sd > 0 else -np.inf
            chi2 = s
        else:
            chi2 = float(np.sum(((region.flux - m) / region.noise) ** 2))
            ll = -0.5 * chi2 + region.norm_const
    if return_chi2:
        return ll, chi2
    return ll


def log_prob(region: Region, theta, return_chi2=False):
    """Per-walker log-posterior = log-prior + log-likelihood; NaN -> -inf (emcee convention)."""
    theta = np.asarray(theta, dtype=np.float64)
    lp = log_prior(region, theta)
    if not np.isfinite(lp):
        return (-np.inf, np.nan) if return_chi2 else -np.inf
    ll, chi2 = log_like(region, theta, return_chi2=True)
    val = lp + ll
    if np.isnan(val):
        val = -np.inf
    return (val, chi2) if return_chi2 else val


def log_prob_batch(region: Region, thetas, return_chi2=False):
    thetas = np.asarray(thetas, dtype=np.float64)
    out = np.empty(thetas.shape[0])
    chi = np.empty(thetas.shape[0])
    for i in range(thetas.shape[0]):
        out[i], chi[i] = log_prob(region, thetas[i], return_chi2=True)
    return (out, chi) if return_chi2 else out


def log_prob_batch_fast(region: Region, thetas):
    """Vectorised over walkers (same arithmetic, used for the timed CPU baseline and the larger
    parity cases).  Only MODE_VOIGT4 / MODE_GAUSS3 / MODE_NBZ3 with known noise or free sd."""
    thetas = np.asarray(thetas, dtype=np.float64)
    W = thetas.shape[0]
    q, K = region.q, region.n_comp
    lp = np.zeros(W)
    tau = np.zeros((W, region.x.size))
    x = region.x[None, :]
    with np.errstate(all="ignore"):
        for k in range(K):
            t = thetas[:, q * k:q * k + q]
            if region.mode == MODE_GAUSS3:
                a, c, s = t[:, 0], t[:, 1], t[:, 2]
                lp += np.where((s >= 0) & (s <= region.sigma_max), -math.log(region.sigma_max), -np.inf)
                prof = a[:, None] * np.exp(-0.5 * ((x - c[:, None]) / s[:, None]) ** 2)
            else:
                if region.mode == MODE_VOIGT4:
                    a, c, L, G = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
                    lp += np.where((L >= 0) & (L <= region.fwhm_max), -math.log(region.fwhm_max), -np.inf)
                else:
                    sig = t[:, 1] * 1.0e3 * math.sqrt(2.0) / (2.355 * (region.line * 1.0e-10))
                    a = t[:, 0] * SIGMA0 / (sig * math.sqrt(2.0 * math.pi))
                    c = (C_LIGHT / (region.line * (1.0 + t[:, 2]) * 1.0e-10) - region.x_origin) / region.x_scale
                    G = (sig / region.x_scale) * FWHM_PER_SIGMA
                    L = np.full(W, region.l_fixed)
                lp += np.where((G >= 0) & (G <= region.fwhm_max), -math.log(region.fwhm_max), -np.inf)
                z = (2.0 * (x - c[:, None]) + 1j * L[:, None]) * SQRT_LN2 / G[:, None]
                prof = (a * L * math.sqrt(math.pi) * SQRT_LN2 / G)[:, None] * wofz(z).real
            lp += np.where(a >= 0, np.log(a * np.exp(-a)), -np.inf)
            lp += np.where((c >= region.c_lo) & (c <= region.c_hi), -math.log(region.c_hi - region.c_lo), -np.inf)
            tau = tau + prof
        m = np.exp(-tau)
        if region.sample_sd:
            sd = thetas[:, -1]
            lp += np.where((sd >= 0) & (sd <= 1), 0.0, -np.inf)
            s = np.sum((region.flux[None, :] - m) ** 2, axis=1)
            tt = 1.0 / sd ** 2
            ll = region.x.size * 0.5 * np.log(tt / (2.0 * math.pi)) - 0.5 * tt * s
        else:
            ll = -0.5 * np.sum(((region.flux[None, :] - m) / region.noise[None, :]) ** 2, axis=1) + region.norm_const
        out = lp + ll
    out = np.where(np.isfinite(lp), out, -np.inf)
    out = np.where(np.isnan(out), -np.inf, out)
    return out


# --------------------------------------------------------------------------------------
# Counter-based RNG shared (bit-for-bit) with the HIP sampler: Philox4x32-10
# (Salmon, Moraes, Dror & Shaw, SC'11 -- the published constants)
# --------------------------------------------------------------------------------------
PHILOX_M0 = 0xD2511F53
PHILOX_M1 = 0xCD9E8D57
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
MASK32 = 0xFFFFFFFF


def philox4x32_10(counter, key):
    c0, c1, c2, c3 = [int(v) & MASK32 for v in counter]
    k0, k1 = [int(v) & MASK32 for v in key]
    for _ in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> 32, p0 & MASK32
        hi1, lo1 = p1 >> 32, p1 & MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & MASK32, lo1, (hi0 ^ c3 ^ k1) & MASK32, lo0
        k0 = (k0 + PHILOX_W0) & MASK32
        k1 = (k1 + PHILOX_W1) & MASK32
    return c0, c1, c2, c3


def _u53(hi, lo):
    """two 32-bit words -> double in [0,1) with 53 random bits"""
    return float(((hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0)


STREAM_MOVE = 0     # (u1 -> z) and partner index
STREAM_ACCEPT = 1   # u2
STREAM_SPLIT = 2    # red/blue membership keys


def draw_move(seed, step, half, walker_gid, n_complement, a=2.0):
    """Per-(step, half, global walker id) draws, independent of the GPU count:
    returns (z, partner_slot, log_u2).  z = ((a-1) u1 + 1)^2 / a."""
    key = (seed & MASK32, (seed >> 32) & MASK32)
    r = philox4x32_10((walker_gid & MASK32, step & MASK32, (half << 8) | STREAM_MOVE, (walker_gid >> 32) & MASK32), key)
    u1 = _u53(r[0], r[1])
    t = (a - 1.0) * u1 + 1.0
    z = t * t / a
    j = (((r[2] << 32) | r[3]) * n_complement) >> 64          # multiply-high map to [0, n)
    r2 = philox4x32_10((walker_gid & MASK32, step & MASK32, (half << 8) | STREAM_ACCEPT, (walker_gid >> 32) & MASK32), key)
    u2 = _u53(r2[0], r2[1])
    logu = math.log(u2) if u2 > 0 else -math.inf
    return z, int(j), logu


def split_perm(seed, step, chunk, slot, block, region=0):
    """Keyed bijection of [0, block): slot -> local walker index inside split chunk ``chunk``.
    Slots [0, block/2) are 'red', [block/2, block) 'blue'.  Affine-multiply / xorshift rounds
    on the next power of two with cycle walking."""
    key = (seed & MASK32, (seed >> 32) & MASK32)
    r = philox4x32_10((chunk & MASK32, step & MASK32, STREAM_SPLIT, region & MASK32), key)
    bits = max(1, (block - 1).bit_length())
    mask = (1 << bits) - 1
    sh = max(1, bits // 2)
    x = slot
    while True:
        x = (x * ((r[0] << 1) | 1) + r[1]) & mask
        x ^= x >> sh
        x = (x * ((r[2] << 1) | 1) + r[3]) & mask
        x ^= x >> sh
        x = (x * 0x9E3779B1 + (r[0] ^ r[3])) & mask
        x ^= x >> sh
        if x < block:
            return x


def split_tables(seed, step, n_walkers, block, region=0):
    """Full membership for one step: returns (red[W/2], blue[W/2]) global walker ids ordered by
    global active-slot index (chunk-major)."""
    assert n_walkers % block == 0 and block % 2 == 0
    half = block // 2
    red = np.empty(n_walkers // 2, dtype=np.int64)
    blue = np.empty(n_walkers // 2, dtype=np.int64)
    for ch in range(n_walkers // block):
        for s in range(half):
            red[ch * half + s] = ch * block + split_perm(seed, step, ch, s, block, region)
            blue[ch * half + s] = ch * block + split_perm(seed, step, ch, half + s, block, region)
    return red, blue


# numpy-vectorised forms of the three functions above (same arithmetic on uint64 arrays), so that
# production-size ensembles can be replayed in seconds; tests/test_oracle.py pins them to the scalar
# forms.
def _philox_vec(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = [np.asarray(v, dtype=np.uint64) & np.uint64(MASK32) for v in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = np.uint64(int(k0) & MASK32), np.uint64(int(k1) & MASK32)
    m = np.uint64(MASK32)
    s32 = np.uint64(32)
    for _ in range(10):
        p0 = np.uint64(PHILOX_M0) * c0
        p1 = np.uint64(PHILOX_M1) * c2
        c0, c1, c2, c3 = ((p1 >> s32) ^ c1 ^ k0) & m, p1 & m, ((p0 >> s32) ^ c3 ^ k1) & m, p0 & m
        k0 = (k0 + np.uint64(PHILOX_W0)) & m
        k1 = (k1 + np.uint64(PHILOX_W1)) & m
    return c0, c1, c2, c3


def _mulhi64(a, b):
    """high 64 bits of the 128-bit product of uint64 arrays"""
    m = np.uint64(MASK32)
    s32 = np.uint64(32)
    a0, a1, b0, b1 = a & m, a >> s32, b & m, b >> s32
    t = a0 * b0
    u = a1 * b0 + (t >> s32)
    v = a0 * b1 + (u & m)
    return a1 * b1 + (u >> s32) + (v >> s32)


def split_perm_batch(seed, step, chunk, slot, block, region=0):
    """split_perm for arrays of (chunk, slot)"""
    chunk = np.asarray(chunk, dtype=np.uint64)
    r0, r1, r2, r3 = _philox_vec(chunk, np.uint64(step & MASK32), np.uint64(STREAM_SPLIT), np.uint64(region & MASK32),
                                 seed & MASK32, (seed >> 32) & MASK32)
    bits = max(1, (block - 1).bit_length())
    mask = np.uint64((1 << bits) - 1)
    sh = np.uint64(max(1, bits // 2))
    one = np.uint64(1)
    x = np.broadcast_to(np.asarray(slot, dtype=np.uint64), r0.shape).copy()
    done = np.zeros(x.shape, dtype=bool)
    while not done.all():
        y = (x * ((r0 << one) | one) + r1) & mask
        y ^= y >> sh
        y = (y * ((r2 << one) | one) + r3) & mask
        y ^= y >> sh
        y = (y * np.uint64(0x9E3779B1) + (r0 ^ r3)) & mask
        y ^= y >> sh
        x = np.where(done, x, y)
        done |= x < np.uint64(block)
    return x.astype(np.int64)


def draw_move_batch(seed, step, half, walker_gid, n_complement, a=2.0):
    """draw_move for an array of global walker ids: (z, partner_slot, log_u2) arrays"""
    gid = np.asarray(walker_gid, dtype=np.uint64)
    s32 = np.uint64(32)
    k0, k1 = seed & MASK32, (seed >> 32) & MASK32
    r = _philox_vec(gid, np.uint64(step & MASK32), np.uint64((half << 8) | STREAM_MOVE), gid >> s32, k0, k1)
    u1 = (((r[0] << s32) | r[1]) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    t = (a - 1.0) * u1 + 1.0
    z = t * t / a
    j = _mulhi64((r[2] << s32) | r[3], np.uint64(n_complement)).astype(np.int64)
    r2 = _philox_vec(gid, np.uint64(step & MASK32), np.uint64((half << 8) | STREAM_ACCEPT), gid >> s32, k0, k1)
    u2 = (((r2[0] << s32) | r2[1]) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    with np.errstate(divide="ignore"):
        logu = np.where(u2 > 0, np.log(u2), -np.inf)
    return z, j, logu


def split_tables_batch(seed, step, n_walkers, block, region=0):
    """split_tables through split_perm_batch"""
    half = block // 2
    ch = np.repeat(np.arange(n_walkers // block), half)
    s_ = np.tile(np.arange(half), n_walkers // block)
    red = ch * block + split_perm_batch(seed, step, ch, s_, block, region)
    blue = ch * block + split_perm_batch(seed, step, ch, s_ + half, block, region)
    return red, blue


def run_sampler_batch(lnprob_fn, X0, lnp0, n_steps, seed, block, a=2.0, step0=0, region=0, walker_off=0):
    """run_sampler with the vectorised draws (same trajectory); returns chain, lnp_chain, n_accept"""
    X = np.array(X0, dtype=np.float64)
    lnp = np.array(lnp0, dtype=np.float64)
    W = X.shape[0]
    nacc = np.zeros(W, dtype=np.int64)
    chain, lchain = [], []
    for it in range(n_steps):
        step = step0 + it
        red, blue = split_tables_batch(seed, step, W, block, region)
        for half in (0, 1):
            act, comp = (red, blue) if half == 0 else (blue, red)
            zz, j, logu = draw_move_batch(seed, step, half, act + walker_off, act.size, a)
            acc, _ = stretch_half_step(X, lnp, act, comp[j], zz, logu, lnprob_fn)
            nacc[act[acc]] += 1
        chain.append(X.copy())
        lchain.append(lnp.copy())
    return np.array(chain), np.array(lchain), nacc


# --------------------------------------------------------------------------------------
# Stretch move
# --------------------------------------------------------------------------------------
def stretch_half_step(X, lnp, active, partner, zz, logu, lnprob_fn):
    """One half-step with every draw injected.

    X[W,D], lnp[W] are updated in place.  ``active[n]`` = walker ids being moved, ``partner[n]``
    = walker ids (members of the frozen complement) they stretch against, ``zz[n]`` the stretch
    factors, ``logu[n]`` = log(u2).  q = c - (c - s) z;  accept iff
    log u2 < (D-1) log z + lnp(q) - lnp(s).  Returns (accepted mask, proposal lnprob)."""
    D = X.shape[1]
    Xc = X[partner]
    Xs = X[active]
    q = Xc - (Xc - Xs) * zz[:, None]
    lnp_q = np.asarray(lnprob_fn(q), dtype=np.float64)
    lnp_q = np.where(np.isnan(lnp_q), -np.inf, lnp_q)
    with np.errstate(invalid="ignore"):
        lnpdiff = (D - 1.0) * np.log(zz) + lnp_q - lnp[active]
    acc = logu < lnpdiff
    acc = np.where(np.isnan(lnpdiff), False, acc)
    X[active[acc]] = q[acc]
    lnp[active[acc]] = lnp_q[acc]
    return acc, lnp_q


def run_sampler(lnprob_fn, X0, lnp0, n_steps, seed, block, a=2.0, step0=0, thin=1, region=0, walker_off=0):
    """Reference trajectory with the same counter-based draws as the HIP sampler.
    Returns chain[n_keep, W, D], lnp_chain[n_keep, W], n_accept[W]."""
    X = np.array(X0, dtype=np.float64)
    lnp = np.array(lnp0, dtype=np.float64)
    W, D = X.shape
    nacc = np.zeros(W, dtype=np.int64)
    chain, lchain = [], []
    for it in range(n_steps):
        step = step0 + it
        red, blue = split_tables(seed, step, W, block, region)
        for half in (0, 1):
            act, comp = (red, blue) if half == 0 else (blue, red)
            n = act.size
            zz = np.empty(n)
            logu = np.empty(n)
            partner = np.empty(n, dtype=np.int64)
            for i, w in enumerate(act):
                z, j, lu = draw_move(seed, step, half, int(w) + walker_off, n, a)
                zz[i], logu[i], partner[i] = z, lu, comp[j]
            acc, _ = stretch_half_step(X, lnp, act, partner, zz, logu, lnprob_fn)
            nacc[act[acc]] += 1
        if (it + 1) % thin == 0:
            chain.append(X.copy())
            lchain.append(lnp.copy())
    return np.array(chain), np.array(lchain), nacc


# --------------------------------------------------------------------------------------
# Data access without an HDF5 library (byte offsets of the contiguous f8[1000]
# datasets inside vamp_1.0/data/simba_*.h5)
# --------------------------------------------------------------------------------------
SIMBA_OFFSETS = {"velocity": 2048, "flux": 10048, "wavelength": 18048, "tau": 28096,
                 "noise": 36096, "density_col": 44096, "temp": 52096}


def read_simba_raw(path):
    buf = open(path, "rb").read()
    return {k: np.frombuffer(buf, "<f8", 1000, off).copy() for k, off in SIMBA_OFFSETS.items()}


def region_from_spectrum(wavelength, flux, noise, start, end, **kw):
    """vpspectrum.py:274-279: slice, flip so frequency ascends, Wave2freq."""
    nu = wave2freq(wavelength[start:end])[::-1]
    f = flux[start:end][::-1]
    n = noise[start:end][::-1]
    return nu, f, n


# --------------------------------------------------------------------------------------
# Region detection, literal restatement of vpspectrum.py:67-175 (loops and full-length
# np.convolve kernels as in the reference; O(N^2) -- used to pin the product's vectorised
# detector and to generate the region fixture of the q1422 spectrum)
# --------------------------------------------------------------------------------------
def compute_detection_regions_ref(wavelength, flux, noise, min_region_width=2, N_sigma=4.0, std_min=2, std_max=11):
    num_pixels = len(wavelength)
    flux_ews = [0.] * num_pixels
    noise_ews = [0.] * num_pixels
    det_ratio = [-float('inf')] * num_pixels
    for i in range(1, num_pixels - 1):                                   # :90-95
        dec = 1.0 - flux[i]
        if dec < noise[i]:
            dec = 0.0
        half = 0.5 * abs(wavelength[i - 1] - wavelength[i + 1])
        flux_ews[i] = half * dec
        noise_ews[i] = half * noise[i]
    xarr = np.array([p - (num_pixels - 1) / 2.0 for p in range(num_pixels)])   # :102
    for std in range(std_min, std_max):                                   # :105-117
        gaussian = gauss_function(xarr, 1.0, 0.0, std)
        flux_func = np.convolve(flux_ews, gaussian, 'same')
        noise_func = np.convolve(np.square(noise_ews), np.square(gaussian), 'same')
        with np.errstate(divide="ignore", invalid="ignore"):
            for i in range(1, num_pixels - 1):
                nf = 1.0 / np.sqrt(noise_func[i])
                if flux_func[i] * nf > det_ratio[i]:
                    det_ratio[i] = flux_func[i] * nf
    start = 0
    endpoints = []
    for i in range(num_pixels):                                           # :122-129
        if start == 0 and det_ratio[i] > N_sigma and flux[i] < 1.0:
            start = i
        elif start != 0 and (det_ratio[i] < N_sigma or flux[i] > 1.0):
            if (i - start) > min_region_width:
                endpoints.append([start, i])
            start = 0
    pixels, waves = [], []
    buffer = 3
    for i in range(len(endpoints)):                                       # :157-173
        s, e = endpoints[i]
        if i < (len(endpoints) - 1) and e > endpoints[i + 1][0]:
            e = endpoints[i + 1][1]
        for j in range(s, e):
            if (1.0 - flux[j]) > abs(noise[j]) * N_sigma:
                if s >= buffer:
                    s -= buffer
                if e < num_pixels - buffer:
                    e += buffer
                waves.append([wavelength[s], wavelength[e]])
                pixels.append([s, e])
                break
    return pixels, waves
