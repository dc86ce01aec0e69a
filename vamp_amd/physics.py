"""Unit maps between fit parameters and physical line parameters.

Same public names and argument meaning as the reference's ``vamp_1.0/physics.py`` (so that code
written against ``from physics import *`` keeps working), same constants -- including the
reference's ``c = 2.98e8`` m/s (physics.py:3; not 2.998e8) and ``sigma0 = 0.0263`` (physics.py:4)
-- because the frequency axis the kernels see is defined through them.  Host-side numpy: these
run once per region, outside the hot path.
"""
import numpy as np

C_LIGHT = 2.98e8      # m/s, the reference's value
SIGMA0 = 0.0263       # cm^2/s
ANGSTROM = 1.0e-10    # m
FWHM_PER_SIGMA_REF = 2.355   # the rounded factor used by DopplerParameter (physics.py:27)

constants = {"c": {"value": C_LIGHT, "units": "m/s", "def": "Speed of light in a vacuum"},
             "sigma0": {"value": SIGMA0, "units": "cm**2 / s", "def": "Cross section for absorption"}}


def Wave2freq(wavelength):
    """lambda [Angstrom] -> nu [Hz]  (physics.py:122-126)"""
    return C_LIGHT / (np.asarray(wavelength) * ANGSTROM)


def Freq2wave(frequency):
    """nu [Hz] -> lambda [Angstrom]  (physics.py:116-120)"""
    return (C_LIGHT / np.asarray(frequency)) / ANGSTROM


def Wave2red(wave, rest_wave):
    """redshift of an observed wavelength (physics.py:128-134)"""
    return (np.asarray(wave) - rest_wave) / rest_wave


def Tau2flux(tau):
    """normalised flux of an optical depth (physics.py:98-105)"""
    return np.exp(-np.asarray(tau))


def Flux2tau(flux):
    """optical depth of a normalised flux (physics.py:107-114)"""
    return -1 * np.log(flux)


def ColumnDensity(amplitude, sigma):
    """N [cm^-2] of a Gaussian tau-profile with peak ``amplitude`` and width ``sigma`` [Hz]
    (physics.py:6-15)"""
    return amplitude * sigma * np.sqrt(2 * np.pi) / SIGMA0


def DopplerParameter(sigma, line):
    """b [km/s] from the frequency-space width sigma [Hz] and rest wavelength ``line`` [Angstrom]
    (physics.py:17-27)"""
    rest_m = line * ANGSTROM
    return (rest_m * sigma * FWHM_PER_SIGMA_REF / np.sqrt(2)) * 1.0e-3


def EquivalentWidthTau(taus, edges):
    """sum of the flux decrement times the bin width (physics.py:29-42)"""
    width = np.abs(edges[-1] - edges[0]) / (len(edges) - 1)
    return np.sum(1 - np.exp(-1 * np.asarray(taus))) * width


def EquivalentWidthFlux(fluxes, edges):
    """as EquivalentWidthTau, from fluxes (physics.py:45-58)"""
    width = np.abs(edges[-1] - edges[0]) / (len(edges) - 1)
    return np.sum((1 - np.asarray(fluxes)) * width)


def ErrorB(std_s, line):
    """sigma_b from sigma_sigma (physics.py:61-69)"""
    return DopplerParameter(std_s, line)


def ErrorN(amplitude, sigma, std_a, std_s, cov_as):
    """sigma_N from the amplitude and width errors; the covariance term is ignored, as in the
    reference (physics.py:71-87)"""
    pref = np.sqrt(2.0 * np.pi) / SIGMA0
    return pref * np.sqrt(np.asarray(sigma) ** 2 * np.asarray(std_a) ** 2 +
                          np.asarray(amplitude) ** 2 * np.asarray(std_s) ** 2)


def Errorl(std_f):
    """error on the line position (physics.py:90-96)"""
    return C_LIGHT * np.asarray(std_f) / ANGSTROM


def NativeFromNbz(N, b, z, line):
    """Inverse of ColumnDensity / DopplerParameter / (Freq2wave, Wave2red):
    (N, b, z) -> (amplitude, centroid [Hz], sigma [Hz]).  Not in the reference; it is the map the
    VAMP_NBZ3 kernels apply on device."""
    sigma = np.asarray(b) * 1.0e3 * np.sqrt(2.0) / (FWHM_PER_SIGMA_REF * (line * ANGSTROM))
    amplitude = np.asarray(N) * SIGMA0 / (sigma * np.sqrt(2.0 * np.pi))
    centroid = C_LIGHT / (line * (1.0 + np.asarray(z)) * ANGSTROM)
    return amplitude, centroid, sigma
