"""GPU tests of the drop-in surface: vamp_amd.vpfits.VPfit used the way the reference's callers
use vpfits.VPfit (vpregion.py:59-91, vpspectrum.py:335-426, vpfits_intro.ipynb cells 13-25)."""
import copy

import numpy as np
import pytest

from conftest import load_golden
from oracle import vamp_oracle as vo

pytestmark = pytest.mark.gpu


def _hi_region(i=0):
    g = load_golden("simba_spectra.npz")
    s, e = g["H1215_region_pixels"][i]
    return vo.region_from_spectrum(g["H1215_wavelength"], g["H1215_flux"], g["H1215_noise"], s, e)


def test_config1_single_component_fit():
    """BASELINE.json config 1: H I region [672,716], 1 component, 100 walkers x 500 steps, and the
    same posterior sampled by the CPU oracle (numpy + scipy.wofz + restated stretch move)."""
    from vamp_amd.vpfits import VPfit
    nu, flux, noise = _hi_region(0)
    fit = VPfit(noise=noise, seed=11)
    fit.nwalkers = 100
    fit.initialise_model(nu, flux, 1, voigt=True)
    fit.map_estimate()
    assert np.isfinite(fit.map.BIC) and np.isfinite(fit.map.AIC)
    fit.mcmc_fit(iterations=500, burnin=200, thinning=5)
    tr = {nm: fit.mcmc.trace(nm)[:] for nm in ("xexp_0", "est_centroid_0", "est_L_0", "est_G_0")}
    assert all(t.shape == (60 * 100,) for t in tr.values())
    st = fit.mcmc.stats()
    assert {"xexp_0", "est_centroid_0", "est_L_0", "est_G_0", "est_sigma_0"} <= set(st)
    assert nu[0] <= tr["est_centroid_0"].min() and tr["est_centroid_0"].max() <= nu[-1]      # Hz, inside the prior
    assert 0.1 < fit.mcmc.acceptance_fraction < 0.9
    assert fit.total.value.shape == flux.shape and len(fit.estimated_profiles) == 1
    assert np.allclose(fit.total.value, np.exp(-fit.estimated_profiles[0].value), rtol=1e-13)
    # the same posterior through the CPU oracle (numpy + scipy.wofz + restated stretch move), started
    # from the same ensemble with the same counter-based draws: the two chains agree walker by
    # walker until a rounding-level difference flips an accept decision, and statistically after
    mid, dnu = 0.5 * (nu[0] + nu[-1]), (nu[-1] - nu[0]) / (nu.size - 1)
    reg = vo.Region(x=(nu - mid) / dnu, flux=flux, noise=noise, n_comp=1, mode=vo.MODE_VOIGT4)
    rng = np.random.default_rng(5)
    X0 = fit._theta_dev * (1 + 1e-2 * rng.standard_normal((100, 4)))
    fn = lambda q: vo.log_prob_batch_fast(reg, q)
    chain, lchain, nacc = vo.run_sampler(fn, X0, fn(X0), 500, seed=77, block=100)
    fit._ctx.sampler_init(X0, seed=77, a=2.0, split_block=100)
    res = fit._ctx.run(500)
    same = np.all(np.abs(res["chain"] - chain) <= 1e-9 * np.abs(chain) + 1e-12, axis=2)   # [step, walker]
    assert same[:20].all(), "the first 20 steps must match the oracle walker by walker"
    ref, got = chain[200:].reshape(-1, 4), res["chain"][200:].reshape(-1, 4)
    sd = ref.std(0)
    assert np.all(np.abs(got.mean(0) - ref.mean(0)) < 0.2 * sd), (got.mean(0), ref.mean(0), sd)
    assert np.all(np.abs(got.std(0) / sd - 1) < 0.25)
    assert abs(res["n_accept"].sum() - nacc.sum()) < 0.05 * nacc.sum()
    # the MAP model is a decent fit of this single-line region
    chi_r = VPfit.ReducedChisquared(flux, fit.total.value, noise, nu.size - 3)
    assert chi_r < 0.2 * VPfit.ReducedChisquared(flux, np.ones_like(flux), noise, nu.size - 3)   # one line cannot fit this blend well


def test_find_bic_reference_flow():
    """VPregion's use (vpregion.py:59-72): VPfit().find_bic(...) with the free-sd likelihood,
    bic_array / red_chi_array of length 3, copy.copy(fit), attribute reads of vpspectrum.py:351-412."""
    from vamp_amd.vpfits import VPfit
    nu, flux, noise = _hi_region(2)
    fit = VPfit(seed=3)
    fit.nwalkers = 32
    fit.find_bic(nu, flux, 1, noise, nu.size - 3, voigt=False, iterations=300, thin=5, burn=100)
    assert len(fit.bic_array) == 3 and len(fit.red_chi_array) == 3
    assert np.all(np.isfinite(fit.bic_array)) and np.all(np.isfinite(fit.red_chi_array))
    old = copy.copy(fit)
    assert old.total.value is fit.total.value
    heights = [fit.estimated_variables[i]['amplitude'].value for i in range(1)]
    sig = [fit.estimated_variables[i]['sigma'].value for i in range(1)]
    cen = [fit.estimated_variables[i]['centroid'].value for i in range(1)]
    assert heights[0] > 0 and 0 < sig[0] <= fit.sigma_max and nu[0] <= cen[0] <= nu[-1]
    stats = fit.mcmc.stats()
    assert stats['est_sigma_0']['standard deviation'] > 0 and stats['xexp_0']['standard deviation'] > 0
    assert 'sd' in stats
    cov = fit.chain_covariance(1, voigt=False)
    assert cov.shape == (1, 3, 3) and np.all(np.diag(cov[0]) > 0)
    assert np.isfinite(fit.mcmc.DIC) and np.isfinite(fit.mcmc.BPIC)
    # BIC definition: k ln(n) - 2 lnL with k = 3n + 1 free scalars (sd included)
    assert abs(fit.map.BIC - (4 * np.log(nu.size) - 2 * fit.map.lnL)) < 1e-9


def test_voigt_function_static_matches_oracle():
    from vamp_amd.vpfits import VPfit
    x = np.linspace(2.4e15, 2.4e15 + 2e12, 257)
    got = VPfit.VoigtFunction(x, 2.4e15 + 9e11, 1.3, 3e10, 8e10)
    want = vo.voigt_function(x, 2.4e15 + 9e11, 1.3, 3e10, 8e10)
    assert np.max(np.abs(got - want) / want) < 1e-9      # x re-centring costs a few ulp of (x - c)
    assert VPfit.GaussianWidth(2.0) == 0.8493218002880191


@pytest.mark.gpu
def test_integration_md_stub_runs():
    """The ctypes stub printed in INTEGRATION.md section 2 (what a reference maintainer would add)
    is executed as written, against the in-tree library, on a simba H I region."""
    import os
    import re
    from conftest import ROOT
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\nimport ctypes as C, numpy as np\n(.*?)```", text, re.S).group(0)
    code = block[len("```python\n"):-3].replace('"libvamp_hip.so"', repr(os.path.join(ROOT, "vamp_amd", "libvamp_hip.so")))
    g = load_golden("simba_spectra.npz")
    s, e = g["H1215_region_pixels"][0]
    freq, flux, noise = vo.region_from_spectrum(g["H1215_wavelength"], g["H1215_flux"], g["H1215_noise"], int(s), int(e))
    n, W, seed, iters, thin = 1, 32, 5, 60, 2
    rng = np.random.default_rng(0)
    best_theta = np.array([2.0, 0.0, 3.0, 8.0])
    start_ball = best_theta * (1.0 + 0.01 * rng.standard_normal((W, 4)))
    env = dict(freq=np.ascontiguousarray(freq), flux=np.ascontiguousarray(flux), noise=np.ascontiguousarray(noise), n=n, W=W,
               seed=seed, iters=iters, thin=thin, start_ball=start_ball, best_theta=best_theta)
    exec(compile(code, "INTEGRATION.md", "exec"), env)
    assert env["chain"].shape == (30, 32, 4) and np.isfinite(env["chain"]).all() and np.isfinite(env["lnp"]).all()
    assert env["nacc"].sum() > 0 and env["secs"].value > 0
    assert env["tau"].shape == (1, freq.size) and np.allclose(env["model"], np.exp(-env["tau"][0]))
