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


def _fit_pair(seed, nwalkers):
    """The product's VPfit twice: on the HIP library, and with the CPU oracle injected as its
    context (tests/oracle_ctx.py) -- same host logic, same seeds, independent arithmetic."""
    from oracle_ctx import OracleContext
    from vamp_amd.vpfits import VPfit
    g, o = VPfit(seed=seed), VPfit(seed=seed)
    o._ctx = OracleContext()
    g.nwalkers = o.nwalkers = nwalkers
    return g, o


@pytest.mark.parametrize("voigt,n", [(False, 1), (True, 2)])
def test_find_bic_and_chain_covariance_match_oracle(voigt, n):
    """find_bic (vpfits.py:398-429) against the oracle: three {model, ensemble run, MAP} repeats of 20
    steps each, short enough that the HIP and the oracle chains coincide walker by walker (same
    counter-based draws; a rounding-level difference would have to flip an accept decision).  The
    oracle side samples with vo.run_sampler and finds the MAP with scipy's own fmin, which is what
    PyMC's MAP.fit runs.  bic_array / red_chi_array agree to 1e-9, the kept chains to 1e-9, and
    chain_covariance equals np.cov of the oracle's chain (vpfits.py:432-456)."""
    nu, flux, noise = _hi_region(2)
    g, o = _fit_pair(seed=3, nwalkers=32)
    for fit in (g, o):
        fit.find_bic(nu, flux, n, noise, nu.size - 3 * n, voigt=voigt, iterations=20, thin=1, burn=5)
    assert len(g.bic_array) == 3 and len(g.red_chi_array) == 3
    assert np.allclose(g._chain_dev, o._chain_dev, rtol=1e-9, atol=1e-12)
    assert np.allclose(g.bic_array, o.bic_array, rtol=1e-9, atol=0), (g.bic_array, o.bic_array)
    assert np.allclose(g.red_chi_array, o.red_chi_array, rtol=1e-9, atol=0), (g.red_chi_array, o.red_chi_array)
    assert np.isclose(g.map.lnL, o.map.lnL, rtol=1e-9) and np.isclose(g.map.AIC, o.map.AIC, rtol=1e-9)
    assert np.allclose(g.total.value, o.total.value, rtol=1e-9, atol=1e-300)
    for k in range(n):
        for key in g.estimated_variables[k]:
            assert np.isclose(g.estimated_variables[k][key].value, o.estimated_variables[k][key].value, rtol=1e-9), (k, key)
    # chain_covariance from the HIP chain == np.cov of the oracle's chain, restated here from
    # vpfits.py:432-456: rows (amplitude, sigma | GaussianWidth(G_fwhm), centroid), caller units (Hz)
    cov = g.chain_covariance(n, voigt=voigt)
    flat = o._chain_dev.reshape(-1, o._ndim)
    mid, dnu, q = 0.5 * (nu[0] + nu[-1]), (nu[-1] - nu[0]) / (nu.size - 1), 4 if voigt else 3
    for k in range(n):
        amp, cen = flat[:, q * k], flat[:, q * k + 1] * dnu + mid
        sig = flat[:, q * k + 3] * dnu / (2.0 * np.sqrt(2.0 * np.log(2.0))) if voigt else flat[:, q * k + 2] * dnu
        want = np.cov(np.array((amp, sig, cen)))
        assert np.allclose(cov[k], want, rtol=1e-7, atol=0), (k, cov[k], want)
    # the BIC is PyMC 2.3's: k ln(n_data) - 2 lnL with k free scalars (sd included)
    kfree = (4 if voigt else 3) * n + 1
    assert abs(g.map.BIC - (kfree * np.log(nu.size) - 2 * g.map.lnL)) < 1e-9


@pytest.mark.parametrize("voigt", [False, True])
def test_find_bic_batched_matches_oracle(voigt):
    """The batched find_bic (vamp_amd/batched.py: the three repeats of every region as 3 R regions of ONE
    context, vpfits.py:417-428) on the HIP library and on the oracle-backed context: the same host logic,
    the same generator and draw keys, independent arithmetic -- oracle sampler, scipy's own fmin.  Three
    H I regions with different numbers of lines, 20 steps: bic_array / red_chi_array of every region to
    1e-9, the kept fit objects' chains, MAP values and model flux likewise; every repeat has a chain of its
    own (its region index keys the draws)."""
    import vamp_amd
    from oracle_ctx import OracleContext
    from vamp_amd.batched import find_bic_batched
    regions = [_hi_region(i) for i in range(3)]
    ns = [1, 2, 1]
    out = {}
    for name, ctx in (("hip", vamp_amd.HipContext(device=0)), ("oracle", OracleContext())):
        out[name] = find_bic_batched(ctx, regions, ns, voigt=voigt, nwalkers=32, iterations=20, thin=1, burn=5, seed=17)
        assert ctx.n_regions == 9 and len(out[name]) == 3
    for r, (g, o) in enumerate(zip(out["hip"], out["oracle"])):
        assert len(g.bic_array) == 3 and len(set(g.bic_array)) == 3, g.bic_array       # three different chains
        assert np.allclose(g.bic_array, o.bic_array, rtol=1e-9, atol=0), (r, g.bic_array, o.bic_array)
        assert np.allclose(g.red_chi_array, o.red_chi_array, rtol=1e-9, atol=0), (r, g.red_chi_array, o.red_chi_array)
        fg, fo = g.detach().fit(), o.detach().fit()
        assert fg._chain_dev.shape == (15, 32, (4 if voigt else 3) * ns[r] + 1)
        assert np.allclose(fg._chain_dev, fo._chain_dev, rtol=1e-9, atol=1e-12)
        assert np.isclose(fg.map.BIC, g.bic_array[-1]) and np.isclose(fg.map.lnL, fo.map.lnL, rtol=1e-9)
        assert np.allclose(fg.total.value, fo.total.value, rtol=1e-9, atol=1e-300)
        assert fg.mcmc.DIC is None and len(fg.estimated_profiles) == ns[r]
        st = fg.mcmc.stats()
        assert np.isclose(st["xexp_0"]["standard deviation"], fo.mcmc.stats()["xexp_0"]["standard deviation"], rtol=1e-7)
        nu = regions[r][0]
        assert nu[0] <= fg.estimated_variables[0]["centroid"].value <= nu[-1]


def test_region_fit_ladder_matches_oracle(monkeypatch):
    """VPregion.region_fit (vpregion.py:42-91) on the HIP path and on the oracle: the same sequence
    of rungs, the same BIC / reduced chi^2 triples on every rung (1e-9) and the same final n."""
    import vamp_amd.vpregion as vr
    from oracle_ctx import OracleContext
    from vamp_amd.vpfits import VPfit
    nu, flux, noise = _hi_region(0)

    class OracleVPfit(VPfit):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self._ctx = OracleContext()

    def ladder(fit_cls):
        monkeypatch.setattr(vr, "VPfit", fit_cls)
        rungs = []

        class Recording(vr.VPregion):
            def _fit_n(self, n, iterations, thin, burn):
                fit = super()._fit_n(n, iterations, thin, burn)
                rungs.append((n, list(fit.bic_array), list(fit.red_chi_array)))
                return fit

        reg = Recording(nu, flux, noise, voigt=False, chi_limit=1.5, nwalkers=32, seed=21)
        reg.region_fit(verbose=False, iterations=20, thin=1, burn=5)
        return reg, rungs

    reg_g, rungs_g = ladder(VPfit)
    reg_o, rungs_o = ladder(OracleVPfit)
    assert [r[0] for r in rungs_g] == [r[0] for r in rungs_o] and len(rungs_g) >= 2
    for (n, bg, cg), (_, bo, co) in zip(rungs_g, rungs_o):
        assert np.allclose(bg, bo, rtol=1e-9, atol=0), (n, bg, bo)
        assert np.allclose(cg, co, rtol=1e-9, atol=0), (n, cg, co)
    assert reg_g.n == reg_o.n and len(reg_g.fit.estimated_profiles) == reg_g.n
    assert np.allclose(reg_g.fit.total.value, reg_o.fit.total.value, rtol=1e-9, atol=1e-300)
    # a retry is a new attempt with fresh draws, not a replay (vpspectrum.py:297-348)
    reg_g.estimate_n()
    first = rungs_g[0]
    rungs_g.clear()
    monkeypatch.setattr(vr, "VPfit", VPfit)
    reg_g.region_fit(verbose=False, iterations=20, thin=1, burn=5)
    assert rungs_g[0][0] == first[0] and rungs_g[0][1] != first[1]


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
