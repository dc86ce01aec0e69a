"""CPU tests (-m "not gpu"): the oracle against the committed golden vectors.

Pinning chain:  reference numpy statics + physics.py (imported once, tests/golden/make_golden.py)
 -> ref_statics.npz -> oracle/vamp_oracle.py;  scipy.special.wofz / mpmath -> wofz_grid.npz ->
C oracle and the host build of the device arithmetic.
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import vamp_oracle as vo


def _dp(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def c_oracle():
    # VAMP_ORACLE_SO: another build of the same library (tests/test_sanitizers.py: the ASan/UBSan build)
    so = os.environ.get("VAMP_ORACLE_SO") or os.path.join(ROOT, "oracle", "libvamp_oracle.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    return C.CDLL(so)


# ---- reference-computed statics ------------------------------------------------------------
def test_reference_statics_reproduced():
    g = load_golden("ref_statics.npz")
    a, c, s = g["gauss_params"]
    assert np.array_equal(vo.gauss_function(g["x"], a, c, s), g["gauss"])                 # vpfits.py:54
    assert np.array_equal(vo.gaussian_width(g["gwidth_in"]), g["gwidth"])                 # vpfits.py:88
    assert vo.chisquared(g["obs"], g["exp"], g["noise"]) == float(g["chisq"])              # vpfits.py:118
    assert vo.reduced_chisquared(g["obs"], g["exp"], g["noise"], 37) == float(g["redchisq"])
    assert np.array_equal(vo.tau2flux(g["tau"]), g["tau2flux"])                            # physics.py:105
    assert np.array_equal(vo.wave2freq(g["wave"]), g["wave2freq"])                         # physics.py:126
    assert np.array_equal(vo.freq2wave(vo.wave2freq(g["wave"])), g["freq2wave"])           # physics.py:120
    assert np.array_equal(vo.wave2red(g["wave"], 1215.67), g["wave2red"])                  # physics.py:134
    assert np.array_equal(vo.column_density(g["amp"], g["sig"]), g["coldens"])             # physics.py:15
    assert np.array_equal(vo.doppler_parameter(g["sig"], 1215.67), g["doppler"])           # physics.py:27
    assert float(g["c_light"]) == vo.C_LIGHT == 2.98e8 and float(g["sigma0"]) == vo.SIGMA0  # physics.py:3-4
    # survey-recorded spot values of the reference statics
    assert vo.gaussian_width(2.0) == 0.8493218002880191
    assert vo.chisquared(np.array([1, .9]), np.array([.95, .92]), np.array([.01, .01])) == 29.000000000000053


def test_nbz_bijection_inverts_reference_maps():
    g = load_golden("ref_statics.npz")
    N, b = g["coldens"], g["doppler"]
    nu = vo.wave2freq(1215.67 * (1 + 0.0123))
    for i in range(N.size):
        a, c, s = vo.nbz_to_native(N[i], b[i], 0.0123, 1215.67)
        assert abs(a - g["amp"][i]) <= 4e-16 * g["amp"][i] * 4
        assert abs(s - g["sig"][i]) <= 4e-16 * g["sig"][i] * 4
        assert abs(c - nu) <= 2.0   # Hz at 2.4e15


# ---- Voigt ------------------------------------------------------------------------------
def test_voigt_forms_agree():
    x = np.linspace(-30, 30, 401)
    for (L, G) in ((1.0, 2.0), (0.1, 2.0), (2.0, 0.5), (0.01, 1.0)):
        a = vo.voigt_function(x, 0.3, 1.7, L, G)
        b = vo.voigt_function_commented(x, 0.3, 1.7, L, G)
        assert np.max(np.abs(a - b) / a) < 5e-14   # two roundings of z differ (SURVEY: 1.5e-15 typical)
    # Lorentzian limit: G -> 0 gives peak amplitude_L (astropy's amplitude_L definition)
    assert abs(vo.voigt_function(np.array([0.0]), 0.0, 1.7, 1.0, 1e-6)[0] - 1.7) < 1e-5


def test_wofz_grid_scipy_vs_mpmath():
    g = load_golden("wofz_grid.npz")
    idx = g["mp_idx"]
    rel = np.abs(g["re_w"][idx] - g["mp_re_w"]) / g["mp_re_w"]
    assert rel.max() < 5e-14      # scipy's own accuracy claim is 1e-13


def test_c_oracle_wofz(c_oracle):
    g = load_golden("wofz_grid.npz")
    x, y, w = g["x"], g["y"], g["re_w"]
    out = np.empty_like(x)
    c_oracle.vo_wofz_re(C.c_int64(x.size), _dp(x), _dp(y), _dp(out))
    ok = w > 1e-300
    assert np.max(np.abs(out[ok] - w[ok]) / w[ok]) < 1e-13
    assert np.max(np.abs(out[g["mp_idx"]] - g["mp_re_w"]) / g["mp_re_w"]) < 2e-14


def test_device_arithmetic_host_build():
    """voigt_math.hpp compiled for the host (tests/host): the same source the kernels inline."""
    so = os.path.join(ROOT, "tests", "host", "libvoigt_host.so")
    if not os.path.exists(so):
        import __graft_entry__ as ge
        ge.build()
    lib = C.CDLL(so)
    g = load_golden("wofz_grid.npz")
    x, y, w = g["x"], g["y"], g["re_w"]
    out = np.empty_like(x)
    lib.voigt_H_host(C.c_int64(x.size), _dp(x), _dp(y), _dp(out))
    ok = w > 1e-300
    assert not np.isnan(out).any()
    assert np.max(np.abs(out[ok] - w[ok]) / w[ok]) < 1e-13           # vs scipy
    assert np.max(np.abs(out[g["mp_idx"]] - g["mp_re_w"]) / g["mp_re_w"]) < 2e-14   # vs mpmath
    xf, yf = x.astype(np.float32), y.astype(np.float32)
    of = np.empty_like(xf)
    lib.humlicek_w4_host(C.c_int64(x.size), _dp(xf), _dp(yf), _dp(of))
    from scipy.special import wofz
    wf = wofz(xf.astype(np.float64) + 1j * yf.astype(np.float64)).real
    m = (yf > 1e-6) & (wf > 1e-30)
    assert np.max(np.abs(of[m] - wf[m]) / wf[m]) < 2e-4               # Humlicek's stated ~1e-4 + fp32


def test_taylor_tables_host_build():
    """Per-line Taylor tables of the near-axis zone (voigt_math.hpp, used by the 4-wavefront
    workgroup kernels): centre values (real and imaginary part of sqrt(pi) w) and table
    evaluations against mpmath at 40 digits."""
    import mpmath as mp
    so = os.path.join(ROOT, "tests", "host", "libvoigt_host.so")
    if not os.path.exists(so):
        import __graft_entry__ as ge
        ge.build()
    lib = C.CDLL(so)
    mp.mp.dps = 40
    ys = np.array([1e-12, 1e-6, 0.01, 0.3, 1.0, 4.4, 4.6, 7.9])
    idx = np.tile(np.arange(16, dtype=np.int32), len(ys))
    yy = np.repeat(ys, 16)
    re, im = np.empty(len(yy)), np.empty(len(yy))
    lib.core_centre_host(C.c_int64(len(yy)), _dp(idx), _dp(yy), _dp(re), _dp(im))
    for i, (k, y) in enumerate(zip(idx, yy)):
        z = mp.mpc((int(k) + 0.5) / 2, float(y))
        w = mp.sqrt(mp.pi) * mp.erfc(-1j * z) * mp.exp(-z * z)
        # (the imaginary part only seeds the coefficient recurrence: 3 ulp of its O(1) value)
        assert abs(re[i] - float(w.real)) < 5e-16 and abs(im[i] - float(w.imag)) < 1e-15, (k, y)
    rng = np.random.default_rng(1)
    for y in (1e-4, 0.04, 0.3, 1.0, 3.0, 7.0):
        x = np.sort(rng.uniform(0, np.sqrt(64 - y * y) - 1e-9, 300))
        x[0], x[-1] = 0.0, np.sqrt(64 - y * y) - 1e-9
        out = np.empty_like(x)
        lib.voigt_H_table_host(C.c_int64(len(x)), _dp(x), _dp(np.full_like(x, y)), _dp(out))
        ref = np.array([float(mp.re(mp.erfc(-1j * mp.mpc(a, y)) * mp.exp(-mp.mpc(a, y) ** 2))) for a in x])
        err = np.abs(out - ref)
        assert err.max() < 4e-16 and (err / ref).max() < 3e-14, y


def test_taylor_tables32_host_build():
    """The single-precision Taylor rows that replace Humlicek's regions III / IV in the line cores of
    fp32 contexts (voigt_math.hpp, TAB32_*): against scipy's wofz the absolute error stays below
    1e-7 of the line centre for dampings from 1e-8 to 7 (W4: 3e-5) -- W4 itself is allowed 2e-4 relative (SURVEY
    8d) and reaches ~1e-4 -- and the relative error below 3e-7 wherever H > 1e-2."""
    from scipy.special import wofz
    so = os.path.join(ROOT, "tests", "host", "libvoigt_host.so")
    import __graft_entry__ as ge
    ge.build()
    lib = C.CDLL(so)
    rng = np.random.default_rng(2)
    worst_abs = worst_rel = 0.0
    for y in (1e-8, 1e-4, 0.04, 0.1, 0.3, 1.0, 3.0, 7.0):
        x = np.sort(rng.uniform(0, np.sqrt(64 - y * y) - 1e-4, 2000))
        x[0], x[-1] = 0.0, np.sqrt(64 - y * y) - 1e-4
        x = x.astype(np.float32).astype(np.float64)               # the kernel sees fp32 abscissae
        out = np.empty_like(x)
        lib.voigt_H_table32_host(C.c_int64(len(x)), _dp(x), _dp(np.full_like(x, y)), _dp(out))
        ref = wofz(x + 1j * y).real
        err = np.abs(out - ref)
        worst_abs = max(worst_abs, err.max())
        big = ref > 1e-2
        worst_rel = max(worst_rel, (err[big] / ref[big]).max())
    assert worst_abs < 1e-7 and worst_rel < 3e-7, (worst_abs, worst_rel)


# ---- log-posterior -----------------------------------------------------------------------
def _region_of(g, name):
    K = int(name.split("_K")[1].split("_")[0])
    mode = int(name.split("_m")[1].split("_")[0])
    sd = bool(int(name.split("_sd")[1]))
    r = vo.Region(x=g[name + "_x"], flux=g[name + "_flux"], noise=g[name + "_noise"], n_comp=K, mode=mode, sample_sd=sd)
    if mode == vo.MODE_NBZ3:
        r.l_fixed, r.line, r.x_origin, r.x_scale = [float(v) for v in g[name + "_nbz"]]
    return r


def test_lnprob_golden_regression_and_fast_path():
    g = load_golden("lnprob_cases.npz")
    for name in g["cases"]:
        name = str(name)
        r = _region_of(g, name)
        th = g[name + "_theta"]
        lnp, chi2 = vo.log_prob_batch(r, th, return_chi2=True)
        want = g[name + "_lnprob"]
        fin = np.isfinite(want)
        assert np.array_equal(fin, np.isfinite(lnp)), name
        assert np.allclose(lnp[fin], want[fin], rtol=1e-13, atol=0), name
        fast = vo.log_prob_batch_fast(r, th)
        assert np.array_equal(np.isfinite(fast), fin), name
        assert np.allclose(fast[fin], want[fin], rtol=1e-11, atol=1e-9), name
        # every fixture holds the documented edge cases
        assert (~fin).sum() >= 3, name


def test_lnprob_decomposition():
    """lnprob = log-prior + log-like; chi^2 equals the reference-form Chisquared on the model."""
    g = load_golden("lnprob_cases.npz")
    name = "H1215_r0_K4_m1_sd0"
    r = _region_of(g, name)
    th = g[name + "_theta"][0]
    m = vo.model_flux(r, th)
    chi = vo.chisquared(r.flux, m, r.noise)
    lnp, chi2 = vo.log_prob(r, th, return_chi2=True)
    assert abs(chi - chi2) <= 1e-12 * chi
    assert abs(lnp - (vo.log_prior(r, th) - 0.5 * chi2)) <= 1e-12 * abs(lnp)
    assert np.allclose(np.exp(-g[name + "_tau0"].sum(0)), g[name + "_flux0"], rtol=1e-15)


def test_c_oracle_lnprob(c_oracle):
    g = load_golden("lnprob_cases.npz")
    for name in g["cases"]:
        name = str(name)
        r = _region_of(g, name)
        th = np.ascontiguousarray(g[name + "_theta"])
        W = th.shape[0]
        out, chi = np.empty(W), np.empty(W)
        nbz = np.array([r.l_fixed, r.line, r.x_origin, r.x_scale])
        rc = c_oracle.vo_lnprob(C.c_int64(r.x.size), _dp(r.x), _dp(r.flux), _dp(r.noise), r.n_comp, r.mode,
                                int(r.sample_sd), 0, None, _dp(nbz), C.c_int64(W), _dp(th), _dp(out), _dp(chi), 2)
        assert rc == 0
        want = g[name + "_lnprob"]
        fin = np.isfinite(want)
        assert np.array_equal(fin, np.isfinite(out)), name
        assert np.max(np.abs(out[fin] - want[fin]) / np.maximum(1, np.abs(want[fin]))) < 1e-9, name


def test_simba_region_kats():
    """Notebook-pinned inputs of the path: region pixels <-> wavelengths, degrees of freedom
    (simba_spec_demo.ipynb cells 9, 15, 23, 25) and the first wavelengths (cell 6)."""
    g = load_golden("simba_spectra.npz")
    assert np.allclose(g["CII1036_wavelength"][:2], [1036.34564212, 1036.36292636], rtol=0, atol=5e-9)
    for tag in ("H1215", "CII1036"):
        wl = g[f"{tag}_wavelength"]
        px = g[f"{tag}_region_pixels"]
        assert np.array_equal(wl[px], g[f"{tag}_region_waves"])
        # dof = num_pixels - 3 n with n = 1 (vpregion.py:37-39)
        assert np.array_equal((px[:, 1] - px[:, 0]) - 3, g[f"{tag}_dof_n1"])
    nu, f, n = vo.region_from_spectrum(g["H1215_wavelength"], g["H1215_flux"], g["H1215_noise"], 672, 716)
    assert nu.size == 44 and np.all(np.diff(nu) > 0)          # vpspectrum.py:274-277: ascending frequency
    assert np.all(n == 0.01)


# ---- RNG + stretch move --------------------------------------------------------------------
def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10 (kat_vectors of the Random123 distribution)."""
    assert vo.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert vo.philox4x32_10((0xffffffff,) * 4, (0xffffffff,) * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert vo.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_split_is_balanced_permutation():
    for block in (2, 8, 16, 100, 1024):
        for step in (0, 1, 7):
            p = [vo.split_perm(12345, step, 3, s, block) for s in range(block)]
            assert sorted(p) == list(range(block))
    red, blue = vo.split_tables(99, 4, 64, 16)
    assert sorted(np.concatenate([red, blue])) == list(range(64)) and red.size == blue.size == 32
    # membership changes from step to step
    red2, _ = vo.split_tables(99, 5, 64, 16)
    assert not np.array_equal(np.sort(red), np.sort(red2))


def test_vectorised_draws_equal_the_scalar_restatement():
    """draw_move_batch / split_perm_batch / run_sampler_batch (numpy on uint64 arrays, used to replay
    production-size ensembles) against the scalar functions the golden trajectories were made with."""
    seed = 0x1234ABCD5678EF01
    gids = np.array([0, 1, 77, 65535, 2**32 + 5, 2**40 + 123], dtype=np.uint64)
    for step, half, n in ((0, 0, 32768), (3, 1, 17), (2**31 + 9, 1, 2**20)):
        z, j, lu = vo.draw_move_batch(seed, step, half, gids, n)
        for i, g in enumerate(gids):
            z1, j1, lu1 = vo.draw_move(seed, step, half, int(g), n)
            assert z[i] == z1 and j[i] == j1 and lu[i] == lu1
    for block in (8, 16, 100, 1024):
        ch = np.repeat(np.arange(3), block)
        sl = np.tile(np.arange(block), 3)
        got = vo.split_perm_batch(seed, 5, ch, sl, block, region=2)
        want = [vo.split_perm(seed, 5, int(c), int(s_), block, region=2) for c, s_ in zip(ch, sl)]
        assert np.array_equal(got, want)
    g = load_golden("stretch_traj.npz")
    region = vo.Region(x=g["x"], flux=g["flux"], noise=g["noise"], n_comp=1, mode=vo.MODE_VOIGT4)
    fn = lambda q: vo.log_prob_batch(region, q)
    for blk in (8, 16):
        chain, lchain, nacc = vo.run_sampler_batch(fn, g["X0"], g["lnp0"], 12, seed=seed, block=blk)
        assert np.array_equal(chain, g[f"philox_chain_b{blk}"]) and np.array_equal(nacc, g[f"philox_nacc_b{blk}"])


def test_stretch_golden_trajectories():
    g = load_golden("stretch_traj.npz")
    r = vo.Region(x=g["x"], flux=g["flux"], noise=g["noise"], n_comp=1, mode=vo.MODE_VOIGT4)
    fn = lambda q: vo.log_prob_batch(r, q)
    X, lnp = g["X0"].copy(), g["lnp0"].copy()
    for i in range(g["active"].shape[0]):
        vo.stretch_half_step(X, lnp, g["active"][i], g["partner"][i], g["zz"][i], g["logu"][i], fn)
        assert np.allclose(X, g["X_after"][i], rtol=1e-13)
        assert np.allclose(lnp, g["lnp_after"][i], rtol=1e-12)
    chain, lchain, nacc = vo.run_sampler(fn, g["X0"], g["lnp0"], 12, seed=0x1234ABCD5678EF01, block=8)
    assert np.allclose(chain, g["philox_chain_b8"], rtol=1e-13)
    assert np.array_equal(nacc, g["philox_nacc_b8"])
    assert 0 < nacc.sum() < 12 * 16


def test_stretch_move_samples_a_gaussian():
    """Statistical check of the move itself: 2-D correlated Gaussian target, mean and covariance
    recovered (detailed balance / the (D-1) log z factor)."""
    cov = np.array([[1.0, 0.6], [0.6, 2.0]])
    icov = np.linalg.inv(cov)
    fn = lambda q: -0.5 * np.einsum("wi,ij,wj->w", q, icov, q)
    rng = np.random.default_rng(3)
    W = 64
    X0 = rng.normal(size=(W, 2))
    chain, _, nacc = vo.run_sampler(fn, X0, fn(X0), 1500, seed=42, block=16)
    s = chain[300:].reshape(-1, 2)
    assert np.all(np.abs(s.mean(0)) < 0.08)
    assert np.allclose(np.cov(s.T), cov, atol=0.15)
    acc = nacc.mean() / 1500
    assert 0.5 < acc < 0.9          # D = 2, a = 2: acceptance ~0.7


def test_c_oracle_sampler_matches_python(c_oracle):
    g = load_golden("stretch_traj.npz")
    r = vo.Region(x=g["x"], flux=g["flux"], noise=g["noise"], n_comp=1, mode=vo.MODE_VOIGT4)
    X = np.ascontiguousarray(g["X0"].copy())
    lnp = g["lnp0"].copy()
    nacc = np.zeros(16, dtype=np.int64)
    rc = c_oracle.vo_sampler_run(C.c_int64(r.x.size), _dp(r.x), _dp(r.flux), _dp(r.noise), 1, 1, 0, 0, None, None,
                                 C.c_int64(16), _dp(X), _dp(lnp), _dp(nacc), C.c_int64(12), C.c_int64(0),
                                 C.c_uint64(0x1234ABCD5678EF01), C.c_double(2.0), C.c_int32(16), 2)
    assert rc == 0
    assert np.allclose(X, g["philox_chain_b16"][-1], rtol=1e-10)
    assert np.array_equal(nacc, g["philox_nacc_b16"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/vamp_1.0"), reason="build container only: needs /root/reference")
def test_committed_fixtures_regenerate_bit_for_bit(tmp_path):
    """tests/golden/make_golden.py, run as committed, reproduces every committed fixture: same
    keys, dtypes, shapes and bytes (the .npz containers themselves carry zip timestamps)."""
    import filecmp
    import subprocess
    import sys
    gold = os.path.join(ROOT, "tests", "golden")
    subprocess.check_call([sys.executable, os.path.join(gold, "make_golden.py"), str(tmp_path)],
                          stdout=subprocess.DEVNULL)
    made = sorted(os.listdir(str(tmp_path)))
    assert made == sorted(f for f in os.listdir(gold) if f.endswith((".npz", ".h5")))
    for f in made:
        a, b = os.path.join(str(tmp_path), f), os.path.join(gold, f)
        if not f.endswith(".npz"):
            assert filecmp.cmp(a, b, shallow=False), f
            continue
        x, y = np.load(a, allow_pickle=False), np.load(b, allow_pickle=False)
        assert set(x.files) == set(y.files), f
        for k in x.files:
            assert x[k].dtype == y[k].dtype and x[k].shape == y[k].shape and x[k].tobytes() == y[k].tobytes(), (f, k)
