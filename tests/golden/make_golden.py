#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Run in the BUILD container only (it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

What is reference-computed and what is oracle-computed:

* ``ref_statics.npz``  -- outputs of the REFERENCE's own numpy-only functions, imported from
  /root/reference/vamp_1.0: ``physics.py`` (imports cleanly) and the statics of ``vpfits.VPfit``
  (``GaussFunction``, ``GaussianWidth``, ``Chisquared``, ``ReducedChisquared``).  ``vpfits.py``
  imports pymc and astropy at module level (absent here, ordinary ModuleNotFoundError); empty
  placeholder modules are inserted in ``sys.modules`` only so that the import statement succeeds --
  they supply no arithmetic and none of the recorded functions touches them.
* ``simba_spectra.npz`` -- the wavelength/flux/noise datasets of the two simba HDF5 data files
  (raw contiguous f8 reads at the byte offsets of SURVEY section 8d) plus the notebook-pinned
  detection regions (simba_spec_demo.ipynb cells 9, 15, 23, 25).
* ``wofz_grid.npz`` -- Re w(z) from scipy.special.wofz on a grid spanning every branch of the
  HIP evaluator, with an mpmath cross-check column.
* ``lnprob_cases.npz`` -- oracle (oracle/vamp_oracle.py) tau_k / flux / chi^2 / log-prior /
  log-posterior for seeded random parameter vectors on the simba regions and a down-scaled
  synthetic twin of the headline workload.
* ``stretch_traj.npz`` -- oracle stretch-move trajectories: one with recorded (injected) draws,
  one with the counter-based Philox draws shared with the HIP sampler.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE      # where the fixtures are written (tests/test_oracle.py regenerates into a temp dir)
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/vamp_1.0"

from oracle import vamp_oracle as vo  # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    import physics as ref_physics  # noqa
    # placeholders so that `import pymc as mc` / `from astropy.modeling.models import Voigt1D`
    # at vpfits.py:16,20 do not abort the import; they carry no behaviour.
    pymc = types.ModuleType("pymc")
    pymc.AdaptiveMetropolis = object           # default argument at vpfits.py:361
    pymc.Uniform = lambda *a, **k: 1.0         # evaluated nowhere in the recorded statics
    astropy = types.ModuleType("astropy")
    modeling = types.ModuleType("astropy.modeling")
    models = types.ModuleType("astropy.modeling.models")
    models.Voigt1D = None
    for name, mod in (("pymc", pymc), ("astropy", astropy), ("astropy.modeling", modeling),
                      ("astropy.modeling.models", models)):
        sys.modules.setdefault(name, mod)
    import matplotlib
    matplotlib.use("Agg")
    import vpfits as ref_vpfits  # noqa
    return ref_physics, ref_vpfits


def make_ref_statics():
    ref_physics, ref_vpfits = import_reference()
    V = ref_vpfits.VPfit
    rng = np.random.default_rng(101)
    x = np.linspace(-7.5, 9.25, 67)
    amp, cen, sig = 1.7, 0.4, 1.3
    obs = rng.uniform(0, 1, 40)
    exp_ = rng.uniform(0, 1, 40)
    noise = rng.uniform(0.005, 0.02, 40)
    tau = rng.uniform(0, 6, 33)
    wave = np.linspace(1215.0, 1236.0, 29)
    a = rng.uniform(0.1, 3, 9)
    s = rng.uniform(1e9, 5e10, 9)
    out = dict(
        x=x, gauss_params=np.array([amp, cen, sig]),
        gauss=V.GaussFunction(x, amp, cen, sig),
        gwidth_in=np.array([2.0, 0.3, 1e11]),
        gwidth=np.array([V.GaussianWidth(2.0), V.GaussianWidth(0.3), V.GaussianWidth(1e11)]),
        obs=obs, exp=exp_, noise=noise,
        chisq=np.array(V.Chisquared(obs, exp_, noise)),
        redchisq=np.array(V.ReducedChisquared(obs, exp_, noise, 37)),
        tau=tau, tau2flux=ref_physics.Tau2flux(tau), flux2tau=ref_physics.Flux2tau(np.exp(-tau)),
        wave=wave, wave2freq=ref_physics.Wave2freq(wave), freq2wave=ref_physics.Freq2wave(ref_physics.Wave2freq(wave)),
        wave2red=ref_physics.Wave2red(wave, 1215.67),
        amp=a, sig=s, coldens=ref_physics.ColumnDensity(a, s),
        doppler=ref_physics.DopplerParameter(s, 1215.67),
        errN=ref_physics.ErrorN(a, s, 0.1 * a, 0.05 * s, 0.0),
        errl=ref_physics.Errorl(s * 1e-3),
        ew_tau=np.array(ref_physics.EquivalentWidthTau(tau, wave[:33])),
        ew_flux=np.array(ref_physics.EquivalentWidthFlux(np.exp(-tau), wave[:33])),
        c_light=np.array(ref_physics.constants['c']['value']),
        sigma0=np.array(ref_physics.constants['sigma0']['value']),
    )
    np.savez(os.path.join(OUT, "ref_statics.npz"), **out)
    print("ref_statics.npz: reference-computed statics recorded")


def make_simba():
    # one of the reference's data files as it is (written by h5py: the real-file fixture of the
    # minimal HDF5 reader / writer, tests/test_h5min.py)
    import shutil
    shutil.copyfile(os.path.join(REF, "data", "simba_H1215.h5"), os.path.join(OUT, "simba_H1215.h5"))
    out = {}
    for tag, fn in (("H1215", "simba_H1215.h5"), ("CII1036", "simba_CII1036.h5")):
        d = vo.read_simba_raw(os.path.join(REF, "data", fn))
        for k in ("wavelength", "flux", "noise"):
            out[f"{tag}_{k}"] = d[k]
    # notebook-pinned detection regions (simba_spec_demo.ipynb cells 9, 23)
    out["CII1036_region_pixels"] = np.array([[677, 714], [720, 765], [774, 791], [932, 955]])
    out["H1215_region_pixels"] = np.array([[672, 716], [711, 762], [767, 796]])
    out["CII1036_region_waves"] = np.array([[1048.0470730649474, 1048.6865899703391],
                                            [1048.790295414457, 1049.5680862453387],
                                            [1049.7236444115154, 1050.017476503182],
                                            [1052.4545544399455, 1052.85209197573]])
    out["H1215_region_waves"] = np.array([[1229.3051685354806, 1230.1972771088056],
                                          [1230.095901134564, 1231.1299360718274],
                                          [1231.2313120460688, 1231.8192926966692]])
    out["CII1036_dof_n1"] = np.array([34, 42, 14, 20])   # cell 15
    out["H1215_dof_n1"] = np.array([41, 48, 26])         # cell 25
    np.savez(os.path.join(OUT, "simba_spectra.npz"), **out)
    print("simba_spectra.npz written")
    return out


def make_wofz_grid():
    import mpmath as mp
    from scipy.special import wofz
    rng = np.random.default_rng(7)
    xs = np.concatenate([[0.0, 1e-12, 1e-3, 0.24, 0.25, 0.26, 0.5, 0.75, 1.0],
                         np.linspace(0.1, 8.4, 60), [7.9, 7.99, 8.0, 8.01, 13.9, 14.0, 14.1, 24.9, 25.0, 25.1,
                                                     99.0, 100.0, 101.0, 9999.0, 1e4, 1.0001e4],
                         10.0 ** np.linspace(1, 9, 25)])
    ys = np.concatenate([[1e-300, 1e-100, 1e-30, 1e-14, 1e-12, 1e-10, 0.9e-9, 1.1e-9, 1e-8, 1e-6, 1e-4,
                          1e-3, 1e-2, 0.03, 0.1, 0.2, 0.3, 0.5, 0.83, 1.0, 2.0, 3.0, 3.9, 4.0, 4.1, 4.4, 4.5, 4.6,
                          5.0, 6.0, 7.0, 7.9, 8.0, 8.1, 10.0, 30.0, 1e2, 1e3, 1e4, 1e6]])
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    xr = 10.0 ** rng.uniform(-3, 2.5, 4000)
    yr = 10.0 ** rng.uniform(-12, 1.2, 4000)
    x = np.concatenate([X.ravel(), xr])
    y = np.concatenate([Y.ravel(), yr])
    w = wofz(x + 1j * y).real
    # mpmath cross-check on a subset (adaptive precision: Re[e^{-z^2} erfc(-iz)] cancels badly)
    idx = rng.choice(x.size, 600, replace=False)
    wm = np.empty(idx.size)
    for n, i in enumerate(idx):
        r2 = x[i] ** 2 + y[i] ** 2
        if r2 < 4000:
            mp.mp.dps = int(60 + r2 / 2)
            z = mp.mpc(x[i], y[i])
            wm[n] = float(mp.re(mp.exp(-z * z) * mp.erfc(-1j * z)))
        else:
            mp.mp.dps = 60
            z = mp.mpc(x[i], y[i])
            t = mp.mpc(0)
            for k in range(80, 0, -1):
                t = mp.mpf(k) / 2 / (z - t)
            wm[n] = float(mp.re(1j / mp.sqrt(mp.pi) / (z - t)))
    np.savez(os.path.join(OUT, "wofz_grid.npz"), x=x, y=y, re_w=w, mp_idx=idx, mp_re_w=wm)
    rel = np.abs(w[idx] - wm) / wm
    print("wofz_grid.npz: %d points; scipy vs mpmath max rel %.2e" % (x.size, rel.max()))


def scaled_region(wl, fl, no, start, end, **kw):
    """vpspectrum.py:274-279 slice + flip, then region-centred pixel-spacing units (what the
    facade hands to the HIP path)."""
    nu, f, n = vo.region_from_spectrum(wl, fl, no, start, end)
    nu_mid = 0.5 * (nu[0] + nu[-1])
    dnu = (nu[-1] - nu[0]) / (nu.size - 1)
    x = (nu - nu_mid) / dnu
    return vo.Region(x=x, flux=f, noise=n, **kw), nu, nu_mid, dnu


def random_thetas(rng, region, W, with_edges=True):
    K, q = region.n_comp, region.q
    th = np.empty((W, region.ndim))
    for k in range(K):
        th[:, q * k + 0] = rng.gamma(2.0, 1.0, W)
        th[:, q * k + 1] = rng.uniform(region.c_lo, region.c_hi, W)
        if region.mode == vo.MODE_GAUSS3:
            th[:, q * k + 2] = rng.uniform(0, region.sigma_max, W)
        else:
            th[:, q * k + 2] = rng.uniform(0, region.fwhm_max, W) * 10.0 ** rng.uniform(-3, 0, W)
            th[:, q * k + 3] = rng.uniform(0, region.fwhm_max, W) * 10.0 ** rng.uniform(-2, 0, W)
    if region.sample_sd:
        th[:, -1] = rng.uniform(0.001, 1.0, W)
    if with_edges and W >= 16:
        th[1, 0] = -0.1                                   # negative amplitude -> -inf
        th[2, 1] = region.c_hi + 1.0                      # centroid outside prior
        th[3, 0] = 0.0                                    # log(0) -> -inf
        if region.mode != vo.MODE_GAUSS3:
            th[4, 2] = region.fwhm_max * 1e-14            # tiny Lorentzian (y ~ 1e-14)
            th[5, 3] = region.fwhm_max * 1e-6             # tiny Gaussian width (huge |x|)
            th[6, 2] = region.fwhm_max * 0.999            # broad Lorentzian
            th[6, 3] = region.fwhm_max * 1e-3
            th[7, 3] = region.fwhm_max * 1.0001           # G outside
            th[8, 2] = 0.0                                # L = 0 exactly
        th[9, 0] = np.nan
        th[10, 0] = 800.0                                 # v*exp(-v) underflows -> -inf (vpfits.py:244)
    return th


def make_lnprob_cases(simba):
    rng = np.random.default_rng(20240517)
    out = {}
    cases = []
    for tag in ("H1215", "CII1036"):
        wl, fl, no = simba[f"{tag}_wavelength"], simba[f"{tag}_flux"], simba[f"{tag}_noise"]
        for ri, (s, e) in enumerate(simba[f"{tag}_region_pixels"]):
            for K, mode, sd in ((1, vo.MODE_VOIGT4, False), (4, vo.MODE_VOIGT4, False),
                                (2, vo.MODE_GAUSS3, False), (1, vo.MODE_GAUSS3, True), (2, vo.MODE_VOIGT4, True)):
                if tag == "CII1036" and (K, mode, sd) not in ((4, vo.MODE_VOIGT4, False), (2, vo.MODE_GAUSS3, False)):
                    continue
                region, nu, nu_mid, dnu = scaled_region(wl, fl, no, s, e, n_comp=K, mode=mode, sample_sd=sd)
                name = f"{tag}_r{ri}_K{K}_m{mode}_sd{int(sd)}"
                th = random_thetas(rng, region, 32)
                lnp, chi2 = vo.log_prob_batch(region, th, return_chi2=True)
                lpr = np.array([vo.log_prior(region, t) for t in th])
                out[name + "_x"] = region.x
                out[name + "_flux"] = region.flux
                out[name + "_noise"] = region.noise
                out[name + "_theta"] = th
                out[name + "_lnprob"] = lnp
                out[name + "_chi2"] = chi2
                out[name + "_lnprior"] = lpr
                # full model for the first clean walker
                t0 = th[0]
                out[name + "_tau0"] = vo.component_taus(region, t0)
                out[name + "_flux0"] = vo.model_flux(region, t0)
                cases.append(name)
    # raw-Hz case (no re-centring): H1215 region 0, K=4 Voigt
    wl, fl, no = simba["H1215_wavelength"], simba["H1215_flux"], simba["H1215_noise"]
    s, e = simba["H1215_region_pixels"][0]
    nu, f, n = vo.region_from_spectrum(wl, fl, no, s, e)
    region = vo.Region(x=nu, flux=f, noise=n, n_comp=4, mode=vo.MODE_VOIGT4)
    th = random_thetas(rng, region, 32)
    name = "H1215_r0_rawHz_K4_m1_sd0"
    out[name + "_x"], out[name + "_flux"], out[name + "_noise"] = region.x, region.flux, region.noise
    out[name + "_theta"] = th
    lnp, chi2 = vo.log_prob_batch(region, th, return_chi2=True)
    out[name + "_lnprob"], out[name + "_chi2"] = lnp, chi2
    out[name + "_lnprior"] = np.array([vo.log_prior(region, t) for t in th])
    out[name + "_tau0"] = vo.component_taus(region, th[0])
    out[name + "_flux0"] = vo.model_flux(region, th[0])
    cases.append(name)
    # NBZ3 on the same region: physical frequency axis, x in pixel units
    region, nu, nu_mid, dnu = scaled_region(wl, fl, no, s, e, n_comp=2, mode=vo.MODE_NBZ3)
    region.x_origin, region.x_scale, region.line = nu_mid, dnu, 1215.67
    region.l_fixed = 0.4
    W = 32
    th = np.empty((W, 6))
    for k in range(2):
        c = rng.uniform(region.c_lo * 1.05, region.c_hi * 1.05, W)
        G = rng.uniform(0.5, region.fwhm_max * 1.02, W)
        A = rng.gamma(2.0, 1.0, W)
        N, b, z = vo.native_to_nbz(A, nu_mid + dnu * c, G * dnu / vo.FWHM_PER_SIGMA, 1215.67)
        th[:, 3 * k:3 * k + 3] = np.stack([N, b, z], axis=1)
    name = "H1215_r0_K2_m2_sd0"
    out[name + "_x"], out[name + "_flux"], out[name + "_noise"] = region.x, region.flux, region.noise
    out[name + "_theta"] = th
    out[name + "_nbz"] = np.array([region.l_fixed, region.line, region.x_origin, region.x_scale])
    lnp, chi2 = vo.log_prob_batch(region, th, return_chi2=True)
    out[name + "_lnprob"], out[name + "_chi2"] = lnp, chi2
    out[name + "_lnprior"] = np.array([vo.log_prior(region, t) for t in th])
    out[name + "_tau0"] = vo.component_taus(region, th[0])
    out[name + "_flux0"] = vo.model_flux(region, th[0])
    cases.append(name)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "lnprob_cases.npz"), **out)
    print("lnprob_cases.npz: %d cases" % len(cases))


def make_stretch(simba):
    rng = np.random.default_rng(99)
    wl, fl, no = simba["H1215_wavelength"], simba["H1215_flux"], simba["H1215_noise"]
    s, e = simba["H1215_region_pixels"][0]
    region, *_ = scaled_region(wl, fl, no, s, e, n_comp=1, mode=vo.MODE_VOIGT4)
    W, D = 16, 4
    X0 = np.stack([rng.uniform(0.3, 1.5, W), rng.uniform(-4, 4, W), rng.uniform(0.5, 3, W), rng.uniform(2, 8, W)], 1)
    lnp0 = vo.log_prob_batch(region, X0)
    fn = lambda q: vo.log_prob_batch(region, q)
    # (a) injected draws, 10 steps
    X, lnp = X0.copy(), lnp0.copy()
    rec = dict(X0=X0, lnp0=lnp0, x=region.x, flux=region.flux, noise=region.noise)
    acts, parts, zzs, lus, Xs, lnps = [], [], [], [], [], []
    for step in range(10):
        inds = np.arange(W) % 2
        rng.shuffle(inds)
        for split in (0, 1):
            S = np.where(inds == split)[0]
            C = np.where(inds != split)[0]
            u1 = rng.uniform(size=S.size)
            zz = ((2.0 - 1.0) * u1 + 1.0) ** 2 / 2.0
            partner = C[rng.integers(0, C.size, S.size)]
            logu = np.log(rng.uniform(size=S.size))
            vo.stretch_half_step(X, lnp, S, partner, zz, logu, fn)
            acts.append(S); parts.append(partner); zzs.append(zz); lus.append(logu)
            Xs.append(X.copy()); lnps.append(lnp.copy())
    rec.update(active=np.array(acts), partner=np.array(parts), zz=np.array(zzs), logu=np.array(lus),
               X_after=np.array(Xs), lnp_after=np.array(lnps))
    # (b) counter-based draws (same generator as the HIP sampler), 12 steps, split block 8 and 16
    for blk in (8, 16):
        chain, lchain, nacc = vo.run_sampler(fn, X0, lnp0, 12, seed=0x1234ABCD5678EF01, block=blk)
        rec[f"philox_chain_b{blk}"] = chain
        rec[f"philox_lnp_b{blk}"] = lchain
        rec[f"philox_nacc_b{blk}"] = nacc
    # raw generator known answers
    rec["philox_kat_ctr"] = np.array([[0, 0, 0, 0], [1, 2, 3, 4], [0xFFFFFFFF] * 4], dtype=np.uint64)
    rec["philox_kat_key"] = np.array([[0, 0], [5, 6], [0xFFFFFFFF, 0xFFFFFFFF]], dtype=np.uint64)
    rec["philox_kat_out"] = np.array([vo.philox4x32_10(c, k) for c, k in zip(rec["philox_kat_ctr"], rec["philox_kat_key"])],
                                     dtype=np.uint64)
    rec["split_b16_s3"] = np.array([vo.split_perm(77, 3, 0, s_, 16) for s_ in range(16)])
    rec["split_b100_s5"] = np.array([vo.split_perm(77, 5, 2, s_, 100) for s_ in range(100)])
    np.savez_compressed(os.path.join(OUT, "stretch_traj.npz"), **rec)
    print("stretch_traj.npz written")


def make_q1422():
    """vamp_1.0/data/q1422.cont (49 106 rows: wavelength, velocity, flux, noise), the spectrum of
    BASELINE.json config 3, stored as exact integers: the file prints 3 / 6 decimals, and
    int/10^k reproduces the parsed double exactly (one correctly rounded division)."""
    a = np.loadtxt(os.path.join(REF, "data", "q1422.cont"))
    wl = np.rint(a[:, 0] * 1000).astype(np.int32)
    fl = np.rint(a[:, 2] * 1e6).astype(np.int32)
    no = np.rint(a[:, 3] * 1e6).astype(np.int32)
    assert np.array_equal(wl / 1000.0, a[:, 0]) and np.array_equal(fl / 1e6, a[:, 2]) and np.array_equal(no / 1e6, a[:, 3])
    px, wv = vo.compute_detection_regions_ref(a[:, 0], a[:, 2], a[:, 3])       # oracle, full-length kernels
    np.savez_compressed(os.path.join(OUT, "q1422_spectrum.npz"), wavelength_milli=wl, flux_micro=fl, noise_micro=no,
                        region_pixels=np.array(px, dtype=np.int32))
    print("q1422_spectrum.npz: %d pixels, %d regions" % (wl.size, len(px)))


def make_q1422_vpm():
    """vamp_1.0/data/q1422.vpm: the AutoVP-style line list of the SAME quasar spectrum that ships with the
    reference (539 H I lines; no reference code reads it).  Data rows: index, N [1e12 cm^-2], velocity
    [km/s], b [km/s], three error columns, a flag-like fraction, observed wavelength [A].  Stored as the
    parsed doubles: a sanity yardstick for the end-to-end fit (another code's answer, not parity)."""
    rows = []
    for ln in open(os.path.join(REF, "data", "q1422.vpm")):
        t = ln.split()
        if len(t) == 9:
            rows.append([float(v) for v in t[1:]])
    a = np.array(rows)
    assert a.shape == (539, 8)
    np.savez_compressed(os.path.join(OUT, "q1422_vpm.npz"), N12=a[:, 0], velocity=a[:, 1], b=a[:, 2], err=a[:, 3:6],
                        frac=a[:, 6], wavelength=a[:, 7])
    print("q1422_vpm.npz: %d lines" % a.shape[0])


def make_all(out_dir=None):
    """Write every fixture into ``out_dir`` (default: this directory)."""
    global OUT
    if not os.path.isdir(REF):
        sys.exit("run in the build container: /root/reference is required")
    if out_dir is not None:
        OUT = out_dir
    make_ref_statics()
    simba = make_simba()
    make_wofz_grid()
    make_lnprob_cases(simba)
    make_stretch(simba)
    make_q1422()
    make_q1422_vpm()


if __name__ == "__main__":
    make_all(sys.argv[1] if len(sys.argv) > 1 else None)
