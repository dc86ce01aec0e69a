"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and
the committed golden vectors.

Stated tolerances (fp64 / accurate-wofz path):
  Re w(z)       relative error <= 1e-13 against scipy.special.wofz wherever w > 1e-300
                (scipy's own claim is 1e-13), <= 2e-14 against the mpmath column
  tau_k, flux   relative <= 1e-12
  chi^2         relative <= 1e-11
  lnprob        |delta| <= 1e-9 * max(1, |lnprob|), identical finite/-inf pattern   
  sampler       identical accept/reject decisions, positions to 1e-10 relative over 10-12 steps
fp32 / Humlicek-W4 path (BASELINE.json config 5): Re w relative <= 2e-4 for y > 1e-6;
  chi^2 relative <= 1e-3 on fixtures whose chi^2 is not dominated by saturated cores.
"""
import numpy as np
import pytest

from conftest import load_golden
from oracle import vamp_oracle as vo

pytestmark = pytest.mark.gpu


def _case(g, name):
    K = int(name.split("_K")[1].split("_")[0])
    mode = int(name.split("_m")[1].split("_")[0])
    sd = bool(int(name.split("_sd")[1]))
    nbz = g[name + "_nbz"] if mode == vo.MODE_NBZ3 else None
    return g[name + "_x"], g[name + "_flux"], g[name + "_noise"], K, mode, sd, nbz


def test_device_wofz_matches_scipy_and_mpmath(hip_ctx):
    g = load_golden("wofz_grid.npz")
    x, y, w = g["x"], g["y"], g["re_w"]
    out = hip_ctx.wofz_re(x, y)
    ok = w > 1e-300
    assert not np.isnan(out).any()
    rel = np.abs(out[ok] - w[ok]) / w[ok]
    assert rel.max() < 1e-13, (rel.max(), x[ok][rel.argmax()], y[ok][rel.argmax()])
    relm = np.abs(out[g["mp_idx"]] - g["mp_re_w"]) / g["mp_re_w"]
    assert relm.max() < 2e-14
    # symmetric in x
    assert np.array_equal(hip_ctx.wofz_re(-x[:500], y[:500]), out[:500])


def test_lnprob_matches_golden(hip_ctx):
    g = load_golden("lnprob_cases.npz")
    for name in g["cases"]:
        name = str(name)
        x, f, n, K, mode, sd, nbz = _case(g, name)
        hip_ctx.set_regions(x, f, n, K, mode=mode, sample_sd=sd, nbz=None if nbz is None else nbz[None, :])
        th = g[name + "_theta"]
        got, chi = hip_ctx.lnprob(th, return_chi2=True)
        want, wchi = g[name + "_lnprob"], g[name + "_chi2"]
        fin = np.isfinite(want)
        assert np.array_equal(fin, np.isfinite(got)), name
        assert np.all(got[~fin] == -np.inf), name
        err = np.abs(got[fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin]))
        assert err.max() <= 1e-9, (name, err.max())
        assert np.max(np.abs(chi[fin] - wchi[fin]) / wchi[fin]) <= 1e-11, name
        tau, flux = hip_ctx.model(th[0])
        t0 = g[name + "_tau0"]
        big = t0 > 1e-290
        assert np.max(np.abs(tau[big] - t0[big]) / t0[big]) <= 1e-12, name
        assert np.allclose(flux, g[name + "_flux0"], rtol=1e-12, atol=1e-300), name


def test_line_records_and_prior_match_oracle(hip_ctx):
    """Device-side parameter maps ((N,b,z) -> native, physics.py:6-27,116-134) and log-prior
    (vpfits.py:239-252,283-297) against the oracle, per walker."""
    g = load_golden("lnprob_cases.npz")
    for name in ("H1215_r0_K2_m2_sd0", "H1215_r0_K4_m1_sd0", "H1215_r1_K2_m0_sd0", "H1215_r0_K2_m1_sd1"):
        x, f, n, K, mode, sd, nbz = _case(g, name)
        hip_ctx.set_regions(x, f, n, K, mode=mode, sample_sd=sd, nbz=None if nbz is None else nbz[None, :])
        r = vo.Region(x=x, flux=f, noise=n, n_comp=K, mode=mode, sample_sd=sd)
        if nbz is not None:
            r.l_fixed, r.line, r.x_origin, r.x_scale = [float(v) for v in nbz]
        th = g[name + "_theta"]
        for w in range(th.shape[0]):
            rec, lp = hip_ctx.line_records(th[w])
            want_lp = g[name + "_lnprior"][w]
            if np.isfinite(want_lp):
                assert abs(lp - want_lp) <= 1e-12 * max(1.0, abs(want_lp)), (name, w)
            else:
                assert lp == -np.inf or np.isnan(lp), (name, w)
            if not np.all(np.isfinite(th[w])):
                continue
            with np.errstate(all="ignore"):
                comps = vo.native_components(r, th[w])
            for k, comp in enumerate(comps):
                if mode == vo.MODE_GAUSS3:
                    a, c, s = comp
                    want = (c, 1.0 / s if s != 0 else np.inf, 0.0, a)
                else:
                    a, c, L, G = comp
                    want = (c, 2 * vo.SQRT_LN2 / G, L * vo.SQRT_LN2 / G, a * L * np.sqrt(np.pi) * vo.SQRT_LN2 / G)
                assert np.allclose(rec[k, :4], want, rtol=1e-13, atol=0), (name, w, k, rec[k], want)


def test_lnprob_include_norm_and_bounds(hip_ctx):
    g = load_golden("lnprob_cases.npz")
    name = "H1215_r1_K4_m1_sd0"
    x, f, n, K, mode, sd, _ = _case(g, name)
    th = g[name + "_theta"]
    bounds = np.array([[x[2], x[-3], 5.0, 9.0]])
    hip_ctx.set_regions(x, f, n, K, mode=mode, include_norm=True, bounds=bounds)
    r = vo.Region(x=x, flux=f, noise=n, n_comp=K, mode=mode, include_norm=True, c_lo=x[2], c_hi=x[-3], sigma_max=5.0, fwhm_max=9.0)
    want = vo.log_prob_batch(r, th)
    got = hip_ctx.lnprob(th)
    fin = np.isfinite(want)
    assert np.array_equal(fin, np.isfinite(got))
    assert fin.sum() >= 1
    assert np.max(np.abs(got[fin] - want[fin]) / np.maximum(1, np.abs(want[fin]))) <= 1e-9


def test_multi_region_batch_matches_single(hip_ctx):
    """Ragged CSR batch of regions == the same regions uploaded one at a time."""
    g = load_golden("lnprob_cases.npz")
    names = [f"H1215_r{i}_K4_m1_sd0" for i in range(3)] + [f"CII1036_r{i}_K4_m1_sd0" for i in range(4)]
    xs, fs, ns = [g[n + "_x"] for n in names], [g[n + "_flux"] for n in names], [g[n + "_noise"] for n in names]
    hip_ctx.set_regions(xs, fs, ns, 4, mode=vo.MODE_VOIGT4)
    for r, name in enumerate(names):
        got = hip_ctx.lnprob(g[name + "_theta"], region=r)
        want = g[name + "_lnprob"]
        fin = np.isfinite(want)
        assert np.array_equal(fin, np.isfinite(got))
        assert np.max(np.abs(got[fin] - want[fin]) / np.maximum(1, np.abs(want[fin]))) <= 1e-9


def test_model_all_equals_model_region_by_region(hip_ctx):
    """vamp_model_all (component optical depths and model flux of EVERY region from one launch, ragged
    outputs) == vamp_model per region, and the oracle's component_taus / model_flux."""
    g = load_golden("lnprob_cases.npz")
    names = [f"H1215_r{i}_K4_m1_sd0" for i in range(3)] + ["H1215_r0_K1_m1_sd0"]
    xs, fs, ns = [g[n + "_x"] for n in names], [g[n + "_flux"] for n in names], [g[n + "_noise"] for n in names]
    ks = [4, 4, 4, 1]
    hip_ctx.set_regions(xs, fs, ns, ks, mode=vo.MODE_VOIGT4)
    thetas = []
    for n_, k in zip(names, ks):
        th, lnp = g[n_ + "_theta"], g[n_ + "_lnprob"]
        thetas.append(th[np.nanargmax(np.where(np.isfinite(lnp), lnp, -np.inf))])
    taus, fluxes = hip_ctx.model_all(thetas)
    for r in range(4):
        t1, f1 = hip_ctx.model(thetas[r], region=r)
        assert np.array_equal(taus[r], t1) and np.array_equal(fluxes[r], f1)
        reg = vo.Region(x=xs[r], flux=fs[r], noise=ns[r], n_comp=ks[r], mode=vo.MODE_VOIGT4)
        want = vo.component_taus(reg, thetas[r])
        big = want > 1e-290
        assert np.max(np.abs(taus[r][big] - want[big]) / want[big]) <= 1e-12
        assert np.allclose(fluxes[r], vo.model_flux(reg, thetas[r]), rtol=1e-12, atol=1e-300)


def test_stretch_injected_draws_parity(hip_ctx):
    g = load_golden("stretch_traj.npz")
    hip_ctx.set_regions(g["x"], g["flux"], g["noise"], 1, mode=vo.MODE_VOIGT4)
    hip_ctx.sampler_init(g["X0"], seed=1, split_block=16)
    X, lnp, _, _ = hip_ctx.get_state()
    assert np.allclose(lnp, g["lnp0"], rtol=1e-10)
    for i in range(g["active"].shape[0]):
        hip_ctx.half_step_ext(g["active"][i], g["partner"][i], g["zz"][i], g["logu"][i])
        X, lnp, nacc, _ = hip_ctx.get_state()
        # identical accept/reject decisions <=> identical set of changed rows
        assert np.allclose(X, g["X_after"][i], rtol=1e-10, atol=1e-12), i
        assert np.allclose(lnp, g["lnp_after"][i], rtol=1e-9, atol=1e-9), i


@pytest.mark.parametrize("block", [8, 16])
def test_stretch_philox_trajectory_parity(hip_ctx, block):
    g = load_golden("stretch_traj.npz")
    hip_ctx.set_regions(g["x"], g["flux"], g["noise"], 1, mode=vo.MODE_VOIGT4)
    hip_ctx.sampler_init(g["X0"], seed=0x1234ABCD5678EF01, a=2.0, split_block=block)
    res = hip_ctx.run(12)
    assert np.allclose(res["chain"], g[f"philox_chain_b{block}"], rtol=1e-10, atol=1e-12)
    assert np.allclose(res["lnprob"], g[f"philox_lnp_b{block}"], rtol=1e-9, atol=1e-9)
    assert np.array_equal(res["n_accept"], g[f"philox_nacc_b{block}"])


def test_sampler_resume_and_thin(hip_ctx):
    """run(12) == run(5) + run(7); thinning keeps every 3rd sample; get/set_state round trip."""
    g = load_golden("stretch_traj.npz")
    hip_ctx.set_regions(g["x"], g["flux"], g["noise"], 1, mode=vo.MODE_VOIGT4)
    hip_ctx.sampler_init(g["X0"], seed=77, split_block=16)
    full = hip_ctx.run(12)
    hip_ctx.sampler_init(g["X0"], seed=77, split_block=16)
    a = hip_ctx.run(5)
    X, lnp, nacc, step = hip_ctx.get_state()
    assert step == 5
    hip_ctx.set_state(X, lnp, step)
    b = hip_ctx.run(7)
    assert np.array_equal(np.concatenate([a["chain"], b["chain"]]), full["chain"])
    hip_ctx.sampler_init(g["X0"], seed=77, split_block=16)
    t = hip_ctx.run(12, thin=3)
    assert np.array_equal(t["chain"], full["chain"][2::3])


def test_sampler_multi_region_matches_oracle(hip_ctx):
    """Two regions sampled in one launch per half-step follow the oracle's per-region chains."""
    g = load_golden("lnprob_cases.npz")
    names = ["H1215_r0_K1_m1_sd0", "H1215_r2_K1_m1_sd0"]
    rng = np.random.default_rng(5)
    xs, fs, ns, X0s, regs = [], [], [], [], []
    for name in names:
        x, f, n = g[name + "_x"], g[name + "_flux"], g[name + "_noise"]
        r = vo.Region(x=x, flux=f, noise=n, n_comp=1, mode=vo.MODE_VOIGT4)
        X0 = np.stack([rng.uniform(0.3, 1.5, 16), rng.uniform(-4, 4, 16), rng.uniform(0.5, 3, 16), rng.uniform(2, 8, 16)], 1)
        xs.append(x); fs.append(f); ns.append(n); X0s.append(X0); regs.append(r)
    hip_ctx.set_regions(xs, fs, ns, 1, mode=vo.MODE_VOIGT4)
    hip_ctx.sampler_init(X0s, seed=2024, split_block=8)
    res = hip_ctx.run(6)
    for ri, r in enumerate(regs):
        fn = lambda q, r=r: vo.log_prob_batch(r, q)
        chain, lchain, nacc = vo.run_sampler(fn, X0s[ri], fn(X0s[ri]), 6, seed=2024, block=8, region=ri, walker_off=ri * 16)
        assert np.allclose(res["chain"][ri], chain, rtol=1e-10, atol=1e-12)
        assert np.array_equal(res["n_accept"][ri], nacc)


def test_mixed_launch_classes_match_oracle():
    """A spectrum-like context under automatic packing is cut into two launch classes (short regions;
    blends of >= 3 lines over >= 96 px with per-walker Taylor tables), each with its own launch per
    half-step over its own region list.  lnprob, lnprob_all and the sampler chains of every region
    against the oracle, region by region; then the same regions with four walkers per wavefront
    forced (draws from the k_draws launch) give the same chains."""
    import vamp_amd
    from tools.bench_c3 import build_regions, start_walkers
    xs, fs, ns, ks = build_regions()
    big = [r for r in range(len(xs)) if ks[r] >= 3 and len(xs[r]) >= 96][:3]
    small = [r for r in range(len(xs)) if ks[r] == 1][:3]
    pick = [small[0], big[0], small[1], big[1], big[2], small[2]]            # interleaved: the lists are not contiguous
    xs, fs, ns, ks = [xs[r] for r in pick], [fs[r] for r in pick], [ns[r] for r in pick], [ks[r] for r in pick]
    rng = np.random.default_rng(7)
    W = 32
    th = [start_walkers(rng, x, k, W) for x, k in zip(xs, ks)]
    regs = [vo.Region(x=x, flux=f, noise=n, n_comp=k, mode=vo.MODE_VOIGT4) for x, f, n, k in zip(xs, fs, ns, ks)]
    chains = {}
    for packing in (0, 16):
        with vamp_amd.HipContext(device=0) as ctx:
            ctx.set_packing(packing)
            ctx.set_regions(xs, fs, ns, ks, mode=vamp_amd.MODE_VOIGT4)
            la, ca = ctx.lnprob_all(th, return_chi2=True)
            for r, reg in enumerate(regs):
                want, wchi = vo.log_prob_batch(reg, th[r], return_chi2=True)
                assert np.isfinite(want).all()
                assert np.max(np.abs(la[r] - want) / np.maximum(1, np.abs(want))) <= 1e-9, (packing, r)
                assert np.max(np.abs(ca[r] - wchi) / wchi) <= 1e-11, (packing, r)
                l1 = ctx.lnprob(th[r], region=r)
                assert np.array_equal(l1, la[r]), (packing, r)
            ctx.sampler_init(th, seed=606, split_block=8)
            res = ctx.run(4)
            chains[packing] = res["chain"]
            for r, reg in enumerate(regs):
                fn = lambda q, reg=reg: vo.log_prob_batch(reg, q)
                chain, lchain, nacc = vo.run_sampler(fn, th[r], fn(th[r]), 4, seed=606, block=8, region=r, walker_off=r * W)
                assert np.allclose(res["chain"][r], chain, rtol=1e-10, atol=1e-12), (packing, r)
                assert np.array_equal(res["n_accept"][r], nacc), (packing, r)
    for r in range(len(regs)):
        assert np.allclose(chains[0][r], chains[16][r], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_packed_launch_classes_at_production_size(dtype):
    """Automatic packing only packs launches of >= 16 384 walkers, so the production shapes of a
    spectrum-like context -- four walkers per wavefront with 2-line and 8-line LDS layouts, draws from
    k_draws, and the 2-wavefront groups with Taylor tables -- run here at W = 32 768 on four q1422
    regions (1, 2, 3 and >= 4 lines).  fp64: lnprob of every walker and two full stretch steps of
    every walker against the oracle.  fp32: chi^2 within the stated 1e-3."""
    import vamp_amd
    from tools.bench_c3 import build_regions, start_walkers
    xs, fs, ns, ks = build_regions()
    pick = [next(r for r in range(len(xs)) if ks[r] == 1 and 30 <= len(xs[r]) <= 60)]
    pick.append(next(r for r in range(len(xs)) if ks[r] >= 4 and 96 <= len(xs[r]) <= 200))
    xs, fs, ns = [xs[r] for r in pick] * 2, [fs[r] for r in pick] * 2, [ns[r] for r in pick] * 2
    ks = [1, ks[pick[1]], 2, 3]        # the same data with 2 and 3 lines: the 2-line layout and the short 8-line one
    xs[3], fs[3], ns[3] = xs[3][:80], fs[3][:80], ns[3][:80]                # < 96 px: stays with the packed class
    W = 32768
    rng = np.random.default_rng(3)
    th = [start_walkers(rng, x, k, W) for x, k in zip(xs, ks)]
    regs = [vo.Region(x=x, flux=f, noise=n, n_comp=k, mode=vo.MODE_VOIGT4) for x, f, n, k in zip(xs, fs, ns, ks)]
    with vamp_amd.HipContext(device=0, dtype=vamp_amd.F64 if dtype == "f64" else vamp_amd.F32) as ctx:
        ctx.set_regions(xs, fs, ns, ks, mode=vamp_amd.MODE_VOIGT4)
        la, ca = ctx.lnprob_all(th, return_chi2=True)
        want = [vo.log_prob_batch_fast(reg, t) for reg, t in zip(regs, th)]
        for r in range(4):
            assert np.isfinite(want[r]).all()
            if dtype == "f64":
                assert np.max(np.abs(la[r] - want[r]) / np.maximum(1, np.abs(want[r]))) <= 1e-9, r
            else:
                assert np.max(np.abs(la[r] - want[r]) / np.abs(want[r])) <= 1e-3, r
        if dtype == "f32":
            return
        ctx.sampler_init(th, seed=77, split_block=1024)
        res = ctx.run(2)
        for r, reg in enumerate(regs):
            fn = lambda q, reg=reg: vo.log_prob_batch_fast(reg, q)
            chain, lchain, nacc = vo.run_sampler_batch(fn, th[r], want[r], 2, seed=77, block=1024, region=r, walker_off=r * W)
            # an accept decision whose margin is at rounding level (|log u - diff| ~ 1e-13 of 65 536
            # decisions per region) may legitimately differ: allow a handful of walkers
            ok = np.all(np.abs(res["chain"][r] - chain) <= 1e-10 * np.abs(chain) + 1e-12, axis=(0, 2))
            assert (~ok).sum() <= 2, (r, int((~ok).sum()))
            assert np.abs(res["n_accept"][r] - nacc).sum() <= 2, r
            assert np.allclose(res["lnprob"][r][:, ok], lchain[:, ok], rtol=1e-9, atol=1e-9)


def test_config2_shape_against_oracle(hip_ctx):
    """BASELINE.json config 2 on its own workload: the H I Ly-alpha region [672, 716) of
    simba_H1215.h5, 4 Voigt components, 4096 walkers, fp64 (tools/bench_c2.py times this shape).
    lnprob of all 4096 starting walkers and five full stretch steps of the whole ensemble against
    the oracle (numpy-vectorised replay of the counter-based draws)."""
    from vamp_amd.physics import Wave2freq
    g = load_golden("simba_spectra.npz")
    s, e = g["H1215_region_pixels"][0]
    nu = np.flip(Wave2freq(g["H1215_wavelength"][s:e]), 0)
    flux, noise = np.flip(g["H1215_flux"][s:e], 0), np.flip(g["H1215_noise"][s:e], 0)
    x = (nu - 0.5 * (nu[0] + nu[-1])) / ((nu[-1] - nu[0]) / (nu.size - 1))
    assert x.size == 44
    rng = np.random.default_rng(2)
    K, W = 4, 4096
    th = np.empty((W, 4 * K))
    for k in range(K):
        th[:, 4 * k] = rng.gamma(2.0, 1.0, W)
        th[:, 4 * k + 1] = rng.uniform(x[0], x[-1], W)
        th[:, 4 * k + 2] = rng.uniform(0.5, 8, W)
        th[:, 4 * k + 3] = rng.uniform(2, 15, W)
    reg = vo.Region(x=x, flux=flux, noise=noise, n_comp=K, mode=vo.MODE_VOIGT4)
    fn = lambda q: vo.log_prob_batch_fast(reg, q)
    want = fn(th)
    hip_ctx.set_regions(x, flux, noise, K, mode=vo.MODE_VOIGT4)
    got = hip_ctx.lnprob(th)
    assert np.isfinite(want).all() and np.max(np.abs(got - want) / np.maximum(1, np.abs(want))) <= 1e-9
    hip_ctx.sampler_init(th, seed=5, split_block=1024)
    res = hip_ctx.run(5)
    chain, lchain, nacc = vo.run_sampler_batch(fn, th, want, 5, seed=5, block=1024)
    ok = np.all(np.abs(res["chain"] - chain) <= 1e-10 * np.abs(chain) + 1e-12, axis=(0, 2))
    assert (~ok).sum() <= 1 and np.abs(res["n_accept"] - nacc).sum() <= 1      # 20 480 decisions: a rounding-level margin may flip one
    assert 0.02 < nacc.mean() / 5 < 0.9


def test_sampler_sd_mode_and_acceptance(hip_ctx):
    """Reference-form likelihood (free sd, vpfits.py:39): stored lnprob equals a fresh evaluation
    of the final positions, and the acceptance fraction is sane."""
    g = load_golden("lnprob_cases.npz")
    name = "H1215_r0_K2_m1_sd1"
    x, f, n, K, mode, sd, _ = _case(g, name)
    hip_ctx.set_regions(x, f, n, K, mode=mode, sample_sd=True)
    rng = np.random.default_rng(11)
    W = 256
    X0 = np.empty((W, 9))
    for k in range(2):
        X0[:, 4 * k:4 * k + 4] = np.stack([rng.uniform(0.5, 2, W), rng.uniform(-6, 6, W), rng.uniform(0.5, 2, W), rng.uniform(3, 9, W)], 1)
    X0[:, 8] = rng.uniform(0.05, 0.5, W)
    hip_ctx.sampler_init(X0, seed=5, split_block=64)
    res = hip_ctx.run(200, store_chain=False)
    X, lnp, nacc, step = hip_ctx.get_state()
    assert step == 200
    fresh = hip_ctx.lnprob(X)
    assert np.allclose(fresh, lnp, rtol=1e-12, atol=1e-9)
    r = vo.Region(x=x, flux=f, noise=n, n_comp=2, mode=mode, sample_sd=True)
    assert np.allclose(vo.log_prob_batch(r, X[:32]), lnp[:32], rtol=1e-9, atol=1e-9)
    acc = nacc.mean() / 200
    assert 0.05 < acc < 0.8, acc
    assert lnp.mean() > hip_ctx.lnprob(X0).mean()          # the ensemble climbed


def test_smallest_shapes(hip_ctx):
    """Edges of the shape space: a 2-pixel region (the minimum vamp_set_regions accepts), the smallest
    ensemble (W = 2, one mover per colour), a sampler asked for zero steps, thinning that keeps
    nothing, and 16 components on a 3-pixel region -- all against the oracle."""
    rng = np.random.default_rng(99)
    x = np.array([-0.5, 0.5])
    f, n = np.array([0.7, 0.8]), np.array([0.05, 0.05])
    hip_ctx.set_regions(x, f, n, 1, mode=vo.MODE_VOIGT4)
    reg = vo.Region(x=x, flux=f, noise=n, n_comp=1, mode=vo.MODE_VOIGT4)
    th = np.stack([rng.uniform(0.1, 1, 6), rng.uniform(-0.5, 0.5, 6), rng.uniform(0.05, 1.1, 6), rng.uniform(0.05, 1.1, 6)], 1)
    want = vo.log_prob_batch(reg, th)
    got = hip_ctx.lnprob(th)
    assert np.isfinite(want).all() and np.max(np.abs(got - want) / np.maximum(1, np.abs(want))) <= 1e-9
    hip_ctx.sampler_init(th[:2], seed=4, split_block=2)
    res = hip_ctx.run(5)
    chain, _, nacc = vo.run_sampler(lambda q: vo.log_prob_batch(reg, q), th[:2], want[:2], 5, seed=4, block=2)
    assert np.allclose(res["chain"], chain, rtol=1e-10, atol=1e-12) and np.array_equal(res["n_accept"], nacc)
    empty = hip_ctx.run(0)
    assert empty["chain"].shape == (0, 2, 4) and np.array_equal(empty["n_accept"], nacc)
    none_kept = hip_ctx.run(2, thin=5)
    assert none_kept["chain"].shape == (0, 2, 4)
    if hip_ctx.packing_request in (16, 65):
        return                                    # those shapes hold at most 8 components
    x3 = np.array([-1.0, 0.0, 1.0])
    f3, n3 = np.array([0.9, 0.2, 0.85]), np.full(3, 0.02)
    hip_ctx.set_regions(x3, f3, n3, 16, mode=vo.MODE_VOIGT4)
    reg3 = vo.Region(x=x3, flux=f3, noise=n3, n_comp=16, mode=vo.MODE_VOIGT4)
    th3 = np.tile(np.array([0.2, 0.0, 0.5, 1.0]), (4, 16)) * (1 + 0.05 * rng.standard_normal((4, 64)))
    th3[:, 1::4] = rng.uniform(-1, 1, (4, 16))
    want3 = vo.log_prob_batch(reg3, th3)
    got3 = hip_ctx.lnprob(th3)
    assert np.isfinite(want3).all() and np.max(np.abs(got3 - want3) / np.maximum(1, np.abs(want3))) <= 1e-9


def test_error_behaviour(hip_ctx):
    import vamp_amd
    g = load_golden("stretch_traj.npz")
    hip_ctx.set_regions(g["x"], g["flux"], g["noise"], 1, mode=vo.MODE_VOIGT4)
    with pytest.raises(ValueError):
        hip_ctx.lnprob(np.zeros((4, 3)))
    with pytest.raises(vamp_amd._lib.VampError) as e:
        hip_ctx.sampler_init(g["X0"][:15], seed=1, split_block=5)
    assert e.value.code == -1
    hip_ctx.sampler_init(g["X0"], seed=1, split_block=16)
    with pytest.raises(vamp_amd._lib.VampError):          # partner inside the active set
        hip_ctx.half_step_ext([0, 1], [1, 2], [1.0, 1.0], [0.0, 0.0])
    with pytest.raises(vamp_amd._lib.VampError):          # index out of range
        hip_ctx.half_step_ext([0], [99], [1.0], [0.0])
    with pytest.raises(vamp_amd._lib.VampError):
        vamp_amd.HipContext(device=0, dtype=vamp_amd.F32, wofz_kind=0)
    # the grid of a region must be strictly monotonic and finite
    for bad in (np.r_[g["x"][:5], g["x"][3], g["x"][6:]], np.r_[g["x"][:4], np.nan, g["x"][5:]],
                np.r_[g["x"][:4], g["x"][3], g["x"][5:]]):
        with pytest.raises(vamp_amd._lib.VampError) as e:
            hip_ctx.set_regions(bad, g["flux"], g["noise"], 1, mode=vo.MODE_VOIGT4)
        assert e.value.code == -1
    hip_ctx.set_regions(g["x"], g["flux"], g["noise"], 1, mode=vo.MODE_VOIGT4)
    # non-finite parameters are -inf, not errors
    th = g["X0"].copy()
    th[0, 0] = np.nan
    th[1, 3] = np.inf
    out = hip_ctx.lnprob(th)
    assert out[0] == -np.inf and out[1] == -np.inf and np.isfinite(out[2:]).all()


def test_fp32_humlicek_path():
    """BASELINE.json config 5: fp32 pixel arithmetic + Humlicek W4 against the fp64 oracle."""
    import vamp_amd
    from scipy.special import wofz
    g = load_golden("wofz_grid.npz")
    ctx = vamp_amd.HipContext(device=0, dtype=vamp_amd.F32)
    try:
        x, y = g["x"].astype(np.float32).astype(np.float64), g["y"].astype(np.float32).astype(np.float64)
        w = wofz(x + 1j * y).real
        out = ctx.wofz_re(x, y)
        m = (y > 1e-6) & (w > 1e-30) & (x < 1e15)
        assert np.max(np.abs(out[m] - w[m]) / w[m]) < 2e-4
        c = load_golden("lnprob_cases.npz")
        for name in ("H1215_r0_K4_m1_sd0", "CII1036_r1_K4_m1_sd0", "H1215_r1_K2_m0_sd0"):
            xx, f, n, K, mode, sd, _ = _case(c, name)
            ctx.set_regions(xx, f, n, K, mode=mode)
            th = c[name + "_theta"]
            got, chi = ctx.lnprob(th, return_chi2=True)
            want, wchi = c[name + "_lnprob"], c[name + "_chi2"]
            fin = np.isfinite(want) & (th[:, 3 if mode == 1 else 2] > 1e-3)   # fp32 cannot resolve G ~ 1e-6 px
            assert np.array_equal(np.isfinite(want), np.isfinite(got)), name
            rel = np.abs(chi[fin] - wchi[fin]) / wchi[fin]
            assert np.median(rel) < 1e-4 and rel.max() < 1e-3, (name, rel.max())
        # long regions: full tiles take the far-field interpolant (fp64 nodes, fp32 Clenshaw) for
        # distant lines and W4 for near ones; with and without a ragged tail, both launch shapes
        from bench import make_workload
        for P, packing in ((2048, 0), (2048, 64), (3000, 0), (1000, 0)):
            wl = make_workload(P=P, K=6, W=64, seed=21, nbz=False)
            ctx.set_packing(packing)
            ctx.set_regions(wl["x"], wl["flux"], wl["noise"], 6, mode=vo.MODE_VOIGT4)
            got, chi = ctx.lnprob(wl["theta0"], return_chi2=True)
            r = vo.Region(x=wl["x"], flux=wl["flux"], noise=wl["noise"], n_comp=6, mode=vo.MODE_VOIGT4)
            want, wchi = vo.log_prob_batch(r, wl["theta0"], return_chi2=True)
            assert np.isfinite(want).all() and np.isfinite(got).all()
            rel = np.abs(chi - wchi) / wchi
            assert np.median(rel) < 1e-4 and rel.max() < 1e-3, (P, packing, rel.max())
        ctx.set_packing(0)
    finally:
        ctx.close()


@pytest.mark.parametrize("packing", [0, 16])
def test_config5_q1422_regions_fp32_vs_fp64_oracle(packing):
    """BASELINE.json config 5 on its own workload: all 421 detected regions of the q1422 spectrum
    (config 3's batch, 9..478 px, <= 8 Voigt lines each) evaluated in ONE fp32 / Humlicek-W4 launch
    and compared region by region with the fp64 scipy.wofz oracle.  Stated tolerance
    (SURVEY 8d): |delta chi^2| / chi^2 <= 1e-3 for every (region, walker)."""
    import vamp_amd
    from tools.bench_c3 import build_regions, start_walkers
    xs, fs, ns, ks = build_regions()
    assert len(xs) == 421 and sum(len(x) for x in xs) == 27536 and max(ks) <= 8
    rng = np.random.default_rng(1422)
    W = 8 if packing == 0 else 16
    thetas = [start_walkers(rng, x, k, W) for x, k in zip(xs, ks)]
    ctx = vamp_amd.HipContext(device=0, dtype=vamp_amd.F32)
    try:
        ctx.set_packing(packing)
        ctx.set_regions(xs, fs, ns, ks, mode=vamp_amd.MODE_VOIGT4)
        got, chi = ctx.lnprob_all(thetas, return_chi2=True)
    finally:
        ctx.close()
    worst = 0.0
    for r, (x, f, n, k) in enumerate(zip(xs, fs, ns, ks)):
        reg = vo.Region(x=x, flux=f, noise=n, n_comp=k, mode=vo.MODE_VOIGT4)
        want, wchi = vo.log_prob_batch(reg, thetas[r], return_chi2=True)
        assert np.isfinite(want).all() and np.isfinite(got[r]).all(), r
        rel = np.abs(chi[r] - wchi) / wchi
        worst = max(worst, rel.max())
        assert rel.max() <= 1e-3, (r, len(x), k, rel.max())
        assert np.max(np.abs(got[r] - want) / np.abs(want)) <= 1e-3, r
    print("config 5: worst relative chi^2 error over 421 regions x %d walkers: %.2e" % (W, worst))


@pytest.mark.parametrize("packing", [0, 16])
def test_config3_q1422_all_regions_fp64_vs_oracle(packing):
    """BASELINE.json config 3 on its own workload, the parity path: ALL 421 detected regions of the q1422
    spectrum in ONE fp64 launch (every launch class: blends with Taylor tables, one- and two-line regions,
    the rest) against the scipy.wofz oracle, region by region -- |delta lnprob| <= 1e-9 max(1, |lnprob|) and
    chi^2 to 1e-11 relative for every (region, walker), identical -inf pattern.  W = 8 walkers per region run
    one walker per wavefront (and per-walker tables for the blends); W = 16 under packing 16 runs four
    walkers per wavefront."""
    import vamp_amd
    from tools.bench_c3 import build_regions, start_walkers
    xs, fs, ns, ks = build_regions()
    assert len(xs) == 421 and sum(len(x) for x in xs) == 27536 and max(ks) <= 8
    rng = np.random.default_rng(1422)
    W = 8 if packing == 0 else 16
    thetas = [start_walkers(rng, x, k, W) for x, k in zip(xs, ks)]
    ctx = vamp_amd.HipContext(device=0, dtype=vamp_amd.F64)
    try:
        ctx.set_packing(packing)
        ctx.set_regions(xs, fs, ns, ks, mode=vamp_amd.MODE_VOIGT4)
        got, chi = ctx.lnprob_all(thetas, return_chi2=True)
    finally:
        ctx.close()
    worst = worst_chi = 0.0
    for r, (x, f, n, k) in enumerate(zip(xs, fs, ns, ks)):
        reg = vo.Region(x=x, flux=f, noise=n, n_comp=k, mode=vo.MODE_VOIGT4)
        want, wchi = vo.log_prob_batch(reg, thetas[r], return_chi2=True)
        fin = np.isfinite(want)
        assert np.array_equal(fin, np.isfinite(got[r])), r
        err = np.abs(got[r][fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin]))
        rel = np.abs(chi[r][fin] - wchi[fin]) / wchi[fin]
        worst, worst_chi = max(worst, err.max(initial=0.0)), max(worst_chi, rel.max(initial=0.0))
        assert err.max(initial=0.0) <= 1e-9, (r, len(x), k, err.max())
        assert rel.max(initial=0.0) <= 1e-11, (r, len(x), k, rel.max())
    print("config 3: worst lnprob error %.2e, worst relative chi^2 error %.2e over 421 regions x %d walkers" % (worst, worst_chi, W))


def test_full_size_properties(hip_ctx):
    """Headline shape (P = 16384, K = 16; fewer walkers): properties that need no CPU reference.
    (1) batch-position independence, (2) component-permutation invariance, (3) tau is the sum of
    single-component taus, (4) a down-scaled twin agrees with the oracle."""
    from bench import make_workload
    if hip_ctx.packing_request in (16, 65):
        pytest.skip("16 components need the one-walker-per-wavefront kernels")
    wl = make_workload(P=16384, K=16, W=64, seed=20240517, nbz=False)
    hip_ctx.set_regions(wl["x"], wl["flux"], wl["noise"], 16, mode=vo.MODE_VOIGT4)
    th = wl["theta0"]
    a = hip_ctx.lnprob(th)
    b = hip_ctx.lnprob(th[::-1].copy())[::-1]
    assert np.array_equal(a, b)
    perm = np.random.default_rng(0).permutation(16)
    thp = th.reshape(64, 16, 4)[:, perm, :].reshape(64, 64)
    c = hip_ctx.lnprob(thp)
    assert np.allclose(a, c, rtol=1e-11)
    tau, flux = hip_ctx.model(th[0])
    assert np.allclose(flux, np.exp(-tau.sum(0)), rtol=1e-13)
    # single-component contexts reproduce each row of tau
    hip_ctx.set_regions(wl["x"], wl["flux"], wl["noise"], 1, mode=vo.MODE_VOIGT4,
                        bounds=np.array([[wl["x"][0], wl["x"][-1], 1e9, 1e9]]))
    for k in (0, 7, 15):
        t1, _ = hip_ctx.model(th[0, 4 * k:4 * k + 4])
        assert np.array_equal(t1[0], tau[k])
    small = make_workload(P=256, K=16, W=64, seed=3, nbz=False)
    hip_ctx.set_regions(small["x"], small["flux"], small["noise"], 16, mode=vo.MODE_VOIGT4)
    r = vo.Region(x=small["x"], flux=small["flux"], noise=small["noise"], n_comp=16, mode=vo.MODE_VOIGT4)
    want = vo.log_prob_batch_fast(r, small["theta0"])
    got = hip_ctx.lnprob(small["theta0"])
    assert np.max(np.abs(got - want) / np.maximum(1, np.abs(want))) <= 1e-9


def test_bench_shape_against_oracle(hip_ctx):
    """The exact shape bench.py times -- P = 16384, K = 16, (N,b,z) D = 48, the workgroup-per-walker
    kernels with the far-field interpolant and the per-line Taylor tables -- against the oracle
    directly: lnprob of the bench's own starting walkers (|delta| <= 1e-9 max(1,|lnprob|)) and two
    full stretch steps with the counter-based draws (identical accept decisions, positions to
    1e-10).  16 walkers x 262 144 scipy.wofz evaluations each."""
    from bench import make_workload
    if hip_ctx.packing_request in (16, 65):
        pytest.skip("16 components need the one-walker-per-wavefront kernels")
    wl = make_workload(P=16384, K=16, W=16, nbz=True)
    assert wl["D"] == 48 and wl["mode"] == vo.MODE_NBZ3
    hip_ctx.set_regions(wl["x"], wl["flux"], wl["noise"], 16, mode=vo.MODE_NBZ3, nbz=wl["nbz"])
    l_fixed, line, x_origin, x_scale = [float(v) for v in wl["nbz"][0]]
    r = vo.Region(x=wl["x"], flux=wl["flux"], noise=wl["noise"], n_comp=16, mode=vo.MODE_NBZ3, l_fixed=l_fixed, line=line,
                  x_origin=x_origin, x_scale=x_scale)
    fn = lambda q: vo.log_prob_batch_fast(r, q)
    want = fn(wl["theta0"])
    got, chi = hip_ctx.lnprob(wl["theta0"], return_chi2=True)
    assert np.isfinite(want).all() and np.isfinite(got).all()
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) <= 1e-9
    assert np.max(np.abs(-0.5 * chi - (want - [vo.log_prior(r, t) for t in wl["theta0"]])) / np.abs(want)) <= 1e-9
    # tau and flux of one walker, pixel by pixel (k_model: the per-point evaluator)
    tau, flux = hip_ctx.model(wl["theta0"][0])
    t0 = vo.component_taus(r, wl["theta0"][0])
    big = t0 > 1e-290
    assert np.max(np.abs(tau[big] - t0[big]) / t0[big]) <= 1e-12
    assert np.allclose(flux, vo.model_flux(r, wl["theta0"][0]), rtol=1e-12, atol=1e-300)
    # two sampler steps through k_half_step<.., NBZ3, Pack<64,16,false,4,true>> (packing auto / 256)
    hip_ctx.sampler_init(wl["theta0"], seed=20240517, a=2.0, split_block=16)
    res = hip_ctx.run(2)
    chain, lchain, nacc = vo.run_sampler(fn, wl["theta0"], want, 2, seed=20240517, block=16)
    assert np.array_equal(res["n_accept"], nacc)
    assert np.allclose(res["chain"], chain, rtol=1e-10, atol=0)
    assert np.allclose(res["lnprob"], lchain, rtol=1e-9, atol=1e-9)
    assert nacc.sum() > 0


def test_rare_branches_in_long_regions(hip_ctx):
    """Data-dependent branches that the seeded fixtures only reach in one-tile regions, forced here
    in a 4096-pixel region (64 tiles per line): a narrow line whose |z| spans decades inside one tile
    (cap + closed far form), a line with y < 1e-9 (missing e^{-x^2} restored in every fraction
    branch), a broad damped line (y > 4.5: no pole term), and a line centred off the grid."""
    if hip_ctx.packing_request == 16:
        pytest.skip("long region: one walker per wavefront")
    rng = np.random.default_rng(77)
    P = 4096
    x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
    noise = np.full(P, 0.02)
    flux = 1.0 + rng.normal(0, 0.02, P)
    cases = np.array([
        # A,    c,       L,       G
        [1.2, -1500.3, 0.5, 0.04,      0.7, 200.7, 1e-12, 30.0,     2.0, 900.0, 400.0, 50.0,    0.9, 1800.25, 3.0, 9.0],
        [0.4, 10.0, 1e-3, 1e-3,        1.5, -700.0, 2e-11, 5.0,     0.3, 0.0, 900.0, 120.0,     2.5, -2000.0, 40.0, 2.0],
        [3.0, 2047.5, 0.02, 0.3,       0.2, -2047.5, 1e-10, 80.0,   1.0, 333.3, 2000.0, 300.0,  0.6, -10.0, 0.5, 700.0],
    ])
    hip_ctx.set_regions(x, flux, noise, 4, mode=vo.MODE_VOIGT4)
    r = vo.Region(x=x, flux=flux, noise=noise, n_comp=4, mode=vo.MODE_VOIGT4)
    want, wchi = vo.log_prob_batch(r, cases, return_chi2=True)
    got, chi = hip_ctx.lnprob(cases, return_chi2=True)
    assert np.isfinite(want).all()
    assert np.max(np.abs(got - want) / np.maximum(1, np.abs(want))) <= 1e-9
    assert np.max(np.abs(chi - wchi) / wchi) <= 1e-11
    for th in cases:
        tau, fl = hip_ctx.model(th)
        ref = vo.component_taus(r, th)
        big = ref > 1e-280
        assert np.max(np.abs(tau[big] - ref[big]) / ref[big]) <= 1e-12
    # the same lines through the sampler kernel's sweep: the stored lnprob of an unmoved ensemble
    X0 = np.repeat(cases, 8, axis=0)[:16] * (1 + 1e-9 * rng.standard_normal((16, 16)))
    hip_ctx.sampler_init(X0, seed=3, split_block=16)
    hip_ctx.run(2, store_chain=False)
    X, lnp, nacc, _ = hip_ctx.get_state()
    assert np.allclose(vo.log_prob_batch(r, X), lnp, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("P", [4096, 3000, 200])
def test_split_workgroup_matches_wave_per_walker(P):
    """One walker per wavefront (packing 64) against one walker per 4-wavefront workgroup (packing
    256: each wavefront sweeps every 4th tile).  Both sum chi^2 in the same order.  The workgroup shape
    evaluates the line cores through per-line Taylor tables: in fp64 (absolute error 2e-16 in H) lnprob
    agrees to rounding; in fp32 the tables replace Humlicek's regions III / IV (1e-4 relative, W4's own
    error), so the two shapes agree to that -- |delta chi^2| / chi^2 <= 1e-3 is the fp32 tolerance of
    SURVEY 8d, and both are checked against the fp64 workgroup run as well."""
    import vamp_amd
    from bench import make_workload
    wl = make_workload(P=P, K=5, W=64, seed=11, nbz=False)
    for dtype in (vamp_amd.F64, vamp_amd.F32):
        got = []
        for packing in (64, 256):
            ctx = vamp_amd.HipContext(device=0, dtype=dtype)
            ctx.set_packing(packing)
            ctx.set_regions(wl["x"], wl["flux"], wl["noise"], 5, mode=vo.MODE_VOIGT4)
            lnp = ctx.lnprob(wl["theta0"])
            ctx.sampler_init(wl["theta0"], seed=5, split_block=16)
            ctx.run(6)
            X, lp, nacc, _ = ctx.get_state()
            got.append((lnp, X, lp, nacc))
            ctx.close()
        assert np.isfinite(got[0][0]).all() and got[0][3].sum() > 0 and got[1][3].sum() > 0
        a, b = got[0][0], got[1][0]
        if dtype == vamp_amd.F32:
            assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(a))) <= 1e-3
            for v in (a, b):
                assert np.max(np.abs(v - ref64) / np.maximum(1.0, np.abs(ref64))) <= 1e-3
            assert np.max(np.abs(b - ref64) / np.maximum(1.0, np.abs(ref64))) <= 3e-5       # tables + W4 wings: well inside
        else:
            assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(a))) <= 1e-12
            ref64 = b


def test_fp32_posterior_means_agree_with_fp64():
    """SURVEY 8d tolerance for config 5: posterior means of the fp32 / Humlicek-W4 run within 0.1
    posterior standard deviations of the fp64 run.  A well-constrained synthetic region (256 px,
    2 Voigt lines, sigma = 0.01: the posterior is narrow, so fp32 error would show), 512 walkers,
    3000 steps thinned by 5, first third discarded; same seed and start (the two chains decorrelate
    within a few steps, so the Monte Carlo error of each mean, ~0.02 sigma, is part of the budget)."""
    import vamp_amd
    from bench import make_workload
    wl = make_workload(P=256, K=2, W=512, seed=4, nbz=False)
    means, sds = [], []
    for dtype in (vamp_amd.F64, vamp_amd.F32):
        with vamp_amd.HipContext(device=0, dtype=dtype) as ctx:
            ctx.set_regions(wl["x"], wl["flux"], wl["noise"], 2, mode=vo.MODE_VOIGT4)
            ctx.sampler_init(wl["theta0"], seed=77)
            res = ctx.run(3000, thin=5)
        ch = res["chain"][200:].reshape(-1, 8)
        assert np.isfinite(ch).all()
        means.append(ch.mean(0))
        sds.append(ch.std(0))
    assert np.all(np.abs(means[0] - means[1]) <= 0.1 * sds[0]), (means, sds)
    assert np.allclose(sds[0], sds[1], rtol=0.15)


def test_map_all_follows_scipy_fmin(hip_ctx):
    """vamp_map_all (every region's Nelder-Mead search advanced together, one launch per
    iteration) against scipy.optimize.fmin run region by region on the same device log-posterior:
    same rules, same start simplex, same stopping test -> the same optimum and iteration count."""
    from scipy.optimize import fmin
    c = load_golden("lnprob_cases.npz")
    names = ["H1215_r0_K1_m1_sd0", "H1215_r1_K4_m1_sd0", "CII1036_r1_K4_m1_sd0", "H1215_r2_K1_m1_sd0"]
    if hip_ctx.packing_request == 256:
        pytest.skip("short regions: same kernels as packing 64")
    xs, fs, ns, Ks, starts = [], [], [], [], []
    for name in names:
        x, f, n, K, mode, sd, _ = _case(c, name)
        th, lnp = c[name + "_theta"], c[name + "_lnprob"]
        xs.append(x); fs.append(f); ns.append(n); Ks.append(K)
        starts.append(th[np.nanargmax(np.where(np.isfinite(lnp), lnp, -np.inf))].copy())
    hip_ctx.set_regions(xs, fs, ns, Ks, mode=vo.MODE_VOIGT4)
    active = np.array([1, 1, 0, 1], dtype=np.uint8)
    best, lnp, chi, its = hip_ctx.map_all(starts, iterlim=250, tol=1e-3, active=active, xtol=1e-3, maxfun=1000)
    # PyMC's call (fmin(maxiter=iterlim, ftol=tol), xtol and maxfun at scipy's defaults) is the wrapper's default
    best_d, lnp_d, _, its_d = hip_ctx.map_all(starts, iterlim=250, tol=1e-3, active=active)
    assert np.array_equal(best[2], starts[2]) and its[2] == 0          # inactive: returned unchanged
    for r in (0, 1, 3):
        def neg(t, r=r):
            v = hip_ctx.lnprob(t, region=r)[0]
            return -v if np.isfinite(v) else 1e300
        xd, fd, itd, _, _ = fmin(neg, starts[r], ftol=1e-3, maxiter=250, disp=False, full_output=True)
        assert itd == its_d[r] + 1 and np.allclose(best_d[r], xd, rtol=1e-13, atol=0) and np.isclose(lnp_d[r], -fd, rtol=1e-13)
        xopt, fopt, it, calls, flag = fmin(neg, starts[r], xtol=1e-3, ftol=1e-3, maxiter=250, maxfun=1000, disp=False,
                                           full_output=True)
        assert it == its[r] + 1, (r, it, its[r])          # fmin counts iterations from 1
        assert np.allclose(best[r], xopt, rtol=1e-13, atol=0), (r, best[r], xopt)
        assert np.isclose(lnp[r], -fopt, rtol=1e-13)
        l1, c1 = hip_ctx.lnprob(best[r], region=r, return_chi2=True)
        assert l1[0] == lnp[r] and c1[0] == chi[r]
        assert lnp[r] >= hip_ctx.lnprob(starts[r], region=r)[0]
    # lnprob_all = lnprob region by region
    blocks = [np.stack([s, s * 1.01, s * 0.99]) for s in starts]
    la, ca = hip_ctx.lnprob_all(blocks, return_chi2=True)
    for r in range(4):
        l1, c1 = hip_ctx.lnprob(blocks[r], region=r, return_chi2=True)
        assert np.array_equal(la[r], l1) and np.array_equal(ca[r], c1, equal_nan=True)


@pytest.mark.parametrize("mode,sd", [(vo.MODE_GAUSS3, True), (vo.MODE_VOIGT4, True), (vo.MODE_VOIGT4, False)])
def test_map_search_on_device_equals_host_driven(hip_ctx, mode, sd):
    """The device-resident MAP search (k_map_search: one workgroup per region, the whole Nelder-Mead search in
    ONE launch, vpfits.py:352-358, 426) against the host-driven search of csrc/map_search.hpp (one launch +
    one synchronisation per iteration; "map_device" = 0), which test_map_all_follows_scipy_fmin ties to
    scipy's fmin: same optimum BIT FOR BIT, same iteration count, for every region of a mixed context --
    short one- and two-line regions, a blend (per-walker Taylor tables), a region of more than 16 lines, an
    inactive region -- under fmin's default limits (maxfun = 200 per dimension) and under tight ones
    (maxiter, maxfun hit before convergence; shrink steps happen on the way)."""
    rng = np.random.default_rng(5 + mode + 2 * sd)
    shapes = [(30, 1), (44, 2), (51, 4), (160, 3), (23, 1), (90, 6)]
    if hip_ctx.packing_request == 256:
        shapes = [(2100, 3), (2300, 5)]
    elif hip_ctx.packing_request not in (16, 65):
        shapes.append((120, 18))
    q = 3 if mode == vo.MODE_GAUSS3 else 4
    xs, fs, ns, Ks, starts = [], [], [], [], []
    for P, K in shapes:
        x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
        c = rng.uniform(x[0] * 0.8, x[-1] * 0.8, K)
        w = rng.uniform(1.5, 0.08 * P + 2.0, K)
        tau = sum(rng.uniform(0.3, 2.0) * np.exp(-0.5 * ((x - ck) / wk) ** 2) for ck, wk in zip(c, w))
        noise = np.full(P, 0.02)
        xs.append(x); fs.append(np.exp(-tau) + rng.normal(0, 0.02, P)); ns.append(np.ones(P) if sd else noise); Ks.append(K)
        th = np.empty((K, q))
        th[:, 0] = rng.uniform(0.2, 1.5, K)
        th[:, 1] = c + rng.normal(0, 1.0, K)
        th[:, 2] = w * rng.uniform(0.7, 1.4, K) if q == 3 else rng.uniform(0.2, 2.0, K)
        if q == 4:
            th[:, 3] = 2.355 * w * rng.uniform(0.7, 1.4, K)
        starts.append(np.concatenate([th.ravel(), [0.05]]) if sd else th.ravel())
    hip_ctx.set_regions(xs, fs, ns, Ks, mode=mode, sample_sd=sd)
    active = np.ones(len(shapes), dtype=np.uint8)
    active[1] = 0
    try:
        for kw in (dict(iterlim=400, tol=1e-3), dict(iterlim=60, tol=1e-8, xtol=1e-8), dict(iterlim=10 ** 6, tol=1e-9, xtol=1e-9, maxfun=150)):
            hip_ctx.set_option("map_device", 1)
            b1, l1, c1, i1 = hip_ctx.map_all(starts, active=active, **kw)
            hip_ctx.set_option("map_device", 0)
            b0, l0, c0, i0 = hip_ctx.map_all(starts, active=active, **kw)
            assert np.array_equal(i1, i0), (kw, i1, i0)
            for r in range(len(shapes)):
                assert np.array_equal(b1[r], b0[r]), (kw, r)
            assert np.array_equal(l1, l0) and np.array_equal(c1, c0, equal_nan=True)
            assert i1[1] == 0 and np.array_equal(b1[1], starts[1]) and i1[0] > 5
            assert np.all(l1 >= hip_ctx.lnprob_all([s_[None, :] for s_ in starts])[:, 0])
    finally:
        hip_ctx.set_option("map_device", 1)


def _mixed_short_context(rng, variant, W, with_xl):
    """regions of a mixed spectrum-like context (every launch class) with W walkers each; variant 0: (amplitude,
    centroid, L, G), 1: Gaussian components, 2: variant 0 + the free precision sd, 3: (N, b, z)"""
    C_LIGHT, SIGMA0, LINE, PIX_HZ = 2.98e8, 0.0263, 1215.67, 4.0e10
    fps = 2.0 * np.sqrt(2.0 * np.log(2.0))
    shapes = [(30, 1), (44, 2), (51, 4), (160, 3), (23, 1), (90, 6), (300, 5), (36, 2)] + ([(120, 18)] if with_xl else [])
    xs, fs, ns, Ks, ths, nbz = [], [], [], [], [], []
    for P, K in shapes:
        x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
        c = rng.uniform(x[0] * 0.8, x[-1] * 0.8, K)
        w = rng.uniform(1.5, 0.05 * P + 2.0, K)
        tau = sum(rng.uniform(0.3, 2.0) * np.exp(-0.5 * ((x - ck) / wk) ** 2) for ck, wk in zip(c, w))
        th = np.empty((W, K, 4))
        th[:, :, 0] = rng.uniform(0.2, 1.5, (W, K))
        th[:, :, 1] = c + rng.normal(0, 1.5, (W, K))
        th[:, :, 2] = 10.0 ** rng.uniform(-2, 0.5, (W, K))
        th[:, :, 3] = fps * w * rng.uniform(0.6, 1.6, (W, K))
        th[: W // 8, 0, 0] = -0.1                                  # a few walkers start outside the prior
        xs.append(x); fs.append(np.exp(-tau) + rng.normal(0, 0.02, P)); Ks.append(K)
        ns.append(np.ones(P) if variant == 2 else np.full(P, 0.02))
        if variant == 1:
            t = np.stack([th[:, :, 0], th[:, :, 1], th[:, :, 3] / fps], axis=2).reshape(W, 3 * K)
        elif variant == 2:
            t = np.hstack([th.reshape(W, 4 * K), rng.uniform(0.01, 0.2, (W, 1))])
        elif variant == 3:
            nu_mid = C_LIGHT / (1225.0 * 1e-10)
            sig_hz = th[:, :, 3] * PIX_HZ / fps
            t = np.stack([th[:, :, 0] * sig_hz * np.sqrt(2 * np.pi) / SIGMA0, (LINE * 1e-10 * sig_hz * 2.355 / np.sqrt(2)) * 1e-3,
                          ((C_LIGHT / (nu_mid + PIX_HZ * th[:, :, 1])) / 1e-10 - LINE) / LINE], axis=2).reshape(W, 3 * K)
            nbz.append([float(10.0 ** rng.uniform(-1, 0.5)), LINE, nu_mid, PIX_HZ])
        else:
            t = th.reshape(W, 4 * K)
        ths.append(np.ascontiguousarray(t))
    kw = [dict(mode=vo.MODE_VOIGT4), dict(mode=vo.MODE_GAUSS3), dict(mode=vo.MODE_VOIGT4, sample_sd=True), dict(mode=vo.MODE_NBZ3)][variant]
    if variant == 3:
        kw["nbz"] = np.array(nbz)
    return xs, fs, ns, Ks, ths, kw


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("W", [32, 70])
def test_resident_step_loop_equals_launch_per_half_step(hip_ctx, variant, W):
    """The device-resident step loop (k_run_resident: one workgroup per region, the whole loop of vpfits.py:361-395 /
    420-425 in ONE launch per launch class, half-step barrier = __syncthreads, draws one half-step ahead by an extra
    wavefront) against vamp_sampler_run with one launch per half-step ("resident" = 0): the same chain, log-posterior
    chain, acceptance counts and final state BIT FOR BIT, on a mixed multi-region context (one- and two-line regions,
    blends with per-walker tables, a region of 18 lines) for every parameterisation, an ensemble that fills its
    wavefronts (W = 32) and one that does not (W = 70: 35 movers), thinning, and a second run that continues the
    first (step counter, draw keys)."""
    if hip_ctx.packing_request == 256:
        pytest.skip("workgroup-per-walker shapes are not resident (long regions are not launch-bound)")
    rng = np.random.default_rng(40 + variant)
    xs, fs, ns, Ks, ths, kw = _mixed_short_context(rng, variant, W, with_xl=hip_ctx.packing_request not in (16, 65))
    out = {}
    try:
        for resident in (1, 0):
            hip_ctx.set_option("resident", 2 * resident)      # 2: every context the kernel can run, not only where it pays
            hip_ctx.set_regions(xs, fs, ns, Ks, **kw)
            hip_ctx.sampler_init(ths, seed=77, split_block=W if W == 32 else 14)
            a = hip_ctx.run_flat(7, thin=3)
            b = hip_ctx.run_flat(4, thin=1)
            hip_ctx.run(3, store_chain=False)
            out[resident] = (a, b, hip_ctx.get_state())
    finally:
        hip_ctx.set_option("resident", 1)
    (a1, b1, s1), (a0, b0, s0) = out[1], out[0]
    for x1, x0 in ((a1, a0), (b1, b0)):
        assert x1[0].shape == x0[0].shape and x1[0].shape[0] in (2, 4)
        assert np.array_equal(x1[0], x0[0]) and np.array_equal(x1[1], x0[1]) and np.array_equal(x1[2], x0[2])
    assert s1[3] == s0[3] == 14
    for r in range(len(xs)):
        assert np.array_equal(s1[0][r], s0[0][r]) and np.array_equal(s1[1][r], s0[1][r]) and np.array_equal(s1[2][r], s0[2][r]), r
    assert a1[2].sum() > 0 and np.isfinite(a1[1]).any()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_long_regions_match_oracle(hip_ctx, seed):
    """Seeded random long regions (full tiles + ragged tail, 1-16 lines) with line widths and
    dampings spread over decades -- lines far narrower than a pixel, lines broader than the region,
    heavily damped lines, centres at the region's edge, crowded blends: every branch of the tile
    code (far-field interpolant, Taylor tables, fractions, cap, near-axis rule in the tail)
    against the oracle's scipy.wofz restatement."""
    if hip_ctx.packing_request in (16, 65):
        pytest.skip("long regions: one walker per wavefront or workgroup")
    rng = np.random.default_rng(1000 + seed)
    worst = 0.0
    for case in range(6):
        P = int(rng.choice([512, 1300, 2048, 2500, 4096]))
        K = int(rng.integers(1, 17))
        x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
        if case % 3 == 2:
            x = np.cumsum(rng.uniform(0.5, 1.5, P))            # uneven pixel spacing
            x -= x.mean()
        W = 8
        th = np.empty((W, K, 4))
        th[:, :, 0] = 10.0 ** rng.uniform(-2, 1.7, (W, K))                      # amplitude
        th[:, :, 1] = rng.uniform(x[0], x[-1], (W, K))                          # centroid
        th[:, :, 1][:, 0] = np.where(rng.random(W) < 0.5, x[0], x[-1])          # one line on the edge
        th[:, :, 2] = 10.0 ** rng.uniform(-6, np.log10(0.4 * (x[-1] - x[0])), (W, K))   # L_fwhm
        th[:, :, 3] = 10.0 ** rng.uniform(-1.3, np.log10(0.4 * (x[-1] - x[0])), (W, K))  # G_fwhm
        if K >= 4:
            th[:, 1:4, 1] = th[:, 1:2, 1] + rng.normal(0, 3.0, (W, 3))          # a crowded blend
            th[:, 1:4, 1] = np.clip(th[:, 1:4, 1], x[0], x[-1])
        th = th.reshape(W, 4 * K)
        noise = np.full(P, 0.05)
        flux = np.clip(1.0 + rng.normal(0, 0.05, P), 0, None)
        hip_ctx.set_regions(x, flux, noise, K, mode=vo.MODE_VOIGT4)
        r = vo.Region(x=x, flux=flux, noise=noise, n_comp=K, mode=vo.MODE_VOIGT4)
        want = vo.log_prob_batch_fast(r, th)
        got = hip_ctx.lnprob(th)
        assert np.array_equal(np.isfinite(want), np.isfinite(got)), (seed, case)
        fin = np.isfinite(want)
        assert fin.any()
        err = np.abs(got[fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin]))
        worst = max(worst, err.max())
        assert err.max() <= 1e-9, (seed, case, P, K, err.max())
    assert worst <= 1e-9


@pytest.mark.parametrize("grid", ["ascending", "descending", "uneven"])
def test_full_size_dispersed_walkers_match_oracle(hip_ctx, grid):
    """The headline's FULL size -- P = 16 384 pixels, K = 16 lines, 64 tiles of 256 pixels, 16 tiles
    per wavefront of the workgroup-per-walker shape -- on walkers that are NOT converged: widths and
    dampings spread over decades (every walker has its own near / far mix per tile, narrow-line caps,
    more than 8 far lines per tile and fewer), one line on the region's edge, one crowded blend, one
    line far narrower than a pixel.  Log-posterior of 8 such walkers against the oracle's scipy.wofz
    restatement (vpfits.py:57-76, 334-341), |delta| <= 1e-9 max(1, |lnprob|), and one stretch step
    against the oracle's sampler -- on an ascending, a descending and an unevenly spaced grid."""
    if hip_ctx.packing_request in (16, 65):
        pytest.skip("long regions: one walker per wavefront or workgroup")
    rng = np.random.default_rng({"ascending": 31, "descending": 32, "uneven": 33}[grid])
    P, K, W = 16384, 16, 8
    x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
    if grid == "uneven":
        x = np.cumsum(rng.uniform(0.5, 1.5, P))
        x -= x.mean()
    span = x[-1] - x[0]
    th = np.empty((W, K, 4))
    th[:, :, 0] = 10.0 ** rng.uniform(-2, 1.7, (W, K))                                  # amplitude
    th[:, :, 1] = rng.uniform(x[0], x[-1], (W, K))                                      # centroid
    th[:, 0, 1] = np.where(rng.random(W) < 0.5, x[0], x[-1])                            # one line on the edge
    th[:, :, 2] = 10.0 ** rng.uniform(-6, np.log10(0.4 * span), (W, K))                 # L_fwhm
    th[:, :, 3] = 10.0 ** rng.uniform(-1.3, np.log10(0.4 * span), (W, K))               # G_fwhm
    th[:, 1:4, 1] = np.clip(th[:, 1:2, 1] + rng.normal(0, 3.0, (W, 3)), x[0], x[-1])    # a crowded blend
    th[:, 1:4, 3] = 10.0 ** rng.uniform(0.5, 1.5, (W, 3))
    th[:, 4, 3] = 10.0 ** rng.uniform(-1.3, -0.7, W)                                    # a sub-pixel line
    th[:, 4, 2] = 10.0 ** rng.uniform(-6, -2, W)
    th[:4, 5:, 3] = 10.0 ** rng.uniform(1.0, 2.3, (4, K - 5))                           # half the walkers: mostly far lines
    th[:4, 5:, 2] = 10.0 ** rng.uniform(-1, 1.3, (4, K - 5))
    th = th.reshape(W, 4 * K)
    noise = np.full(P, 0.05)
    flux = np.clip(1.0 + rng.normal(0, 0.05, P), 0, None)
    # the oracle restates the reference, whose grids ascend (vpspectrum.py:274-277; bounds from x[0], x[-1]); the
    # ABI takes either direction: the descending case uploads the mirrored arrays of the same region
    r = vo.Region(x=x, flux=flux, noise=noise, n_comp=K, mode=vo.MODE_VOIGT4)
    if grid == "descending":
        hip_ctx.set_regions(x[::-1].copy(), flux[::-1].copy(), noise[::-1].copy(), K, mode=vo.MODE_VOIGT4)
    else:
        hip_ctx.set_regions(x, flux, noise, K, mode=vo.MODE_VOIGT4)
    want = vo.log_prob_batch_fast(r, th)
    got = hip_ctx.lnprob(th)
    assert np.isfinite(want).all() and np.isfinite(got).all()
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert err.max() <= 1e-9, (grid, err)
    hip_ctx.sampler_init(th, seed=5, split_block=W)
    res = hip_ctx.run(1)
    fn = lambda q: vo.log_prob_batch_fast(r, q)
    chain, lchain, nacc = vo.run_sampler(fn, th, want, 1, seed=5, block=W)
    assert np.array_equal(res["n_accept"], nacc), grid
    assert np.allclose(res["chain"], chain, rtol=1e-10, atol=1e-12), grid
    fin = np.isfinite(lchain[-1])
    assert np.allclose(res["lnprob"][-1][fin], lchain[-1][fin], rtol=1e-9, atol=1e-9), grid


@pytest.mark.parametrize("case", ["threshold", "prior"])
def test_lines_wider_than_a_tile_match_oracle(hip_ctx, case):
    """Round 4: near lines far wider than a tile are evaluated at the tile's 16 Chebyshev nodes and reach the pixels
    through the tile's interpolant (ff_wide_nodes) when half a tile is at most 3/4 in the line's own z, i.e. from
    G_fwhm ~ 284 px on a unit grid.  Headline size (P = 16 384, K = 16): (threshold) widths within 10 % above the
    switch -- where the interpolation error is largest -- mixed with widths just below it (evaluated per pixel), dampings
    from 1e-6 to 300 px, optical depths up to ~50; (prior) every parameter drawn from the priors of vpfits.py:283-297 as
    bench.py --ensemble prior does: every line wide, nothing far.  Log-posterior against the oracle to 1e-9, and one
    stretch step against the oracle's sampler."""
    if hip_ctx.packing_request in (16, 65):
        pytest.skip("long regions: one walker per wavefront or workgroup")
    rng = np.random.default_rng(71 if case == "threshold" else 72)
    P, K, W = 16384, 16, 8
    x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
    th = np.empty((W, K, 4))
    th[:, :, 1] = rng.uniform(x[0], x[-1], (W, K))
    if case == "threshold":
        import os
        wide_max = float(os.environ.get("VAMP_TEST_WIDE_MAX", "0.75"))       # VAMP_WIDE_MAX of the library under test
        edge = 128.0 * 2.0 * np.sqrt(np.log(2.0)) / wide_max      # G at which half a tile is wide_max in z
        th[:, :, 3] = edge * np.where(rng.random((W, K)) < 0.6, 1.0 + rng.uniform(0.0, 0.1, (W, K)), 1.0 - rng.uniform(0.0, 0.1, (W, K)))
        th[:, :, 2] = 10.0 ** rng.uniform(-6, 2.5, (W, K))
        th[:, :, 0] = 10.0 ** rng.uniform(-2, 0.7, (W, K))
        th[:, 0, 0] = 50.0                                        # one saturated wide line per walker
    else:
        fw = (x[-1] - x[0]) / 2.0 * 2.0 * np.sqrt(2.0 * np.log(2.0))
        th[:, :, 3] = rng.uniform(0.0, fw, (W, K))
        th[:, :, 2] = rng.uniform(0.0, fw, (W, K))
        th[:, :, 0] = rng.gamma(2.0, 1.0, (W, K))
    th = th.reshape(W, 4 * K)
    noise = np.full(P, 0.05)
    flux = np.clip(1.0 + rng.normal(0, 0.05, P), 0, None)
    r = vo.Region(x=x, flux=flux, noise=noise, n_comp=K, mode=vo.MODE_VOIGT4)
    hip_ctx.set_regions(x, flux, noise, K, mode=vo.MODE_VOIGT4)
    want = vo.log_prob_batch_fast(r, th)
    got = hip_ctx.lnprob(th)
    assert np.isfinite(want).all() and np.isfinite(got).all()
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert err.max() <= 1e-9, (case, err)
    tau, fl = hip_ctx.model(th[0])                                 # (k_model evaluates every pixel directly: the same answer)
    assert np.allclose(fl, vo.model_flux(r, th[0]), rtol=1e-12, atol=1e-300)
    hip_ctx.sampler_init(th, seed=9, split_block=W)
    res = hip_ctx.run(1)
    chain, lchain, nacc = vo.run_sampler(lambda q: vo.log_prob_batch_fast(r, q), th, want, 1, seed=9, block=W)
    assert np.array_equal(res["n_accept"], nacc) and np.allclose(res["chain"], chain, rtol=1e-10, atol=1e-12), case
    print("wide lines (%s): worst lnprob error %.2e" % (case, err.max()))


FAR_FIELD_CASES = {   # G_fwhm range [px], log10 L range, amplitudes up to (< 0: log-uniform up to 10^-v)
    "headline-like": (20, 200, -3, 0, 3.0), "damped": (5, 120, 0, 2.0, 10.0), "narrow strong": (2, 40, -1, 1.5, 50.0),
    "mixed": (1, 400, -4, 2.5, 20.0), "damped wings": (5, 30, 1.5, 2.5, -2.5)}


@pytest.mark.parametrize("case", list(FAR_FIELD_CASES))
def test_far_field_on_well_fitted_data(hip_ctx, case):
    """Round 4 moved the far field's borders inward: far from 2 tile half-widths (was 4), lines between far and near
    and lines far wider than a tile at the nodes through their Taylor tables.  Its error is relative to a line's
    value at the tile, so it weighs most where |lnprob| is smallest: data that IS the model of 16 lines plus noise at
    S/N 200, walkers 1e-4 around the truth (|lnprob| ~ P/2, headline size) -- lines of every width and damping, strong
    damped wings (optical depth ~1 hundreds of pixels from the centre), narrow lines of amplitude 50.  Log-posterior
    against the oracle: the bar is 1e-9; measured 2e-16 .. 1e-13 (profiles/r04_h_ff_dist.txt), asserted at 1e-11."""
    if hip_ctx.packing_request in (16, 65):
        pytest.skip("long regions: one walker per wavefront or workgroup")
    g_lo, g_hi, l_lo, l_hi, a_hi = FAR_FIELD_CASES[case]
    P, K, W, sd = 16384, 16, 8, 0.005
    x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
    rng = np.random.default_rng(11)
    t = np.empty((K, 4))
    t[:, 1] = rng.uniform(x[0], x[-1], K)
    t[:, 3] = rng.uniform(g_lo, g_hi, K)
    t[:, 2] = 10.0 ** rng.uniform(l_lo, l_hi, K)
    t[:, 0] = rng.uniform(0.3, a_hi, K) if a_hi > 0 else 10.0 ** rng.uniform(0.0, -a_hi, K)
    truth = t.reshape(-1)
    noise = np.full(P, sd)
    r0 = vo.Region(x=x, flux=np.ones(P), noise=noise, n_comp=K, mode=vo.MODE_VOIGT4)
    flux = vo.model_flux(r0, truth) + rng.normal(0, sd, P)
    r = vo.Region(x=x, flux=flux, noise=noise, n_comp=K, mode=vo.MODE_VOIGT4)
    th = truth[None, :] * (1.0 + 1e-4 * rng.standard_normal((W, 4 * K)))
    hip_ctx.set_regions(x, flux, noise, K, mode=vo.MODE_VOIGT4)
    want = vo.log_prob_batch_fast(r, th)
    got = hip_ctx.lnprob(th)
    assert np.isfinite(want).all() and np.abs(want).max() < 2.0 * P
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert err.max() <= 1e-11, (case, err)


@pytest.mark.parametrize("packing", [0, 64])
def test_full_size_dispersed_walkers_fp32(packing):
    """The same non-converged walkers at the headline's full size through the fp32 path -- single-precision
    Taylor rows in the line cores, W4 regions I / II in the wings, the 8-node far field (packing 0 = the
    workgroup shape) and plain W4 everywhere (packing 64) -- against the fp64 oracle: SURVEY 8d's fp32
    tolerance |delta chi^2| / chi^2 <= 1e-3, and the identical -inf pattern."""
    import vamp_amd
    rng = np.random.default_rng(31)
    P, K, W = 16384, 16, 8
    x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
    span = x[-1] - x[0]
    th = np.empty((W, K, 4))
    th[:, :, 0] = 10.0 ** rng.uniform(-2, 0.7, (W, K))
    th[:, :, 1] = rng.uniform(x[0], x[-1], (W, K))
    th[:, 0, 1] = np.where(rng.random(W) < 0.5, x[0], x[-1])
    th[:, :, 2] = 10.0 ** rng.uniform(-6, np.log10(0.4 * span), (W, K))
    th[:, :, 3] = 10.0 ** rng.uniform(-1.3, np.log10(0.4 * span), (W, K))
    th[:, 1:4, 1] = np.clip(th[:, 1:2, 1] + rng.normal(0, 3.0, (W, 3)), x[0], x[-1])
    th[:, 1:4, 3] = 10.0 ** rng.uniform(0.5, 1.5, (W, 3))
    th[:, 4, 3] = 10.0 ** rng.uniform(-1.3, -0.7, W)
    th[:4, 5:, 3] = 10.0 ** rng.uniform(1.0, 2.3, (4, K - 5))
    th[:4, 5:, 2] = 10.0 ** rng.uniform(-1, 1.3, (4, K - 5))
    th = th.reshape(W, 4 * K)
    noise = np.full(P, 0.05)
    flux = np.clip(1.0 + rng.normal(0, 0.05, P), 0, None)
    r = vo.Region(x=x, flux=flux, noise=noise, n_comp=K, mode=vo.MODE_VOIGT4)
    want, chi_w = vo.log_prob_batch_fast(r, th), None
    with vamp_amd.HipContext(device=0, dtype=vamp_amd.F32) as ctx:
        ctx.set_packing(packing)
        ctx.set_regions(x, flux, noise, K, mode=vo.MODE_VOIGT4)
        got, chi = ctx.lnprob(th, return_chi2=True)
    assert np.array_equal(np.isfinite(want), np.isfinite(got))
    fin = np.isfinite(want)
    with vamp_amd.HipContext(device=0) as ctx:                    # chi^2 itself from the fp64 device path (== oracle to 1e-11)
        ctx.set_regions(x, flux, noise, K, mode=vo.MODE_VOIGT4)
        ref, chi_ref = ctx.lnprob(th, return_chi2=True)
    assert np.max(np.abs(ref[fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin]))) <= 1e-9
    assert np.max(np.abs(chi[fin] - chi_ref[fin]) / chi_ref[fin]) <= 1e-3, packing


@pytest.mark.parametrize("seed", [8, 10, 53, 58, 108])
def test_fp32_heavily_damped_lines_in_the_far_field(seed):
    """Regression (round 3, found by tests/soak_long_regions.py f32): a line with a sub-pixel Gaussian width and a
    Lorentzian width of hundreds of pixels has y = L sqrt(ln 2) / G ~ 1e4 -- a broad Lorentzian with a large optical
    depth far from its centre.  In the far-field node pass the lanes of a wavefront hold different lines and share one
    W4 region; region II used to clamp x at 3e4 to keep |t|^8 finite, which for such a line is not a no-op, and the
    log-posterior was off by up to 44 %.  The soak's regions for the seeds that failed, both fp32 launch shapes."""
    import vamp_amd
    rng = np.random.default_rng(7000 + seed)
    P = int(rng.choice([512, 768, 1300, 2048, 2560, 4096, 6144]))
    K = int(rng.integers(1, 17))
    kind = seed % 4
    if kind == 0:
        x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
    elif kind == 1:
        x = np.cumsum(rng.uniform(0.5, 1.5, P))
    elif kind == 2:
        x = np.cumsum(np.geomspace(1.0, 3.0, P))
    else:
        x = -(np.arange(P, dtype=np.float64) - (P - 1) / 2.0) * 0.37
    x = x - x.mean()
    span = abs(x[-1] - x[0])
    W = 8
    th = np.empty((W, K, 4))
    th[:, :, 0] = 10.0 ** rng.uniform(-2, 1.7, (W, K))
    th[:, :, 1] = rng.uniform(min(x[0], x[-1]), max(x[0], x[-1]), (W, K))
    th[:, :, 2] = 10.0 ** rng.uniform(-6, np.log10(0.4 * span), (W, K))
    th[:, :, 3] = 10.0 ** rng.uniform(-1.3, np.log10(0.4 * span), (W, K))
    th = th.reshape(W, 4 * K)
    noise = np.full(P, 0.05)
    flux = np.clip(1.0 + rng.normal(0, 0.05, P), 0, None)
    r = vo.Region(x=x, flux=flux, noise=noise, n_comp=K, mode=vo.MODE_VOIGT4)
    want = vo.log_prob_batch_fast(r, th)
    for packing in (64, 256):
        with vamp_amd.HipContext(device=0, dtype=vamp_amd.F32) as ctx:
            ctx.set_packing(packing)
            ctx.set_regions(x, flux, noise, K, mode=vo.MODE_VOIGT4)
            got = ctx.lnprob(th)
        assert np.array_equal(np.isfinite(want), np.isfinite(got))
        fin = np.isfinite(want)
        assert np.max(np.abs(got[fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin]))) <= 1e-3, (seed, packing)


def test_regions_of_more_than_16_lines(hip_ctx):
    """The reference sets no limit on the lines of a region (it plans for more than 15 and more than
    22.5, vpspectrum.py:287-294); the fast launch shapes hold 16, regions of 17 .. 32 lines
    (VAMP_MAX_COMPONENTS) take a plain shape in a launch class of their own.  One context with a
    20-line, a 32-line and a 3-line region: log-posteriors and three stretch steps against the oracle,
    with the free precision sd of the reference's likelihood on (D = 4 K + 1 = 129 for 32 lines)."""
    if hip_ctx.packing_request in (16, 65):
        pytest.skip("packings 16 and 65 hold 8 lines")
    rng = np.random.default_rng(77)
    xs, fs, ns, Ks, ths, regs = [], [], [], [], [], []
    W = 280                                                     # >= 2 D + 2 for D = 129
    for P, K in ((300, 20), (700, 32), (64, 3)):
        x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
        th = np.empty((W, K, 4))
        th[:, :, 0] = rng.uniform(0.05, 1.5, (W, K))
        th[:, :, 1] = rng.uniform(x[0], x[-1], (W, K))
        th[:, :, 2] = 10.0 ** rng.uniform(-3, 0.8, (W, K))
        th[:, :, 3] = 10.0 ** rng.uniform(-0.3, 1.6, (W, K))
        th = np.hstack([th.reshape(W, 4 * K), rng.uniform(0.01, 0.2, (W, 1))])
        noise = np.ones(P)
        flux = np.clip(1.0 + rng.normal(0, 0.05, P), 0, None)
        xs.append(x); fs.append(flux); ns.append(noise); Ks.append(K); ths.append(th)
        regs.append(vo.Region(x=x, flux=flux, noise=noise, n_comp=K, mode=vo.MODE_VOIGT4, sample_sd=True))
    hip_ctx.set_regions(xs, fs, ns, Ks, mode=vo.MODE_VOIGT4, sample_sd=True)
    assert hip_ctx.ndims == [81, 129, 13]
    want = [vo.log_prob_batch_fast(r, t) for r, t in zip(regs, ths)]
    got = hip_ctx.lnprob_all(ths)
    for r in range(3):
        assert np.isfinite(want[r]).all()
        assert np.max(np.abs(got[r] - want[r]) / np.maximum(1.0, np.abs(want[r]))) <= 1e-9, r
        assert np.array_equal(hip_ctx.lnprob(ths[r], region=r), got[r])
    hip_ctx.sampler_init(ths, seed=3, split_block=W)
    res = hip_ctx.run(3)
    for r in range(3):
        fn = lambda q, r=r: vo.log_prob_batch_fast(regs[r], q)
        chain, _, nacc = vo.run_sampler(fn, ths[r], want[r], 3, seed=3, block=W, region=r, walker_off=r * W)
        assert np.array_equal(res["n_accept"][r], nacc), r
        assert np.allclose(res["chain"][r], chain, rtol=1e-10, atol=1e-12), r
    tau, flux = hip_ctx.model(ths[1][0], region=1)
    assert np.allclose(tau, np.array(list(vo.component_taus(regs[1], ths[1][0]))), rtol=1e-12, atol=1e-300)
    assert np.allclose(flux, vo.model_flux(regs[1], ths[1][0]), rtol=1e-12)


def test_descending_grid_long_region(hip_ctx):
    """A region uploaded in descending coordinate order (the reference flips to ascending
    frequency, vpspectrum.py:274-277, but the ABI takes either direction): same log-posterior as
    the ascending upload and as the oracle, through the tile code of long regions."""
    if hip_ctx.packing_request == 16:
        pytest.skip("long region: one walker per wavefront or workgroup")
    from bench import make_workload
    wl = make_workload(P=2304, K=7, W=16, seed=31, nbz=False)
    hip_ctx.set_regions(wl["x"], wl["flux"], wl["noise"], 7, mode=vo.MODE_VOIGT4)
    up = hip_ctx.lnprob(wl["theta0"])
    hip_ctx.set_regions(wl["x"][::-1].copy(), wl["flux"][::-1].copy(), wl["noise"][::-1].copy(), 7, mode=vo.MODE_VOIGT4)
    down = hip_ctx.lnprob(wl["theta0"])
    r = vo.Region(x=wl["x"], flux=wl["flux"], noise=wl["noise"], n_comp=7, mode=vo.MODE_VOIGT4)
    want = vo.log_prob_batch_fast(r, wl["theta0"])
    for got in (up, down):
        assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) <= 1e-9


def test_long_region_all_modes(hip_ctx):
    """The long-region launch shapes in every parameterisation and likelihood form: Gaussian
    components (no far field, tiles dealt to the wavefronts), Voigt with the reference's free
    precision sd ~ U(0,1) as last dimension, and (N, b, z) -- log-posterior against the oracle and
    a few stretch steps against the oracle's sampler."""
    if hip_ctx.packing_request == 16:
        pytest.skip("long regions: one walker per wavefront or workgroup")
    from bench import make_workload
    cases = []
    wl = make_workload(P=2304, K=5, W=32, seed=41, nbz=False)
    th4 = wl["theta0"].reshape(32, 5, 4)
    thg = np.stack([th4[:, :, 0], th4[:, :, 1], th4[:, :, 3] / 2.3548200450309493], axis=2).reshape(32, 15)
    cases.append(("gauss", dict(mode=vo.MODE_GAUSS3), wl, thg))
    sd = np.random.default_rng(2).uniform(0.005, 0.05, (32, 1))
    cases.append(("voigt+sd", dict(mode=vo.MODE_VOIGT4, sample_sd=True), wl, np.hstack([wl["theta0"], sd])))
    wn = make_workload(P=2560, K=5, W=32, seed=43, nbz=True)
    cases.append(("nbz", dict(mode=vo.MODE_NBZ3), wn, wn["theta0"]))
    for name, kw, w, th in cases:
        nbz = w["nbz"] if kw["mode"] == vo.MODE_NBZ3 else None
        hip_ctx.set_regions(w["x"], w["flux"], w["noise"], 5, nbz=nbz, **kw)
        r = vo.Region(x=w["x"], flux=w["flux"], noise=w["noise"], n_comp=5, **kw)
        if nbz is not None:
            r.l_fixed, r.line, r.x_origin, r.x_scale = [float(v) for v in nbz[0]]
        want = vo.log_prob_batch_fast(r, th)
        got = hip_ctx.lnprob(th)
        assert np.isfinite(want).all(), name
        assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) <= 1e-9, name
        hip_ctx.sampler_init(th, seed=7, split_block=16)
        res = hip_ctx.run(3)
        fn = lambda q, r=r: vo.log_prob_batch_fast(r, q)
        chain, _, nacc = vo.run_sampler(fn, th, want, 3, seed=7, block=16)
        assert np.allclose(res["chain"], chain, rtol=1e-10, atol=1e-12), name
        assert np.array_equal(res["n_accept"], nacc), name


def test_long_region_symmetries(hip_ctx):
    """Size-independent properties of the tile code on a 4096-pixel region (no CPU reference
    needed): chi^2 is unchanged when the grid and the centroids are translated, when grid,
    centroids and widths are scaled together, and when the spectrum is mirrored (a descending
    grid).  Each transformation moves the tile boundaries relative to the lines, so the far /
    near classification, the table intervals and the interpolation nodes all change."""
    if hip_ctx.packing_request in (16, 65):
        pytest.skip("long region: one walker per wavefront or workgroup")
    from bench import make_workload
    wl = make_workload(P=4096, K=9, W=24, seed=77, nbz=False)
    x, f, n, th = wl["x"], wl["flux"], wl["noise"], wl["theta0"].reshape(24, 9, 4)

    def chi2(xg, fg, ng, t):
        hip_ctx.set_regions(xg, fg, ng, 9, mode=vo.MODE_VOIGT4)
        lnp, c = hip_ctx.lnprob(t.reshape(24, 36), return_chi2=True)
        assert np.isfinite(lnp).all()
        return c

    base = chi2(x, f, n, th)
    t = th.copy(); t[:, :, 1] += 137.3
    assert np.allclose(chi2(x + 137.3, f, n, t), base, rtol=2e-11)
    t = th.copy(); t[:, :, 1:] *= 0.37
    assert np.allclose(chi2(x * 0.37, f, n, t), base, rtol=2e-11)
    t = th.copy(); t[:, :, 1] *= -1.0
    assert np.allclose(chi2(-x, f, n, t), base, rtol=2e-11)
    t = th.copy(); t[:, :, 1] = -t[:, :, 1] + 5.5
    assert np.allclose(chi2((-x + 5.5)[::-1].copy(), f[::-1].copy(), n[::-1].copy(), t), base, rtol=2e-11)


@pytest.mark.parametrize("script,args", [("soak_long_regions.py", ["24"]), ("soak_long_regions.py", ["24", "f32"]),
                                         ("soak_short_regions.py", ["10"]), ("soak_short_regions.py", ["10", "f32"]),
                                         ("soak_sampler.py", ["8"]), ("soak_resident_map.py", ["24"])])
def test_soaks_in_short(script, args):
    """The developer soaks (tests/soak_*.py: random long and short regions with widths and dampings over decades,
    every packing, fp64 at 1e-9 and fp32 at 1e-3 against the oracle; the resident step loop and the device MAP search
    against their launch-per-step forms bit for bit on random contexts) over a few seeds on every GPU test run; the
    fp32 long-region soak is what found the W4 region II clamp in round 3."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script)] + args, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "soak ok" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]

