#!/usr/bin/env python3
"""End-to-end demo of BASELINE.json config 3 as a user would run it: the q1422 quasar spectrum
(tests/golden/q1422_spectrum.npz = the reference's vamp_1.0/data/q1422.cont), every detected region
fitted with the batched BIC ladder.   python tools/fit_q1422.py [--iterations 600] [--max-regions N]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--iterations", type=int, default=600)
ap.add_argument("--burn", type=int, default=200)
ap.add_argument("--walkers", type=int, default=32)
ap.add_argument("--max-regions", type=int, default=0)
ap.add_argument("--voigt", action="store_true")
ap.add_argument("--attempts", type=int, default=4)
ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
ap.add_argument("--profile", default=None, help="write a cProfile of fit_spectrum (sorted by total time) to this file")
ap.add_argument("--dump", default=None, help="save per-region n and best reduced chi^2 to this .npz (to compare two runs)")
ap.add_argument("--quiet", action="store_true")
a = ap.parse_args()
from vamp_amd.vpspectrum import VPspectrum
q = np.load(os.path.join(ROOT, "tests", "golden", "q1422_spectrum.npz"))
sp = VPspectrum(1215.67, voigt=a.voigt, convergence_attempts=a.attempts, nwalkers=a.walkers, iterations=a.iterations, thin=5, burn=a.burn, seed=1, verbose=not a.quiet,
                dtype=a.dtype)
wl, fl, no = q["wavelength_milli"] / 1000.0, q["flux_micro"] / 1e6, q["noise_micro"] / 1e6
if a.max_regions:
    end = int(q["region_pixels"][a.max_regions - 1][1]) + 50
    wl, fl, no = wl[:end], fl[:end], no[:end]
sp.set_arrays(wl, fl, no)
prof = None
if a.profile:
    import cProfile
    prof = cProfile.Profile()
    prof.enable()
t0 = time.perf_counter()
params = sp.fit_spectrum(batched=True)
dt = time.perf_counter() - t0
if prof is not None:
    import io
    import pstats
    prof.disable()
    buf = io.StringIO()
    pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(35)
    open(a.profile, "w").write(buf.getvalue())
chi = np.array([r.best_chi_squared for r in sp.regions])
n = np.array([r.n for r in sp.regions])
if a.dump:
    np.savez(a.dump, n=n, chi=chi)
out = {"dtype": a.dtype, "regions": len(sp.regions), "lines": int(n.sum()), "seconds": dt, "median_reduced_chi2": float(np.median(chi)),
       "frac_regions_chi2_below_1.5": float(np.mean(chi < 1.5)), "n_hist": np.bincount(n).tolist(),
       "difficult_fit": bool(sp.flux_model["difficult_fit"])}


def compare_with_vpm(params):
    """The one answer the reference holds for this spectrum: vamp_1.0/data/q1422.vpm, an AutoVP-style list of
    539 H I lines (tests/golden/q1422_vpm.npz: N [1e12 cm^-2], b [km/s], observed wavelength [A]).  Another
    code's fit of the same data: a sanity yardstick, not parity.  Compared inside the list's own wavelength
    range; a line is matched when a fitted centre lies within +-0.3 A (one-to-one, nearest first)."""
    v = np.load(os.path.join(ROOT, "tests", "golden", "q1422_vpm.npz"))
    lo, hi = float(v["wavelength"].min()) - 0.3, float(v["wavelength"].max()) + 0.3
    mine = np.asarray(params["centers"], dtype=float)
    sel = (mine >= lo) & (mine <= hi)
    mc, mN, mb = mine[sel], np.asarray(params["N"], dtype=float)[sel], np.asarray(params["b"], dtype=float)[sel]
    order = np.argsort(mc)
    mc, mN, mb = mc[order], mN[order], mb[order]
    used = np.zeros(mc.size, dtype=bool)
    pairs = []
    for i in np.argsort(v["wavelength"]):
        d = np.abs(mc - v["wavelength"][i])
        d[used] = np.inf
        j = int(np.argmin(d)) if d.size else -1
        if j >= 0 and d[j] <= 0.3:
            used[j] = True
            pairs.append((i, j))
    pi = np.array([p[0] for p in pairs], dtype=int)
    pj = np.array([p[1] for p in pairs], dtype=int)
    strong = v["N12"] >= 10.0                      # N >= 1e13: lines any fitter should see
    res = {"vpm_lines": int(v["wavelength"].size), "fitted_lines_in_vpm_range": int(mc.size),
           "lines_ratio_fitted_over_vpm": float(mc.size / v["wavelength"].size),
           "vpm_lines_matched_within_0.3A": float(len(pairs) / v["wavelength"].size),
           "vpm_lines_N_ge_1e13_matched": float(np.isin(np.nonzero(strong)[0], pi).mean()),
           "fitted_lines_matched": float(used.mean()) if used.size else 0.0,
           "median_log10_N_vpm": float(np.median(np.log10(v["N12"] * 1e12))),
           "median_log10_N_fitted": float(np.median(np.log10(np.maximum(mN, 1.0)))),
           "median_b_vpm_kms": float(np.median(v["b"])), "median_b_fitted_kms": float(np.median(mb))}
    if len(pairs):
        res["matched_median_dlog10N"] = float(np.median(np.log10(np.maximum(mN[pj], 1.0)) - np.log10(v["N12"][pi] * 1e12)))
        res["matched_median_b_ratio"] = float(np.median(mb[pj] / v["b"][pi]))
    return res


try:
    out["vs_q1422_vpm"] = compare_with_vpm(params)
except Exception as e:                                # noqa: BLE001  (a yardstick must not hide the fit's own numbers)
    out["vs_q1422_vpm"] = {"error": repr(e)}
print(json.dumps(out))
