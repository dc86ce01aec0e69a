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
a = ap.parse_args()
from vamp_amd.vpspectrum import VPspectrum
q = np.load(os.path.join(ROOT, "tests", "golden", "q1422_spectrum.npz"))
sp = VPspectrum(1215.67, voigt=a.voigt, convergence_attempts=a.attempts, nwalkers=a.walkers, iterations=a.iterations, thin=5, burn=a.burn, seed=1, verbose=True)
wl, fl, no = q["wavelength_milli"] / 1000.0, q["flux_micro"] / 1e6, q["noise_micro"] / 1e6
if a.max_regions:
    end = int(q["region_pixels"][a.max_regions - 1][1]) + 50
    wl, fl, no = wl[:end], fl[:end], no[:end]
sp.set_arrays(wl, fl, no)
t0 = time.perf_counter()
params = sp.fit_spectrum(batched=True)
dt = time.perf_counter() - t0
chi = np.array([r.best_chi_squared for r in sp.regions])
n = np.array([r.n for r in sp.regions])
print(json.dumps({"regions": len(sp.regions), "lines": int(n.sum()), "seconds": dt, "median_reduced_chi2": float(np.median(chi)),
                  "frac_regions_chi2_below_1.5": float(np.mean(chi < 1.5)), "n_hist": np.bincount(n).tolist()}))
