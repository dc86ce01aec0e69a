#!/usr/bin/env python3
"""The q1422 fit in the reference's own loop order (one region after the other: VPspectrum.fit_spectrum(batched=False)) against the
batched ladder, on the first N regions of the spectrum.   python tools/fit_sequential_vs_batched.py [N]   (GPU box)"""
import os, sys, time, json
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch  # noqa
from vamp_amd.vpspectrum import VPspectrum
q = np.load(os.path.join(ROOT, "tests", "golden", "q1422_spectrum.npz"))
wl, fl, no = q["wavelength_milli"] / 1000.0, q["flux_micro"] / 1e6, q["noise_micro"] / 1e6
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
end = int(q["region_pixels"][n - 1][1]) + 50
out = {}
for batched in (False, True):
    sp = VPspectrum(1215.67, voigt=False, convergence_attempts=4, nwalkers=32, iterations=600, thin=5, burn=200, seed=1, verbose=False)
    sp.set_arrays(wl[:end], fl[:end], no[:end])
    t0 = time.perf_counter()
    sp.fit_spectrum(batched=batched)
    dt = time.perf_counter() - t0
    chi = np.array([r.best_chi_squared for r in sp.regions])
    out["batched" if batched else "sequential"] = {"regions": len(sp.regions), "seconds": dt, "lines": int(sum(r.n for r in sp.regions)),
                                                    "median_reduced_chi2": float(np.median(chi)), "seconds_per_region": dt / len(sp.regions)}
print(json.dumps(out))
