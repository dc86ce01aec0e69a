#!/usr/bin/env python3
"""Developer probe (not a test): log-posterior error of the far field on WELL-FITTED data (|lnprob| ~ P/2) for lines
of every damping -- the far field's relative error applies to a line's wing, so heavily damped, strong lines are
its worst case -- for one or more builds of the library (e.g. different -DVAMP_FF_DIST).
usage (GPU box): python tests/ff_probe.py lib.so [lib.so ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vamp_amd                                   # noqa: E402
from oracle import vamp_oracle as vo              # noqa: E402
from vamp_amd import _lib                         # noqa: E402

P, K, W = 16384, 16, 8
x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
cases = {}
for name, (g_lo, g_hi, l_lo, l_hi, a_hi) in {"headline-like": (20, 200, -3, 0, 3.0), "damped": (5, 120, 0, 2.0, 10.0),
                                            "narrow strong": (2, 40, -1, 1.5, 50.0), "mixed": (1, 400, -4, 2.5, 20.0), "damped wings": (5, 30, 1.5, 2.5, -2.5)}.items():
    for sd in (0.05, 0.005):
        rng = np.random.default_rng(11)
        t = np.empty((K, 4))
        t[:, 1] = rng.uniform(x[0], x[-1], K)
        t[:, 3] = rng.uniform(g_lo, g_hi, K)
        t[:, 2] = 10.0 ** rng.uniform(l_lo, l_hi, K)
        t[:, 0] = rng.uniform(0.3, a_hi, K) if a_hi > 0 else 10.0 ** rng.uniform(0.0, -a_hi, K)     # (< 0: log-uniform up to 10^-a_hi:
                                                                                                    #  optical depth ~ 1 hundreds of px out)
        truth = t.reshape(-1)
        noise = np.full(P, sd)
        r0 = vo.Region(x=x, flux=np.ones(P), noise=noise, n_comp=K, mode=vo.MODE_VOIGT4)
        flux = vo.model_flux(r0, truth) + rng.normal(0, sd, P)
        r = vo.Region(x=x, flux=flux, noise=noise, n_comp=K, mode=vo.MODE_VOIGT4)
        th = truth[None, :] * (1.0 + 1e-4 * rng.standard_normal((W, 4 * K)))
        cases[(name, sd)] = (flux, noise, th, vo.log_prob_batch_fast(r, th))
for path in sys.argv[1:]:
    ctx = vamp_amd.HipContext(device=0, lib=_lib.bind(os.path.abspath(path)))
    for (name, sd), (flux, noise, th, want) in cases.items():
        ctx.set_regions(x, flux, noise, K, mode=vamp_amd.MODE_VOIGT4)
        got = ctx.lnprob(th)
        aerr = np.abs(got - want)
        print(f"{os.path.basename(path):18s} {name:14s} sd {sd:5.3f}: |lnprob| ~ {np.abs(want).mean():.3e}  abs err {aerr.max():.2e}  rel {np.max(aerr / np.maximum(1, np.abs(want))):.2e}", flush=True)
    ctx.close()
