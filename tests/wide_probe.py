#!/usr/bin/env python3
"""Developer probe (not a test): the log-posterior error of the tile interpolant for lines at the width where they
start to count as wide (VAMP_WIDE_MAX), on WELL-FITTED data -- |lnprob| ~ P/2, where an absolute error weighs most
against the relative bar of 1e-9 -- for one or more builds of the library.
usage (GPU box): python tests/wide_probe.py WIDE_MAX=lib.so [WIDE_MAX=lib.so ...]"""
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
for spec in sys.argv[1:]:
    wm, _, path = spec.partition("=")
    wide_max = float(wm)
    edge = 128.0 * 2.0 * np.sqrt(np.log(2.0)) / wide_max
    ctx = vamp_amd.HipContext(device=0, lib=_lib.bind(os.path.abspath(path)))
    for case in ("centred", "random", "one"):
        for sd in (0.05, 0.005):
            rng = np.random.default_rng(5)
            t = np.empty((K, 4))
            tiles = rng.permutation(P // 256)[:K]
            t[:, 1] = x[0] + 256.0 * tiles + (127.5 if case != "random" else rng.uniform(0, 256, K))
            t[:, 3] = edge * (1.0 + rng.uniform(0.0, 0.05, K))
            t[:, 2] = 10.0 ** rng.uniform(-3, 0.5, K)
            t[:, 0] = rng.uniform(0.3, 3.0, K)
            if case == "one":
                t[1:, 3] = rng.uniform(20, 100, K - 1)
            truth = t.reshape(-1)
            noise = np.full(P, sd)
            r0 = vo.Region(x=x, flux=np.ones(P), noise=noise, n_comp=K, mode=vo.MODE_VOIGT4)
            flux = vo.model_flux(r0, truth) + rng.normal(0, sd, P)
            r = vo.Region(x=x, flux=flux, noise=noise, n_comp=K, mode=vo.MODE_VOIGT4)
            th = truth[None, :] * (1.0 + 1e-4 * rng.standard_normal((W, 4 * K)))
            th[:, 3::4] = np.maximum(th[:, 3::4], edge * 1.0001) if case != "one" else th[:, 3::4]
            ctx.set_regions(x, flux, noise, K, mode=vamp_amd.MODE_VOIGT4)
            want = vo.log_prob_batch_fast(r, th)
            got = ctx.lnprob(th)
            aerr = np.abs(got - want)
            print(f"WIDE_MAX {wide_max:4.2f} {case:8s} sd {sd:5.3f}: |lnprob| ~ {np.abs(want).mean():.3e}  abs err {aerr.max():.2e}  rel {np.max(aerr / np.maximum(1, np.abs(want))):.2e}", flush=True)
    ctx.close()
