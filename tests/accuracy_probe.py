#!/usr/bin/env python3
"""Developer probe (not a test): how far the device log-posterior (the chi^2 sweep with far field and
Taylor tables) of the bench shape (P = 16 384, K = 16, NBZ3) lies from the numpy + scipy.wofz oracle, for one or more builds
of the library.  usage (GPU box): python tests/accuracy_probe.py [lib.so ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vamp_amd                                   # noqa: E402
from bench import make_workload                   # noqa: E402
from oracle import vamp_oracle as vo              # noqa: E402
from vamp_amd import _lib                         # noqa: E402

libs = sys.argv[1:] or [_lib.LIB_PATH]
wl = make_workload(W=16)
region = vo.Region(x=wl["x"], flux=wl["flux"], noise=wl["noise"], n_comp=wl["K"], mode=vo.MODE_NBZ3)
region.l_fixed, region.line, region.x_origin, region.x_scale = [float(v) for v in wl["nbz"][0]]
want = vo.log_prob_batch_fast(region, wl["theta0"])
for path in libs:
    ctx = vamp_amd.HipContext(device=0, lib=_lib.bind(os.path.abspath(path)))
    ctx.set_regions(wl["x"], wl["flux"], wl["noise"], wl["K"], mode=vamp_amd.MODE_NBZ3, nbz=wl["nbz"])
    got = ctx.lnprob(wl["theta0"])
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    print(f"{os.path.basename(path):28s} lnprob: max rel err {err.max():.3e}  mean {err.mean():.3e}  (|lnprob| ~ {np.abs(want).mean():.3e})", flush=True)
    ctx.close()
