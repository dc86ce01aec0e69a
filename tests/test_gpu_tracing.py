"""SURVEY section 5 "Tracing": the library's roctx ranges, seen by `rocprofv3 --marker-trace`.
(Marker + kernel trace only -- never combined with counter collection.)"""
import csv
import glob
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

SCRIPT = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import torch                      # first: one ROCm runtime in the process (INTEGRATION.md)
import vamp_amd
x = np.arange(64, dtype=np.float64) - 31.5
flux = np.exp(-1.2 * np.exp(-0.5 * (x / 4.0) ** 2))
ctx = vamp_amd.HipContext(device=0)
ctx.set_regions(x, flux, np.full(64, 0.02), 1, mode=vamp_amd.MODE_GAUSS3)
rng = np.random.default_rng(1)
X0 = np.stack([rng.uniform(0.8, 1.6, 32), rng.uniform(-2, 2, 32), rng.uniform(3, 5, 32)], 1)
ctx.sampler_init(X0, seed=1)
ctx.set_option("resident", 0)     # one launch per half-step: a range per half-step
ctx.run(5)
ctx.set_option("resident", 1)     # a 32-walker region is taken by the device-resident loop: one range, one launch
ctx.run(7)
best, lnp, chi, its = ctx.map_all([X0[0]], iterlim=20)
ctx.close()
print("done")
"""


@pytest.mark.gpu
def test_roctx_ranges_show_in_a_marker_trace(tmp_path):
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not installed")
    script = tmp_path / "traced.py"
    script.write_text(SCRIPT % ROOT)
    out = tmp_path / "trace"
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("VAMP_ROCTX", None)
    rc = subprocess.run(["rocprofv3", "--marker-trace", "--kernel-trace", "--output-format", "csv", "-d", str(out), "--",
                         sys.executable, str(script)], capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
    assert rc.returncode == 0 and "done" in rc.stdout, rc.stderr[-2000:]
    files = glob.glob(str(out / "**" / "*marker_api_trace.csv"), recursive=True)
    assert files, "no marker trace written"
    names = [row["Function"] for f in files for row in csv.DictReader(open(f))]
    assert names.count("vamp_sampler_init") == 1 and names.count("vamp_sampler_run") == 2
    assert names.count("vamp resident step loop") == 1
    assert names.count("vamp half-step 0 (red moves)") == 5 and names.count("vamp half-step 1 (blue moves)") == 5
    assert names.count("vamp_map_all") == 1
    kernels = [row["Kernel_Name"] for f in glob.glob(str(out / "**" / "*kernel_trace.csv"), recursive=True) for row in csv.DictReader(open(f))]
    assert sum("k_half_step" in k for k in kernels) == 10 and sum("k_run_resident" in k for k in kernels) == 1
    assert sum("k_map_search" in k for k in kernels) == 1          # the whole MAP search: one launch
