#!/usr/bin/env python3
"""How robust is the headline figure?    The headline ensemble is SURVEY 8d's: every
walker within 1e-3 of the truth, so all walkers share one near / far classification per tile.  This tool times the
same kernel on other ensemble states and line widths and says why the rate moves:

    python tools/robustness.py [--steps 10] [--dtype f64]      (GPU box)  -> a table on stdout

per case: walker-steps/s, ms per half-step launch, and -- computed on the host from the case's own walkers with the
kernel's classification rule (a line is FAR from a 256-pixel tile when its centre lies >= 4 tile half-widths beyond
the tile's edge and the tile is outside its |z|^2 < 64 zone) -- the share of stretch proposals that fall inside the priors (the rest are
rejected before the sweep and cost almost nothing), the mean number of near (line, tile) pairs per
walker (of 1024), how many of those lie in a line's table zone, and the share of tiles with more than 8 far lines
(a second node pass)."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import FWHM_PER_SIGMA, LINE, C_LIGHT, PIX_HZ, SIGMA0, SQRT_LN2, make_workload  # noqa: E402


def classify(wl, n_walkers=64):
    """near pairs per walker etc. from the (N, b, z) walkers of a workload (inverse maps of physics.py:15,27,120,134)"""
    P, K = wl["P"], wl["K"]
    th = wl["theta0"][:n_walkers].reshape(-1, K, 3)
    l_fixed, line, x_origin, x_scale = [float(v) for v in wl["nbz"][0]]
    sig = th[:, :, 1] * 1e3 * np.sqrt(2.0) / (2.355 * (line * 1e-10))
    c = (C_LIGHT / (line * (1.0 + th[:, :, 2]) * 1e-10) - x_origin) / x_scale
    G = sig / x_scale * FWHM_PER_SIGMA
    s = 2.0 * SQRT_LN2 / G
    y = l_fixed * SQRT_LN2 / G
    w8 = np.sqrt(np.maximum(64.0 - y * y, 0.0)) / s
    x = wl["x"]
    nt = P // 256
    lo, hi = x[0::256][:nt], x[255::256][:nt]
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    dist = np.abs(mid[None, None, :] - c[:, :, None]) - half[None, None, :]            # [W, K, tiles]
    # the library's rules (vamp_hip.hip: VAMP_FF_DIST 2, VAMP_WIDE_MAX 3/4, VAMP_MID_Z2 30.25): far = centre >= 2 half-widths beyond
    # the edge and the tile outside |z| < 8; at the nodes through the table = wide (half a tile <= 3/4 z) or that far away and
    # outside |z|^2 < 30.25; near = the rest, evaluated at every pixel
    wmid = np.sqrt(np.maximum(30.25 - y * y, 0.0)) / s
    away = dist >= 2.0 * half[None, None, :]
    far = away & (dist >= w8[:, :, None])
    wide = (s * half.max() <= 0.75)[:, :, None] & ~far
    midz = away & (dist >= wmid[:, :, None]) & ~far
    near = ~(far | wide | midz)
    in_zone = near & (dist + 2 * half[None, None, :] <= w8[:, :, None])                 # the whole tile inside |z| < 8
    nfar = far.sum(axis=1)                                                              # [W, tiles]
    n_nodes = (wide | midz).sum(axis=(1, 2)).mean()
    # share of stretch proposals that land inside the priors (the others are rejected before the sweep: a walker
    # outside its prior costs the staging only): random pairs of this ensemble, z = ((a - 1) u + 1)^2 / a, a = 2
    rng = np.random.default_rng(5)
    full = wl["theta0"]
    i, j = rng.integers(0, full.shape[0], 4096), rng.integers(0, full.shape[0], 4096)
    zz = (rng.random(4096) + 1.0) ** 2 / 2.0
    q = (full[j] - (full[j] - full[i]) * zz[:, None]).reshape(-1, K, 3)
    qsig = q[:, :, 1] * 1e3 * np.sqrt(2.0) / (2.355 * (line * 1e-10))
    qa = q[:, :, 0] * SIGMA0 / (qsig * np.sqrt(2.0 * np.pi))
    qc = (C_LIGHT / (line * (1.0 + q[:, :, 2]) * 1e-10) - x_origin) / x_scale
    qG = qsig / x_scale * FWHM_PER_SIGMA
    fw = (x[-1] - x[0]) / 2.0 * FWHM_PER_SIGMA
    ok = ((qa >= 0) & (qc >= x[0]) & (qc <= x[-1]) & (qG >= 0) & (qG <= fw)).all(axis=1)
    return {"proposals_inside_prior": float(ok.mean()),
            "near_pairs_per_walker": float(near.sum(axis=(1, 2)).mean()),
            "near_pairs_wholly_in_table_zone": float(in_zone.sum(axis=(1, 2)).mean()),
            "table_pairs_at_the_nodes": float(n_nodes),
            "tiles_with_more_than_8_far_lines": float((nfar > 8).mean()),
            "nfar_hist": np.bincount(nfar.ravel(), minlength=K + 1).tolist()}


def valu_busy(args, outdir):
    """SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) of k_half_step: one rocprofv3 --pmc pass of a
    2-step run of the case (counters only; the program directly after `--`)"""
    import csv
    import glob
    os.makedirs(outdir, exist_ok=True)
    cmd = ["rocprofv3", "--pmc", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "--output-format", "csv", "-d", outdir, "--",
           "python3", os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-chain",
           "--sustain-seconds", "0"] + args
    subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
    acc = {}
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_half_step" in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    if not {"SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"} <= set(acc):
        return None, None
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    return m["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0), m["SQ_INSTS_VALU"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--pmc-dir", default="", help="also collect VALU counters of every case (one rocprofv3 --pmc pass each) under this folder")
    a = ap.parse_args()
    cases = [("truth", 1.0), ("truth", 0.25), ("truth", 4.0), ("dispersed", 1.0), ("dispersed", 0.25), ("dispersed", 4.0), ("prior", 1.0)]
    rows = []
    for ens, ws in cases:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", "2", "--no-cpu-baseline",
                              "--sustain-seconds", "0", "--dtype", a.dtype, "--ensemble", ens, "--width-scale", str(ws)],
                             capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:                                  # noqa: BLE001
            print(ens, ws, "failed", out.stderr[-500:], flush=True)
            continue
        wl = make_workload(W=256, ensemble=ens, width_scale=ws)
        cl = classify(wl)
        busy = instr = None
        if a.pmc_dir:
            busy, instr = valu_busy(["--dtype", a.dtype, "--ensemble", ens, "--width-scale", str(ws)], os.path.join(a.pmc_dir, f"{ens}_{ws:g}"))
        rows.append((ens, ws, j["value"], j["roofline"]["avg_launch_ms"], j["acceptance_fraction"], j["finite_lnprob_fraction"], cl))
        print(f"{ens:10s} widths x{ws:<5g} {j['value'] / 1e6:8.2f} M walker-steps/s  {j['roofline']['avg_launch_ms']:7.3f} ms/half-step  "
              f"accept {j['acceptance_fraction']:.3f}  proposals inside the prior {cl['proposals_inside_prior']:.2f} "
              f"({j['roofline']['avg_launch_ms'] / max(cl['proposals_inside_prior'], 1e-9):6.3f} ms per half-step of swept proposals)  near pairs {cl['near_pairs_per_walker']:6.1f} (+ {cl['table_pairs_at_the_nodes']:5.1f} at the nodes through the tables) "
              f"(in table zone {cl['near_pairs_wholly_in_table_zone']:6.1f})  tiles with > 8 far lines {cl['tiles_with_more_than_8_far_lines']:.2f}  "
              f"nfar hist {cl['nfar_hist']}"
              + ("" if busy is None else f"  VALU busy {busy:.2f}  VALU wave-instructions per launch {instr:.3e}"), flush=True)
    if rows:
        base = rows[0][2]
        print("relative to the headline:", ", ".join(f"{e}/x{w:g}: {v / base:.2f}" for e, w, v, *_ in rows))


if __name__ == "__main__":
    main()
