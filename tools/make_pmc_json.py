#!/usr/bin/env python3
"""profiles/pmc_traffic*.json from a tools/pmc.sh summary: the counters bench.py / tools/bench_c3.py quote in their
roofline objects (HBM traffic and VALU instructions per half-step).

    python tools/make_pmc_json.py profiles/r03_x_headline_pmc_summary.txt > profiles/pmc_traffic.json
    python tools/make_pmc_json.py profiles/r03_x_headline_f32_pmc_summary.txt --dtype f32 > profiles/pmc_traffic_f32.json
    python tools/make_pmc_json.py profiles/r03_x_config3_pmc_summary.txt --workload q1422 --walkers 16384 > profiles/pmc_traffic_c3.json

A half-step of the headline is ONE launch; a half-step of the q1422 batch is one launch per launch class plus their
k_draws launches: the per-kernel means are weighted by launches per half-step (n of the kernel / n of a k_half_step)."""
import argparse
import json

ap = argparse.ArgumentParser()
ap.add_argument("summary")
ap.add_argument("--dtype", default="f64")
ap.add_argument("--workload", default="headline")
ap.add_argument("--walkers", type=int, default=65536)
a = ap.parse_args()

kernels = {}
name = ""
for ln in open(a.summary):
    if ln.startswith("# kernel:"):
        name = ln.split(":", 1)[1].strip()
        kernels[name] = {}
        continue
    parts = ln.split()
    if len(parts) >= 3 and parts[-1].startswith("mean=") and name:
        kernels[name][parts[0]] = (float(parts[-1][5:]), int(parts[-2]) if parts[-2].isdigit() else int(parts[-2].split("=")[-1]))
# launches per half-step of a kernel = its sample count of a counter / the sample count of the SAME counter for a
# k_half_step kernel (the passes of different counter groups may see different numbers of launches)
ref = next(v for k, v in sorted(kernels.items()) if "k_half_step" in k)
tot = {}
for k, v in kernels.items():
    for cname, (mean, n) in v.items():
        tot[cname] = tot.get(cname, 0.0) + mean * n / ref[cname][1]
cfg = {"workload": a.workload, "walkers": a.walkers, "n_gpus": 1, "dtype": a.dtype}
if a.workload == "headline":
    cfg.update(pixels=16384, components=16, ndim=48)
out = {
    "source": f"{a.summary} (rocprofv3 --pmc, one pass per counter group, tools/pmc.sh; kernels: {', '.join(sorted(kernels))})",
    "config": cfg,
    "per": "half-step (headline: one k_half_step launch; q1422: one launch per class + k_draws)",
    "FETCH_SIZE_KB_per_launch": tot["FETCH_SIZE"],
    "WRITE_SIZE_KB_per_launch": tot["WRITE_SIZE"],
    "SQ_INSTS_VALU_per_launch": tot["SQ_INSTS_VALU"],
    "SQ_INSTS_SALU_per_launch": tot.get("SQ_INSTS_SALU"),
    "SQ_INSTS_LDS_per_launch": tot.get("SQ_INSTS_LDS"),
    "GRBM_GUI_ACTIVE_per_launch": tot.get("GRBM_GUI_ACTIVE"),
    "SQ_ACTIVE_INST_VALU_per_launch": tot.get("SQ_ACTIVE_INST_VALU"),
    # share of SIMD cycles with a VALU instruction in flight: the counter ticks once per 4 cycles and SIMD,
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs (fp64 kernels; an fp32 instruction holds the SIMD for 2 cycles only)
    "valu_busy_frac_pmc": (tot["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * tot["GRBM_GUI_ACTIVE"] / 8.0))
                          if "SQ_ACTIVE_INST_VALU" in tot and "GRBM_GUI_ACTIVE" in tot and a.dtype == "f64" else None,
    "note": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a coalesced stream (MI355X_MICROARCH.md, HBM section): "
            "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 B.  SQ_INSTS_VALU counts wave instructions.",
}
print(json.dumps(out, indent=1))
