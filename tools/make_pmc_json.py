#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a tools/pmc.sh summary of `python bench.py` (the counters bench.py
quotes in its roofline object: HBM traffic and VALU instructions per half-step launch).

    python tools/make_pmc_json.py profiles/r02_x_pmc_summary.txt > profiles/pmc_traffic.json
"""
import json
import sys

vals = {}
kernel = ""
for ln in open(sys.argv[1]):
    if ln.startswith("# kernel:"):
        kernel = ln.split(":", 1)[1].strip()
        continue
    parts = ln.split()
    if len(parts) >= 3 and parts[-1].startswith("mean="):
        vals[parts[0]] = float(parts[-1][5:])
out = {
    "source": f"{sys.argv[1]} (rocprofv3 --pmc, one pass per counter group, tools/pmc.sh; kernel {kernel or 'k_half_step'})",
    "config": {"pixels": 16384, "components": 16, "walkers": 65536, "ndim": 48, "n_gpus": 1, "dtype": "f64"},
    "FETCH_SIZE_KB_per_launch": vals["FETCH_SIZE"],
    "WRITE_SIZE_KB_per_launch": vals["WRITE_SIZE"],
    "SQ_INSTS_VALU_per_launch": vals["SQ_INSTS_VALU"],
    "SQ_INSTS_SALU_per_launch": vals.get("SQ_INSTS_SALU"),
    "SQ_INSTS_LDS_per_launch": vals.get("SQ_INSTS_LDS"),
    "GRBM_GUI_ACTIVE_per_launch": vals.get("GRBM_GUI_ACTIVE"),
    "SQ_ACTIVE_INST_VALU_per_launch": vals.get("SQ_ACTIVE_INST_VALU"),
    # share of SIMD cycles with a VALU instruction in flight: the counter ticks once per 4 cycles and SIMD,
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    "valu_busy_frac_pmc": (vals["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * vals["GRBM_GUI_ACTIVE"] / 8.0))
                          if "SQ_ACTIVE_INST_VALU" in vals and "GRBM_GUI_ACTIVE" in vals else None,
    "note": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a coalesced stream (MI355X_MICROARCH.md, HBM section): "
            "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 B.  SQ_INSTS_VALU counts wave instructions.",
}
print(json.dumps(out, indent=1))
