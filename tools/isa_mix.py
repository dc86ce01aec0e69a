#!/usr/bin/env python3
"""Static instruction mix of a kernel in the device ISA listing of the library.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o build/vamp_dev.s vamp_amd/csrc/vamp_hip.hip
    python tools/isa_mix.py build/vamp_dev.s 'k_half_step<true, 0, 2, Pack<64, 16, false, 4, true, true, true>' [--dump out.s]

Classes: fp64 VALU (4 cycles per wave instruction on a SIMD), fp32 VALU (2 cycles), transcendental fp32
(8), conversions, integer, lane moves, SALU, LDS, VMEM.  Static counts: loops are not weighted."""
import collections
import re
import subprocess
import sys


def classify(i):
    if i.startswith("v_"):
        if "f64" in i:
            return "v_trans64" if any(t in i for t in ("rcp", "rsq", "sqrt")) else "v_f64"
        if i.startswith("v_pk"):
            return "v_pk"
        if "f32" in i and any(t in i for t in ("exp", "log", "rcp", "rsq", "sqrt", "sin", "cos")):
            return "v_trans32"
        if "cvt" in i:
            return "v_cvt"
        if "f32" in i:
            return "v_f32"
        if any(t in i for t in ("mul_lo", "mul_hi", "mad_u64", "mad_i64")):
            return "v_int_mul"
        if i.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "v_lane"
        if i.startswith(("v_cmp", "v_cndmask")):
            return "v_cmp_cnd"
        if i.startswith(("v_mov", "v_accvgpr")):
            return "v_mov"
        return "v_other"
    if i.startswith("s_"):
        if i.startswith("s_waitcnt"):
            return "s_waitcnt"
        if i.startswith(("s_cbranch", "s_branch")):
            return "s_branch"
        if i.startswith("s_load"):
            return "s_load"
        return "salu"
    if i.startswith("ds_"):
        return "ds"
    if i.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem_" + "_".join(i.split("_")[:2])
    return "other"


def main():
    src = open(sys.argv[1]).read()
    pat = sys.argv[2]
    parts = re.split(r"\n\t\.globl\t", src)
    names = [p.split("\n", 1)[0].strip() for p in parts[1:]]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for p, d in zip(parts[1:], dem):
        d = d.replace("(anonymous namespace)::", "")
        if pat not in d:
            continue
        body = p.split("s_endpgm")[0]
        ins = []
        for ln in body.splitlines():
            t = ln.strip()
            if not ln.startswith("\t") or not t or t.startswith((".", ";")):
                continue
            ins.append(t.split()[0])
        c = collections.Counter(classify(i) for i in ins)
        print(d[:140])
        print("  instructions:", len(ins))
        for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
            print(f"  {k:18s} {v}")
        if "--dump" in sys.argv:
            open(sys.argv[sys.argv.index("--dump") + 1], "w").write(body)
        return
    print("no kernel matches", pat)


if __name__ == "__main__":
    main()
