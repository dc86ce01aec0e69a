"""Build libvamp_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "vamp_hip.hip")
OUT = os.path.join(HERE, "libvamp_hip.so")
# every file vamp_hip.hip #includes: a regenerated matrix or an edited header must trigger a rebuild
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("ff_matrix.inc", "ff_matrix32.inc", "voigt_math.hpp", "map_search.hpp", "host_plan.hpp")
                if os.path.exists(os.path.join(HERE, "csrc", f))] + [os.path.join(HERE, "..", "include", "vamp_hip.h")]


# -fno-slp-vectorize: left to itself the SLP vectoriser pairs the independent fp32 chains of a lane's four pixels
# into v_pk_fma_f32 / v_pk_mul_f32 and pays for the register pairing with v_mov (429 packed + 464 moves in the fp32
# headline kernel).  On gfx950 a packed fp32 instruction issues no faster than its two halves, so this is pure
# overhead: 2.699 -> 2.220 ms per half-step on the fp32 headline, fp64 unchanged (3.251 -> 3.239)
# (profiles/r03_b_f32_variants.txt; MI355X_MICROARCH.md lists packed fp32 VALU as an anti-lever).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC"]


def build(force=False, verbose=True):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
