"""Build libvamp_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "vamp_hip.hip")
OUT = os.path.join(HERE, "libvamp_hip.so")
DEPS = [SRC, os.path.join(HERE, "csrc", "ff_matrix.inc"), os.path.join(HERE, "csrc", "voigt_math.hpp"), os.path.join(HERE, "csrc", "map_search.hpp"), os.path.join(HERE, "..", "include", "vamp_hip.h")]


def build(force=False, verbose=True):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
