"""CPU tests of the C-ABI boundary: the library loads, exports every symbol include/vamp_hip.h
declares (and nothing the header does not), the ctypes table mirrors the header, and the product
refuses to run without a GPU instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _header_functions():
    src = open(os.path.join(ROOT, "include", "vamp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vamp_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def libpath():
    import vamp_amd.build as vb
    return vb.build(verbose=False)


def test_library_exports_header_symbols(libpath):
    lib = C.CDLL(libpath)
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vamp_hip.h but not exported"
    out = subprocess.check_output(["nm", "-D", "--defined-only", libpath], text=True)
    exported = sorted(set(re.findall(r"\bT (vamp_[a-z0-9_]+)\b", out)))
    assert exported == names, "exported vamp_* symbols and the header disagree"


def test_ctypes_table_mirrors_header():
    from vamp_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_functions()


def test_version_and_error_string(libpath):
    from vamp_amd import _lib
    lib = _lib.load()
    assert lib.vamp_version() == 4          # VAMP_ABI_VERSION of include/vamp_hip.h
    assert isinstance(lib.vamp_last_error(), bytes)
    # NULL-argument calls are rejected before any HIP call
    assert lib.vamp_ctx_set_stream(None, None) == -1
    assert b"NULL" in lib.vamp_last_error()
    assert lib.vamp_sampler_half_step(None, 0) == -1
    assert lib.vamp_comm_init_rank(None, None, 0, 1) == -1 and lib.vamp_sampler_pack_get(None, 0, None) == -1


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product must fail loudly (no oracle / CPU path behind it)."""
    import vamp_amd
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(vamp_amd._lib.VampError):
        vamp_amd.HipContext(device=0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "vamp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "libvamp_oracle" not in txt and "vamp_oracle" not in txt, f


def test_default_split_block():
    import vamp_amd
    assert vamp_amd.default_split_block(65536) == 1024
    assert vamp_amd.default_split_block(65536, world=8) == 1024
    assert vamp_amd.default_split_block(100) == 100
    assert vamp_amd.default_split_block(4096, world=8) == 512
    b = vamp_amd.default_split_block(16384, world=6 if False else 4)
    assert 16384 % b == 0 and b % 2 == 0 and (16384 // b) % 4 == 0


def _build_c_client(tmp_path):
    import subprocess
    exe = os.path.join(str(tmp_path), "abi_client")
    libdir = os.path.join(ROOT, "vamp_amd")
    subprocess.check_call(["gcc", "-O1", "-std=c99", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "host", "abi_client.c"),
                           "-L" + libdir, "-lvamp_hip", "-lm", "-Wl,-rpath," + libdir])
    return exe


def test_c_client_compiles_and_links(tmp_path):
    """include/vamp_hip.h is plain C99 and libvamp_hip.so links from gcc without Python or torch."""
    assert os.path.exists(_build_c_client(tmp_path))


@pytest.mark.gpu
def test_c_client_runs(tmp_path):
    """The plain-C client (tests/host/abi_client.c): log-posterior against its own closed form, a
    short sampler run and the error path, through the C ABI only."""
    import subprocess
    out = subprocess.run([_build_c_client(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "abi_client ok" in out.stdout


def test_import_order_hazard_is_diagnosed():
    """INTEGRATION.md "Two ROCm runtimes in one process": loading libvamp_hip.so before torch is
    diagnosed with a RuntimeWarning (once, at load); torch first -- what bench.py and the tests do --
    is silent.  (Loading needs no GPU.)"""
    import subprocess
    code = "import warnings; warnings.simplefilter('always'); %s import vamp_amd; vamp_amd._lib.load(); print('loaded')"
    env = {k: v for k, v in os.environ.items() if k != "VAMP_NO_IMPORT_ORDER_WARNING"}
    bad = subprocess.run([sys.executable, "-c", code % ""], capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert bad.returncode == 0 and "loaded" in bad.stdout and "two ROCm runtimes" in bad.stderr
    good = subprocess.run([sys.executable, "-c", code % "import torch;"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert good.returncode == 0 and "loaded" in good.stdout and "two ROCm runtimes" not in good.stderr
