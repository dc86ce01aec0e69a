"""Sanitizer runs of the HOST side (CPU only; GPU AddressSanitizer is not available on this pool).

`make -C oracle asan` builds oracle/_asan/libvamp_cpu.so (the C ABI of include/vamp_hip.h on the
host -- it compiles the product's csrc/voigt_math.hpp and csrc/map_search.hpp) and
oracle/_asan/libvamp_oracle.so with -fsanitize=address,undefined.  The boundary tests' error paths,
pack / scatter arithmetic, sampler and MAP search then run against the instrumented library, and
the plain-C client of the ABI (tests/host/abi_client.c) is built and run instrumented as well.
Any heap/stack overflow, use-after-free or undefined behaviour fails the run."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

ASAN_DIR = os.path.join(ROOT, "oracle", "_asan")


def _runtime(name):
    p = subprocess.check_output(["gcc", "-print-file-name=" + name], text=True).strip()
    if not os.path.isabs(p):
        pytest.skip(name + " not installed")
    return p


@pytest.fixture(scope="module")
def asan_env():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ)
    env.update(LD_PRELOAD=_runtime("libasan.so") + ":" + _runtime("libubsan.so"),
               ASAN_OPTIONS="detect_leaks=0:exitcode=97", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=98",
               VAMP_CPU_SO=os.path.join(ASAN_DIR, "libvamp_cpu.so"), OMP_NUM_THREADS="4")
    return env


def test_boundary_tests_under_asan_ubsan(asan_env):
    """error codes and call order, pack / scatter, the single-rank exchange rehearsal, Philox
    trajectories, injected draws, smallest shapes, the MAP search, regions of 17 .. 32 lines (the stack arrays
    sized by VAMP_MAX_COMPONENTS), the launch / shard plan arithmetic shared with the product
    (csrc/host_plan.hpp): tests/test_cpu_boundary.py against the instrumented library"""
    sel = ("error_codes or pack_and_scatter or single_rank or philox or injected_draws or smallest_shapes "
           "or map_all or exports_the_whole_header or lnprob_matches_golden or regions_of_more_than_16_lines or batched_find_bic "
           "or host_plan")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_cpu_boundary.py"), "-x", "-q",
                          "-p", "no:cacheprovider", "-k", sel], env=asan_env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "passed" in out.stdout and "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr


def test_c_client_under_asan_ubsan(asan_env, tmp_path):
    """tests/host/abi_client.c (plain C99, the ABI only) built with the sanitizers against the
    instrumented host library: log-posterior vs its own closed form, a sampler run, the error path"""
    exe = str(tmp_path / "abi_client_asan")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g",
                           "-o", exe, os.path.join(ROOT, "tests", "host", "abi_client.c"), "-L" + ASAN_DIR, "-lvamp_cpu", "-lm",
                           "-Wl,-rpath," + ASAN_DIR])
    env = dict(asan_env)
    env.pop("LD_PRELOAD")                      # the executable carries the runtimes itself
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "abi_client ok" in out.stdout


def test_c_oracle_under_asan_ubsan(asan_env):
    """the plain-C restatement (oracle/vamp_oracle.c) instrumented: its own tests in tests/test_oracle.py"""
    env = dict(asan_env, VAMP_ORACLE_SO=os.path.join(ASAN_DIR, "libvamp_oracle.so"))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-x", "-q",
                          "-p", "no:cacheprovider", "-k", "c_oracle"], env=env, capture_output=True,
                         text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0 and "3 passed" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
