"""Property tests (hypothesis) of the Voigt arithmetic that the kernels inline, on its host build:
agreement with scipy.special.wofz, evenness in x, positivity, monotone decay in |x|, the
Lorentzian and Gaussian limits, continuity across the branch boundaries."""
import ctypes as C
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st
from scipy.special import wofz

from conftest import ROOT


@pytest.fixture(scope="module")
def H():
    so = os.path.join(ROOT, "tests", "host", "libvoigt_host.so")
    if not os.path.exists(so):
        import __graft_entry__ as ge
        ge.build()
    lib = C.CDLL(so)

    def f(x, y):
        x = np.ascontiguousarray(np.atleast_1d(x), dtype=np.float64)
        y = np.ascontiguousarray(np.broadcast_to(np.atleast_1d(y), x.shape), dtype=np.float64)
        out = np.empty_like(x)
        lib.voigt_H_host(C.c_int64(x.size), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
                         out.ctypes.data_as(C.c_void_p))
        return out
    return f


logx = st.floats(min_value=-6, max_value=6)
logy = st.floats(min_value=-14, max_value=4)


@settings(max_examples=400, deadline=None)
@given(logx, logy)
def test_matches_scipy_everywhere(H, lx, ly):
    x, y = 10.0 ** lx, 10.0 ** ly
    w = wofz(complex(x, y)).real
    if w > 1e-300:
        assert abs(H(x, y)[0] - w) <= 1e-13 * w


@settings(max_examples=200, deadline=None)
@given(logx, logy)
def test_even_positive_and_decaying(H, lx, ly):
    x, y = 10.0 ** lx, 10.0 ** ly
    a, b, c = H([x, -x, x * 1.05], y)
    assert a == b                      # even in x
    assert a > 0 or wofz(complex(x, y)).real < 1e-300
    assert c <= a * (1 + 1e-13)         # monotone decay in |x| at fixed y


def test_branch_boundaries_are_continuous(H):
    """Across |z|^2 = 64, 196, 625, 1e4, 1e8 the two neighbouring methods agree to rounding."""
    for r2 in (64.0, 196.0, 625.0, 1.0e4, 1.0e8):
        for y in (1e-12, 1e-6, 1e-3, 0.1, 1.0, 0.9 * np.sqrt(r2)):
            x = np.sqrt(r2 - y * y)
            lo, hi = H([x * (1 - 1e-9), x * (1 + 1e-9)], y)
            ref = wofz(complex(x, y)).real
            assert abs(lo - hi) <= 2e-8 * ref + 1e-13 * ref * 10   # slope term + rounding
            assert abs(lo - ref) <= 1e-13 * ref * 3 + 3e-8 * ref


def test_limits(H):
    # Lorentzian wing: H -> y / (sqrt(pi) (x^2 + y^2)); Gaussian core: H(x, 0+) -> exp(-x^2)
    x = np.array([1e3, 1e5, 1e7])
    assert np.allclose(H(x, 2.0), 2.0 / (np.sqrt(np.pi) * (x * x + 4.0)), rtol=1e-5)
    x = np.linspace(0, 5, 11)
    assert np.allclose(H(x, 1e-200), np.exp(-x * x), rtol=2e-14)
    assert np.isnan(H(np.nan, 1.0)[0])
