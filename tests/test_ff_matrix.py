"""The far-field transform of the tile kernels (vamp_amd/csrc/ff_matrix.inc, tools/gen_ff_matrix.py):
node values at a tile's 16 Chebyshev nodes -> four local power series (one per quarter of the tile).
CPU checks of the committed constants: they reproduce the degree-15 interpolant, they reproduce
far-line optical depths to the accuracy DESIGN.md states, and the generator regenerates the file."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "vamp_amd", "csrc", "ff_matrix.inc")
N, DEG, ROWS = 16, 13, 56


def _load():
    txt = open(INC).read()
    assert f"FF_DEG = {DEG}" in txt and f"FF_ROWS = {ROWS}" in txt
    body = txt[txt.index("{") + 1:txt.rindex("}")]
    vals = np.array([float(v) for v in re.findall(r"[-+]?\d\.\d+e[-+]\d+", body)])
    assert vals.size == ROWS * N + N
    m = vals[:ROWS * N].reshape(N // 2, ROWS, 2)          # [n/2][lane][n%2]
    M = np.empty((ROWS, N))
    M[:, 0::2] = m[:, :, 0].T
    M[:, 1::2] = m[:, :, 1].T
    return M, vals[ROWS * N:]


def _series_eval(a, tt):
    """the kernel's evaluation: quarter by pixel index (64 pixels each), Horner in u = 4 (t - t0)"""
    out = np.empty_like(tt)
    for i, t in enumerate(tt):
        q = i // 64
        u = (t - (-0.75 + 0.5 * q)) * 4.0
        acc = a[q * (DEG + 1) + DEG]
        for j in range(DEG - 1, -1, -1):
            acc = acc * u + a[q * (DEG + 1) + j]
        out[i] = acc
    return out


def test_nodes_and_constant_row_sums():
    M, nodes = _load()
    assert np.allclose(nodes, np.cos(np.pi * (np.arange(N) + 0.5) / N), rtol=0, atol=1e-15)
    # a constant maps to a_0 = const and nothing else, in every quarter
    s = M.sum(axis=1).reshape(4, DEG + 1)
    assert np.allclose(s[:, 0], 1.0, atol=1e-14) and np.abs(s[:, 1:]).max() < 2e-14


def test_reproduces_polynomials_of_degree_13_exactly():
    """for a polynomial of degree <= 13 the local series ARE its re-expansion: nothing is truncated"""
    M, nodes = _load()
    rng = np.random.default_rng(3)
    tt = -1.0 + 2.0 * np.arange(256) / 255.0
    for _ in range(5):
        c = rng.standard_normal(DEG + 1) / (1.0 + np.arange(DEG + 1)) ** 2
        f = np.polynomial.chebyshev.chebval(nodes, c)
        got = _series_eval(M @ f, tt)
        want = np.polynomial.chebyshev.chebval(tt, c)
        assert np.abs(got - want).max() < 2e-14 * np.abs(want).max()


@pytest.mark.parametrize("dist,yy,tol", [(2.0, 1e-6, 4e-11), (2.0, 0.5, 3e-12), (3.0, 1e-6, 4e-13), (4.0, 1e-6, 4e-14), (4.0, 0.5, 4e-14),
                                         (6.0, 1e-3, 4e-14), (20.0, 1e-3, 4e-14), (200.0, 0.3, 4e-14)])
def test_far_line_profile_accuracy(dist, yy, tol):
    """a Lorentzian wing whose centre lies `dist` half-widths beyond the tile's edge (the far-field
    criterion is dist >= FF_DIST = 2 since round 4, 4 before): series vs the exact profile at the 256 pixels of a
    uniform tile.  At dist = 2: 3e-11 of the wing's own value (the terms u^14, u^15 the series drop, 9^-14; the
    degree-15 interpolant itself 6e-13); from dist = 4: interpolation 4e-15 + the rounding of the transform (row sums
    of |M| up to 280: <= 3e-14)."""
    M, nodes = _load()
    c = 1.0 + dist
    f = lambda t: 1.0 / ((t - c) ** 2 + yy ** 2)
    tt = -1.0 + 2.0 * np.arange(256) / 255.0
    got = _series_eval(M @ f(nodes), tt)
    assert (np.abs(got - f(tt)) / f(tt)).max() < tol


def test_descending_and_stretched_grids_stay_inside_the_series_range():
    """|u| <= 1 on uniform grids; a grid whose spacing drifts by 10 % across the tile keeps |u| < 1.2,
    where the dropped terms (u^14, u^15) are still < 1e-16 of the value"""
    M, nodes = _load()
    x = np.cumsum(1.0 + 0.1 * np.arange(256) / 255.0)
    mid, half = 0.5 * (x[0] + x[-1]), 0.5 * (x[-1] - x[0])
    tt = (x - mid) / half
    c = 1.0 + 4.0
    f = lambda t: 1.0 / ((t - c) ** 2 + 1e-4)
    u = (tt - (-0.75 + 0.5 * (np.arange(256) // 64))) * 4.0
    assert np.abs(u).max() < 1.2
    got = _series_eval(M @ f(nodes), tt)
    assert (np.abs(got - f(tt)) / f(tt)).max() < 4e-14


def test_generator_reproduces_committed_file(tmp_path):
    pytest.importorskip("mpmath")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import importlib
    g = importlib.import_module("gen_ff_matrix")
    M, nodes = g.build()
    Mc, nc = _load()
    got = np.array([[float(M[i, j]) for j in range(N)] for i in range(ROWS)])
    assert np.array_equal(got, Mc) and np.array_equal(np.array([float(v) for v in nodes]), nc)


# ---- fp32 contexts: the 8-node transform (vamp_amd/csrc/ff_matrix32.inc) ----------------------------
INC32 = os.path.join(ROOT, "vamp_amd", "csrc", "ff_matrix32.inc")
N32, DEG32, ROWS32 = 8, 7, 32


def _load32():
    txt = open(INC32).read()
    assert f"FF32_DEG = {DEG32}" in txt and f"FF32_ROWS = {ROWS32}" in txt and f"FF32_NODES = {N32}" in txt
    body = txt[txt.index("{") + 1:txt.rindex("}")]
    vals = np.array([np.float32(v) for v in re.findall(r"([-+]?\d\.\d+e[-+]\d+)f", body)], dtype=np.float32)
    assert vals.size == ROWS32 * N32 + N32
    m = vals[:ROWS32 * N32].reshape(N32 // 4, ROWS32, 4)            # [n/4][lane][n%4]
    M = np.empty((ROWS32, N32), dtype=np.float32)
    for n4 in range(N32 // 4):
        M[:, 4 * n4:4 * n4 + 4] = m[n4]
    return M, vals[ROWS32 * N32:]


def _series_eval32(a, tt):
    """the fp32 kernel's evaluation, in float32 arithmetic: quarter by pixel index, Horner in u = 4 (t - t0)"""
    out = np.empty(tt.size, dtype=np.float32)
    for i, t in enumerate(tt.astype(np.float32)):
        q = i // 64
        u = np.float32((t - np.float32(-0.75 + 0.5 * q)) * np.float32(4.0))
        acc = a[q * (DEG32 + 1) + DEG32]
        for j in range(DEG32 - 1, -1, -1):
            acc = np.float32(acc * u + a[q * (DEG32 + 1) + j])
        out[i] = acc
    return out


def test_f32_matrix_nodes_conditioning_and_constants():
    M, nodes = _load32()
    assert np.allclose(nodes, np.cos(np.pi * (np.arange(N32) + 0.5) / N32), rtol=0, atol=1e-7)
    assert np.abs(M.astype(np.float64)).sum(axis=1).max() < 3.5          # (the 16-node matrix: 280) -- safe in fp32
    s = M.astype(np.float64).sum(axis=1).reshape(4, DEG32 + 1)
    assert np.allclose(s[:, 0], 1.0, atol=3e-7) and np.abs(s[:, 1:]).max() < 1e-6


@pytest.mark.parametrize("dist,yy,tol", [(2.0, 1e-6, 3e-5), (2.0, 0.5, 3e-5), (4.0, 1e-6, 1e-6), (4.0, 0.5, 1e-6), (6.0, 1e-3, 1e-6),
                                         (20.0, 1e-3, 1e-6), (200.0, 0.3, 1e-6)])
def test_f32_far_line_profile_accuracy(dist, yy, tol):
    """the 8-node series in float32 arithmetic against the exact Lorentzian wing of a line `dist` half-widths
    beyond the tile's edge: 2e-5 of the wing's own value at the far-field criterion (dist = FF_DIST = 2), <= 1e-6
    from dist = 4 (interpolation 3e-7 plus single-precision rounding) -- against W4's own 1e-4"""
    M, nodes = _load32()
    c = 1.0 + dist
    f = lambda t: 1.0 / ((t - c) ** 2 + yy ** 2)
    tt = -1.0 + 2.0 * np.arange(256) / 255.0
    a = (M @ f(nodes.astype(np.float64)).astype(np.float32)).astype(np.float32)
    got = _series_eval32(a, tt)
    assert (np.abs(got - f(tt)) / f(tt)).max() < tol


def test_generator_reproduces_committed_f32_file():
    pytest.importorskip("mpmath")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import importlib
    g = importlib.import_module("gen_ff_matrix")
    M, nodes = g.build(N32, DEG32)
    Mc, nc = _load32()
    got = np.array([[np.float32(float("%.9e" % float(M[i, j]))) for j in range(N32)] for i in range(ROWS32)], dtype=np.float32)
    assert np.array_equal(got, Mc)
    assert np.array_equal(np.array([np.float32(float("%.9e" % float(v))) for v in nodes], dtype=np.float32), nc)
