"""CPU check of the z-window form's mathematics (lammps-user-conp2_amd/csrc/conp_zn.hip, conp_fix.cpp zn_order_list /
zn_ensure_tables): the class table of km_ewald.cpp:728-825,
    Hc[r][c] = sum_m w(r,m) sum_j A_rj [cos(m th_j) Tc[m][c] + sin(m th_j) Ts[m][c]],
evaluated through the window  phi(t) = exp(beta (sqrt(1 - t^2) - 1))  with the library's parameters (W = 15 taps, grid = the multiple of 16
at or above 3.8 nz, beta = 0.97 pi W (1 - nz / n), window transform by 64-point Gauss-Legendre quadrature), reproduces the exact sums to
1e-12 of the largest entry -- the budget the GPU parity tests (1e-11 against the full contraction) leave it.  The parameters are read
from the library's source so that a change there is seen here."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "lammps-user-conp2_amd", "csrc", "conp_fix.cpp")


def library_parameters():
    s = open(SRC).read()
    W = int(re.search(r"ZN_W\s*=\s*(\d+)", s).group(1))
    m = re.search(r"int n = std::max\(64, \((\d+) \* plan\.nz / (\d+) \+ 15\) / 16 \* 16\)", s)
    num, den = int(m.group(1)), int(m.group(2))
    g = float(re.search(r"zn_beta = ([0-9.]+) \* 3\.14159265358979323846 \* ZN_W", s).group(1))
    return W, num, den, g


def window_error(nz, rng, nl=2048, rows=6, nzc=3):
    W, num, den, gam = library_parameters()
    n = max(64, (num * nz // den + 15) // 16 * 16)
    beta = gam * np.pi * W * (1.0 - nz / n)
    h = 2 * np.pi / n
    a = 0.5 * W * h
    th = np.sort(rng.uniform(0.0, 2 * np.pi, nl))
    A = rng.normal(size=(rows, nl))
    w = rng.uniform(0.1, 1.0, size=(rows, nz)) * (rng.uniform(size=(rows, nz)) < 0.8)      # weights with a sphere cut
    thz = rng.uniform(0, 2 * np.pi, nzc)
    m = np.arange(nz)
    Tc, Ts = np.cos(np.outer(m, thz)), np.sin(np.outer(m, thz))
    C, S = np.cos(np.outer(m, th)), np.sin(np.outer(m, th))
    exact = (w * (A @ C.T)) @ Tc + (w * (A @ S.T)) @ Ts
    xs, ws = np.polynomial.legendre.leggauss(64)                                          # zn_ensure_tables: 64 points
    phihat = np.array([np.sum(ws * a * np.exp(beta * (np.sqrt(1 - xs * xs) - 1)) * np.cos(k * a * xs)) for k in range(nz)])
    g = np.arange(n)
    E = np.exp(1j * np.outer(g * h, m))
    P = np.empty((rows, nzc, n))
    for c in range(nzc):
        P[:, c, :] = h * np.real((w * (Tc[:, c] - 1j * Ts[:, c])[None, :] / phihat[None, :]) @ E.T)
    app = np.zeros((rows, nzc))
    ncol = 32
    for s0 in range(0, nl, 16):                                                            # chunks of 16 atoms, one window origin each
        u = th[s0:s0 + 16] / h
        i0 = np.ceil(u - 0.5 * W).astype(int)
        g0 = i0.min() - 2
        assert i0.max() + W - g0 <= ncol
        cols = g0 + np.arange(ncol)
        d = (cols[None, :] - u[:, None]) * (2.0 / W)
        Phi = np.where(np.abs(d) < 1, np.exp(beta * (np.sqrt(np.maximum(1 - d * d, 0)) - 1)), 0.0)
        acc = A[:, s0:s0 + 16] @ Phi
        app += np.einsum("rg,rcg->rc", acc, P[:, :, cols % n])
    return np.abs(app - exact).max() / np.abs(exact).max(), n, W


@pytest.mark.parametrize("nz", [126, 266, 48])          # headline (ffield), 16384 / 262144, a small plan
def test_window_reproduces_the_trigonometric_sums(nz):
    err, n, W = window_error(nz, np.random.default_rng(nz))
    assert W == 15 and n % 16 == 0 and n >= 3.8 * nz - 16
    assert err < 1e-12, (nz, n, err)


def test_fewer_taps_would_not_do():
    """the tap count is not generous: the same construction with W = 11 misses the budget by two orders of magnitude"""
    import unittest.mock as mock
    with mock.patch(__name__ + ".library_parameters", lambda: (11, 38, 10, 0.97)):
        err, _, _ = window_error(126, np.random.default_rng(1))
    assert err > 1e-11
