"""Prototype (numpy) of the z-NUFFT form of the class table (round 5 design study):
   Hc[r][c] = sum_m w(r,m) sum_j A_rj [cos(m th_j) Tc[m][c] + sin(m th_j) Ts[m][c]]     (exact: what sk_gemm + its epilogue form)
            ~ sum_j A_rj sum_g phi(g h - th_j) P[r][g][c]                                    (window of W taps on a grid of N points)
   P[r][g][c] = (1/N) sum_{|m| < nz} (1/phihat(m)) Khat_rc(m) e^{i m g h},  Khat the Fourier coefficients of the trigonometric polynomial
   K_rc(th) = sum_m w(r,m) [cos(m th) Tc[m][c] + sin(m th) Ts[m][c]].
Checks the error of the windowed form against the exact sums on random data of the headline's shape."""
import numpy as np

rng = np.random.default_rng(3)
nz, N, W = 126, 512, 15
beta = 2.30 * W
h = 2 * np.pi / N
Nl, R, nzc = 4096, 8, 2
th = np.sort(rng.uniform(0.2, 6.0, Nl))                 # atoms' z phases, sorted
A = rng.normal(size=(R, Nl))                             # a_pj / b_pj rows
w = rng.uniform(0.1, 1.0, size=(R, nz)) * (rng.uniform(size=(R, nz)) < 0.8)      # weights with a "sphere cut"
thz = rng.uniform(0, 2 * np.pi, nzc)                     # electrode z classes
m = np.arange(nz)
Tc, Ts = np.cos(np.outer(m, thz)), np.sin(np.outer(m, thz))
# exact
C = np.cos(np.outer(m, th)); S = np.sin(np.outer(m, th))             # [nz][Nl]
G_c = A @ C.T; G_s = A @ S.T                                          # [R][nz]
H_exact = (w * G_c) @ Tc + (w * G_s) @ Ts                             # [R][nzc]

def phi(t):                                                           # ES kernel on [-1, 1]
    return np.where(np.abs(t) <= 1, np.exp(beta * (np.sqrt(np.maximum(1 - t * t, 0)) - 1)), 0.0)
xs, ws = np.polynomial.legendre.leggauss(200)
a = W * h / 2
phihat = np.array([np.sum(ws * a * np.exp(beta * (np.sqrt(1 - xs * xs) - 1)) * np.cos(k * a * xs)) for k in range(nz)])   # FT of phi(x / a)

# K_rc(th) = sum_m w [cos(m th) Tc + sin(m th) Ts] = Re sum_m w (Tc - i Ts) e^{i m th}:  type-2 evaluation through the grid:
#   K(th_j) ~ sum_g phi((g h - th_j) / a) * P[g],   P[g] = (h / 1) * Re sum_m (w (Tc - i Ts) / phihat(m)) e^{i m g h} ... normalisation found below
g = np.arange(N)
E = np.exp(1j * np.outer(g * h, m))                                   # [N][nz]
P = np.zeros((R, N, nzc))
for c in range(nzc):
    coef = w * (Tc[:, c] - 1j * Ts[:, c])[None, :] / phihat[None, :]  # [R][nz]
    P[:, :, c] = h * np.real(coef @ E.T)                              # [R][N]
# windowed evaluation, chunked like the kernel: 16 sorted atoms per chunk, window origin from the chunk's first atom
H_app = np.zeros((R, nzc))
maxspan = 0
for s0 in range(0, Nl, 16):
    t = th[s0:s0 + 16]
    i0 = np.ceil(t / h - W / 2).astype(int)                            # first tap of every atom
    g0 = i0.min()
    span = i0.max() - g0 + W
    maxspan = max(maxspan, span)
    cols = g0 + np.arange(48)
    Phi = phi((cols[None, :] * h - t[:, None]) / a)                    # [16][48], zero outside every atom's W taps
    acc = A[:, s0:s0 + 16] @ Phi                                       # [R][48]   (the MFMA part)
    H_app += np.einsum('rg,rgc->rc', acc, P[:, cols % N, :])           # epilogue: project the window on P
print('max footprint of a chunk:', maxspan, 'grid points')
print('H: max abs err / max |H| = %.3e' % (np.abs(H_app - H_exact).max() / np.abs(H_exact).max()))
