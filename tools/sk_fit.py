#!/usr/bin/env python3
"""Least-squares fit of the stream-K cost model of sk_gemm (conp_fix.cpp build_items) to per-segment lengths measured by the stamp
build (tools/sk_stamp.py -> gpurun_out/sk_segments.txt):   us = a * chunks * (mean kz blocks + C0) + a * CSEG  per segment,
kz blocks = active 8-kz column fragments / 2.  Prints a, C0, CSEG (the constants SK_C0 / SK_CSEG) and the residual spread per XCD."""
import sys

import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sk_segments.txt"
d = np.loadtxt(path)
wg, g0, rf, chunks, us, xcc = d[:, 0], d[:, 1], d[:, 7], d[:, 8], d[:, 9], d[:, 10]
nb = d[:, 2:7].sum(axis=1) / 8.0           # the cost model's unit: 0.125 x the band's active column fragments
# us = a*chunks*nb + (a*C0_4)*chunks*[rf = 4] + (a*C0_5)*chunks*[rf = 5] + (a*CSEG)
X = np.stack([chunks * nb, chunks * (rf == 4), chunks * (rf == 5), np.ones_like(chunks)], 1)
coef, *_ = np.linalg.lstsq(X, us, rcond=None)
a, ac4, ac5, acseg = coef
res = us - X @ coef
print(f"segments {len(us)}: a = {a:.4f} us per chunk per unit, C0 (bands of 4) = {ac4 / a:.3f}, C0 (bands of 5) = {ac5 / a:.3f}, "
      f"CSEG = {acseg / a:.2f} (in chunk units); rms residual {res.std():.2f} us of mean {us.mean():.1f} us")
for band in sorted(set(zip(g0, rf))):
    m = (g0 == band[0]) & (rf == band[1])
    print(f"  band g0 {int(band[0]):3d} rf {int(band[1])}: {m.sum():3d} segments, units {nb[m].mean():5.2f}, us per chunk {np.mean(us[m] / chunks[m]):6.3f}, "
          f"mean residual {res[m].mean():+.2f} us")
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  XCD {x}: {m.sum():3d} segments, mean residual {res[m].mean():+.2f} us ({100 * res[m].mean() / us[m].mean():+.2f} %)")
# per-workgroup totals: the slowest workgroup sets the kernel's length
tot = {}
for w, u in zip(wg, us):
    tot[w] = tot.get(w, 0.0) + u
t = np.array(list(tot.values()))
print(f"  per-workgroup totals: min {t.min():.1f}  median {np.median(t):.1f}  max {t.max():.1f} us  (max / median = {t.max() / np.median(t):.3f})")
