#!/usr/bin/env python3
"""cost of hipHostRegister / hipHostUnregister on a pageable buffer of the size of LAMMPS' x + q arrays, and of an H2D copy from
it before / after (the host-buffer update's upload): decides whether registering the host's arrays pays"""
import ctypes as C
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
for n in (36864 * 3, 294912 * 3):
    a = np.random.rand(n)
    nb = a.nbytes
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), nb) == 0
    p = a.ctypes.data

    def h2d(k=20):
        ts = []
        for _ in range(k):
            t0 = time.perf_counter()
            assert hip.hipMemcpyAsync(d, p, nb, 1, None) == 0
            t1 = time.perf_counter()
            hip.hipStreamSynchronize(None)
            t2 = time.perf_counter()
            ts.append((t1 - t0, t2 - t0))
        ts = np.array(ts[5:]) * 1e6
        return ts[:, 0].mean(), ts[:, 1].mean()
    e0, t0_ = h2d()
    tr = []
    for _ in range(10):
        t0 = time.perf_counter()
        r = hip.hipHostRegister(p, nb, 0)
        t1 = time.perf_counter()
        assert r == 0, r
        e1, t1_ = h2d(8)
        t2 = time.perf_counter()
        hip.hipHostUnregister(p)
        t3 = time.perf_counter()
        tr.append((t1 - t0, t3 - t2))
    tr = np.array(tr[2:]) * 1e6
    print(f"{nb / 1e6:.2f} MB: pageable H2D enqueue {e0:.1f} us, done {t0_:.1f} us; registered: enqueue {e1:.1f}, done {t1_:.1f}; "
          f"hipHostRegister {tr[:, 0].mean():.0f} us, hipHostUnregister {tr[:, 1].mean():.0f} us")
