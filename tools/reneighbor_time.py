#!/usr/bin/env python3
"""Cost of a re-neighbour (conp_fix_post_neighbor: index bookkeeping, electrode-row regrouping of the half list, uploads,
stream-K schedule) -- LAMMPS calls it every ~10-20 steps, so it is amortised over that many updates.
usage: python tools/reneighbor_time.py [workload]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-user-conp2_amd"))
sys.path.insert(0, ROOT)


def page_locked_lists(fx, lists):
    """copies of the neighbour lists whose arrays live in page-locked memory (conp_host_alloc)"""
    import ctypes as C
    import copy
    import numpy as np
    keep, out, done = [], [], {}
    for l in lists:
        if id(l) in done:
            out.append(done[id(l)]); continue
        c = copy.copy(l)
        for name in ("ilist", "numneigh", "first", "neigh"):
            a = np.ascontiguousarray(getattr(l, name), dtype=np.int32)
            nbytes = max(a.nbytes, 4)
            ptr = fx.lib.conp_host_alloc(nbytes)
            assert ptr, "conp_host_alloc failed"
            buf = (C.c_char * nbytes).from_address(ptr)
            v = np.frombuffer(buf, dtype=np.int32, count=a.size)
            v[:] = a
            keep.append((ptr, buf))
            setattr(c, name, v)
        done[id(l)] = c
        out.append(c)
    return out, keep


def main():
    import bench
    from conp_amd import FixConp, neighbor
    wl = sys.argv[1] if len(sys.argv) > 1 else "headline"
    s = bench.make_workload(wl)
    t0 = time.perf_counter()
    at, alist, blist = neighbor.build_lists(s)
    t1 = time.perf_counter()
    fx = FixConp(s)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    n = 10
    t2 = time.perf_counter()
    for _ in range(n):
        fx.init_lists(alist, blist)
        fx.post_neighbor(at)
    t3 = time.perf_counter()
    # the same with the flattened list in page-locked memory, as the LAMMPS glue keeps it (lammps_glue/fix_conp_hip.cpp PinnedInts)
    locked, keep = page_locked_lists(fx, (alist, blist))
    for _ in range(3):
        fx.init_lists(*locked)
        fx.post_neighbor(at)
    t3a = time.perf_counter()
    for _ in range(n):
        fx.init_lists(*locked)
        fx.post_neighbor(at)
    t3b = time.perf_counter()
    print(f"{wl}: post_neighbor with the list page-locked (the glue's PinnedInts) {(t3b - t3a) / n * 1e3:.3f} ms")
    fx.init_lists(alist, blist)
    fx.post_neighbor(at)
    ren_pageable = (t3 - t2) / n
    t3 = time.perf_counter()
    for k in range(n):
        fx.pre_force(at, k + 1, s.potdiff)
    t4 = time.perf_counter()
    fx.pin_host_arrays(at)
    fx.pre_force(at, n + 1, s.potdiff)
    tp = time.perf_counter()
    for k in range(n):
        fx.pre_force(at, k + 1, s.potdiff)
    print(f"{wl}: pre_force with the host arrays page-locked in place {(time.perf_counter() - tp) / n * 1e3:.3f} ms")
    fx.unpin_host_arrays()
    t4b = time.perf_counter()
    for k in range(n):
        fx.post_force(at)
    t5 = time.perf_counter()
    for k in range(n):
        fx.post_force_step(at, n)            # the step of the last pre_force: x, q are resident
    t6 = time.perf_counter()
    print(f"{wl}: post_force (host buffers) {(t5 - t4b) / n * 1e3:.3f} ms, same step as pre_force {(t6 - t5) / n * 1e3:.3f} ms")
    print(f"{wl}: nall {at.nall}, blist pairs {blist.npairs}: post_neighbor {ren_pageable * 1e3:.3f} ms, "
          f"pre_force (host buffers) {(t4 - t3) / n * 1e3:.3f} ms   [python list build: {t1 - t0:.1f} s, not part of the library]")
    fx.close()


if __name__ == "__main__":
    main()
