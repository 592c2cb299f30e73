#!/usr/bin/env python3
"""Per-workgroup timeline of zn_gemm_kernel (diagnostic build: hipcc -DZN_TIMELINE on conp_zn.hip, loaded through CONP_LIB).
Prints, for the last update, the distribution of prologue / main loop / epilogue durations and the CU occupancy picture."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lammps-user-conp2_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from conp_amd import FixConp, neighbor, capi

wl = sys.argv[1] if len(sys.argv) > 1 else "headline"
s = bench.make_workload(wl)
at, alist, blist = neighbor.build_lists(s)
fx = FixConp(s, device=0)
fx.init_lists(alist, blist); fx.setup_post_neighbor(at); fx.linalg_setup(at)
d_x = torch.from_numpy(np.ascontiguousarray(at.x)).cuda(); d_q = torch.from_numpy(at.q.copy()).cuda()
for _ in range(40):
    fx.pre_force_device(d_x.data_ptr(), d_q.data_ptr(), s.potdiff)
torch.cuda.synchronize()
info = fx.info()
lib = ctypes.CDLL(os.environ["CONP_LIB"])
nb = 8192
buf = (ctypes.c_ulonglong * (6 * nb))()
rc = lib.conp_debug_zn_timeline(buf, nb)
a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 6)
a = a[a[:, 0] > 0]
t = a[:, :4].astype(np.int64)
base = t[:, 0].min()
t = (t - base) * 0.01     # us
print("rc", rc, "workgroups", len(a), "zn_cols", info.zn_cols, "grid", info.zn_grid)
for k, nm in enumerate(["start", "after prologue", "after main loop", "end"]):
    print("%-16s min %7.2f  median %7.2f  max %7.2f us" % (nm, t[:, k].min(), np.median(t[:, k]), t[:, k].max()))
print("prologue  median %.2f  max %.2f" % (np.median(t[:, 1] - t[:, 0]), (t[:, 1] - t[:, 0]).max()))
print("main loop median %.2f  max %.2f  min %.2f" % (np.median(t[:, 2] - t[:, 1]), (t[:, 2] - t[:, 1]).max(), (t[:, 2] - t[:, 1]).min()))
print("epilogue  median %.2f  max %.2f" % (np.median(t[:, 3] - t[:, 2]), (t[:, 3] - t[:, 2]).max()))
hw = a[:, 4] & 0xffffffff; xcc = a[:, 4] >> 32
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
key = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
u, cnt = np.unique(key, return_counts=True)
print("distinct CUs", len(u), "workgroups per CU: min %d max %d" % (cnt.min(), cnt.max()))
b = np.arange(len(a))
same = sum(1 for i in range(min(256, len(a))) if i + 256 < len(a) and key[i] == key[i + 256])
print("blocks b and b+256 on the same CU: %d of 256" % same)
print("block -> xcc of the first 16:", xcc[:16].tolist())
# main loop duration vs chunk count
cyc = (a[:, 5] >> 16).astype(np.float64); a[:, 5] &= 0xffff
print("chunks per item: min %d max %d" % (a[:, 5].min(), a[:, 5].max()))
print("core clock over the workgroups' lifetimes (clock64 / wall clock): median %.0f MHz, min %.0f, max %.0f" % tuple(f(cyc / (t[:, 3] - t[:, 0])) for f in (np.median, np.min, np.max)))
# sorted end times: how ragged the finish is
e = np.sort(t[:, 3]); print("end-time percentiles 10/50/90/100: %.1f %.1f %.1f %.1f" % tuple(np.percentile(e, [10, 50, 90, 100])))
st = np.sort(t[:, 0]); print("start-time percentiles 10/50/90/100: %.1f %.1f %.1f %.1f" % tuple(np.percentile(st, [10, 50, 90, 100])))
# per age slot on the CU (blocks b, b + 256, ... share a CU; the lower block is the older workgroup)
nb_ = len(a)
for sl in range((nb_ + 255) // 256):
    m = t[sl * 256:(sl + 1) * 256]
    print("slot %d: prologue end %6.2f  main loop end %6.2f  end %6.2f   (medians; main loop %5.2f us = %.0f ns / chunk)" %
          (sl, np.median(m[:, 1]), np.median(m[:, 2]), np.median(m[:, 3]), np.median(m[:, 2] - m[:, 1]),
           1e3 * np.median(m[:, 2] - m[:, 1]) / np.median(a[sl * 256:(sl + 1) * 256, 5].astype(float))))
# one CU in detail
k0 = key[0]
for i in np.where(key == k0)[0]:
    print("  CU of block 0: block %4d  start %6.2f  prologue end %6.2f  main end %6.2f  end %6.2f" % (i, t[i, 0], t[i, 1], t[i, 2], t[i, 3]))
buf2 = (ctypes.c_ulonglong * (16 * 64))()
if hasattr(lib, "conp_debug_zn_timeline_chunks") and lib.conp_debug_zn_timeline_chunks(buf2) == 0:
    c = np.frombuffer(buf2, dtype=np.uint64).reshape(16, 64).astype(np.int64)
    for sl in range(16):
        v = c[sl][c[sl] > 0]
        if len(v) == 0:
            continue
        tt = (v - base) * 0.01
        print("  slot %d chunk end times:" % sl, " ".join("%.1f" % x for x in tt))
