#!/usr/bin/env python3
"""Where a chunk of sk_gemm spends its cycles: runs the headline workload on the DIAGNOSTIC build of the library
(`make -C lammps-user-conp2_amd/csrc stamp`: s_memtime stamps around load issue / MFMA / panel build / barrier, summed per wave)
and prints the shares per wave role.  The stamped kernel is slower than the real one (its fences forbid overlaps): shares only.

    python3 tools/sk_stamp.py [workload] [updates]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-user-conp2_amd"))
sys.path.insert(0, ROOT)
os.environ["CONP_LIB"] = os.path.join(ROOT, "lammps-user-conp2_amd", "conp_amd", "libconp_hip_stamp.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.cuda.init()
import bench  # noqa: E402
from conp_amd import FixConp, capi, neighbor  # noqa: E402

s = bench.make_workload(sys.argv[1] if len(sys.argv) > 1 else "headline")
nupd = int(sys.argv[2]) if len(sys.argv) > 2 else 5
at, alist, blist = neighbor.build_lists(s)
fx = FixConp(s)
fx.init_lists(alist, blist); fx.setup_post_neighbor(at); fx.linalg_setup(at)
d_x = torch.from_numpy(np.ascontiguousarray(at.x)).cuda(); d_q = torch.from_numpy(at.q.copy()).cuda()
lib = capi.load_library()
buf = np.zeros(1024 * 8 * 8, dtype=np.uint64)
for _ in range(3):
    fx.pre_force_device(d_x.data_ptr(), d_q.data_ptr(), s.potdiff)
torch.cuda.synchronize()
lib.conp_debug_sk_stamps(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), 1)
for _ in range(nupd):
    fx.pre_force_device(d_x.data_ptr(), d_q.data_ptr(), s.potdiff)
torch.cuda.synchronize()
lib.conp_debug_sk_stamps(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), 0)
t = buf.reshape(1024, 8, 8).astype(np.float64)
used = t[:, :, 6].sum(axis=1) > 0
t = t[used]
names = ["prologue", "load issue", "mfma", "build", "barrier", "epilogue"]
print(f"workgroups with work: {t.shape[0]}, updates {nupd}")
for role, sel in (("early waves 0-3 (multiply, then build)", slice(0, 4)), ("late waves 4-7 (build, then multiply)", slice(4, 8))):
    w = t[:, sel, :]
    tot = w[:, :, :6].sum()
    print(f"  {role}: cycles per wave per update {tot / (w.shape[0] * 4 * nupd):.0f}, chunks per wave {w[:, :, 6].sum() / (w.shape[0] * 4 * nupd):.1f}")
    for k, n in enumerate(names):
        print(f"      {n:11s} {100 * w[:, :, k].sum() / tot:5.1f} %   ({w[:, :, k].sum() / max(w[:, :, 6].sum(), 1):8.0f} cycles per chunk)")
per_wave = t[:, :, :6].sum(axis=2) / nupd
print("  per-wave total cycles: min %.0f  median %.0f  max %.0f" % (per_wave.min(), np.median(per_wave), per_wave.max()))
for wv in range(8):
    w = t[:, wv, :]
    print(f"  wave {wv}: mfma {w[:, 2].sum() / w[:, 6].sum():7.0f}  build {w[:, 3].sum() / w[:, 6].sum():6.0f}  barrier {w[:, 4].sum() / w[:, 6].sum():6.0f}  load {w[:, 1].sum() / w[:, 6].sum():5.0f} cycles per chunk")
ck = np.zeros(1024 * 4, dtype=np.uint64)
if hasattr(lib, "conp_debug_sk_clock") and lib.conp_debug_sk_clock(ck.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0:
    ck = ck.reshape(1024, 4).astype(np.float64)[:t.shape[0]]
    dc, dr = ck[:, 1] - ck[:, 0], ck[:, 3] - ck[:, 2]
    ok = dr > 0
    print("  shader clock during the (stamped) kernel: median %.0f MHz (s_memtime ticks per 100-MHz s_memrealtime tick); "
          "kernel length per workgroup: median %.1f us" % (np.median(dc[ok] / dr[ok]) * 100.0, np.median(dr[ok]) / 100.0))
fx.close()

# per-segment lengths (the stream-K cost model of conp_fix.cpp build_items is fitted to these): workgroup, first row fragment of
# the band, active column fragments per row fragment (five slots), row fragments of the band, chunks, microseconds, XCD
fx2 = None
sg = np.zeros(4096 * 4, dtype=np.uint64)
if hasattr(lib, "conp_debug_sk_segs") and lib.conp_debug_sk_segs(sg.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0:
    sg = sg.reshape(4096, 4)
    sg = sg[(sg[:, 2] & np.uint64(0xffffffff)) > 0]
    out = os.path.join(ROOT, "gpurun_out", "sk_segments.txt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        f.write("# wg g0 nbf0 nbf1 nbf2 nbf3 nbf4 rf chunks us xcc\n")
        for r in sg:
            nbf = int(r[1])
            f.write("%d %d %d %d %d %d %d %d %d %.2f %d\n" % (int(r[0]) >> 32, int(r[0]) & 0xffff, nbf & 255, (nbf >> 8) & 255, (nbf >> 16) & 255,
                                                            (nbf >> 24) & 255, (nbf >> 32) & 255, int(r[2]) >> 32, int(r[2]) & 0xffffffff,
                                                            float(r[3]) / 100.0, (int(r[0]) >> 16) & 15))
    print(f"  {len(sg)} segments -> {out}")
