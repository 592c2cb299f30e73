#!/usr/bin/env python3
"""Exit-path probe: runs a small piece of the library under a profiler and leaves /proc/self/maps behind, so that the raw
program counters of a fault inside exit() can be resolved to (library, offset) afterwards (tools/exit_probe.sh).

    python3 tools/exit_probe.py VARIANT OUTDIR
VARIANT: torch (no libconp_hip at all) | load (dlopen only) | create (handle created and destroyed) |
         update (il_onelayer: setup + 5 updates, handle destroyed) | update_noclose (same, handle left to the interpreter) |
         host (host-buffer hooks too: pinned staging, re-neighbour rows on the device)
"""
import atexit
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-user-conp2_amd"))
variant, outdir = sys.argv[1], sys.argv[2]
os.makedirs(outdir, exist_ok=True)


def dump_maps():
    tagged = variant + os.environ.get("PROBE_TAG", "")
    with open("/proc/self/maps") as f, open(os.path.join(outdir, f"maps_{tagged}.txt"), "w") as o:
        o.write(f.read())


atexit.register(dump_maps)          # python-level atexit: runs before the C library's exit handlers

import torch  # noqa: E402

assert torch.cuda.is_available()
torch.cuda.init()
x = torch.ones(1024, device="cuda")
print("torch sum", float(x.sum()))
if variant == "torch":
    sys.exit(0)

from conp_amd import FixConp, capi, neighbor, systems  # noqa: E402

capi.load_library()
if variant == "load":
    sys.exit(0)
s = systems.deck("il_onelayer", "ffield")
fx = FixConp(s, device=0)
if variant == "create":
    fx.close()
    sys.exit(0)
import numpy as np  # noqa: E402

if variant in ("project", "invert", "invert_single"):
    rng = np.random.default_rng(1)
    n = 256
    a = rng.standard_normal((n, n)) + n * np.eye(n)
    if variant == "project":
        fx.inv_project(a)                      # plain kernels of our code object, no cooperative launch
    else:
        fx.invert(a)                           # blocked Gauss-Jordan: cooperative panel unless CONP_PANEL_SINGLE is set
    fx.close()
    print("probe done", variant)
    sys.exit(0)
at, alist, blist = neighbor.build_lists(s)
fx.init_lists(alist, blist)
fx.setup_post_neighbor(at)
if variant in ("setup", "setup_rows_host"):
    fx.close()
    print("probe done", variant)
    sys.exit(0)
fx.linalg_setup(at)
if variant == "linalg":
    fx.close()
    print("probe done", variant)
    sys.exit(0)
d_x = torch.from_numpy(at.x.copy()).cuda()
d_q = torch.from_numpy(at.q.copy()).cuda()
for _ in range(5):
    fx.pre_force_device(d_x.data_ptr(), d_q.data_ptr(), s.potdiff)
torch.cuda.synchronize()
if variant == "host":
    for k in range(3):
        fx.pre_force(at, k, s.potdiff)
    fx.post_neighbor(at)
    fx.pre_force(at, 3, s.potdiff)
print("charge sum", float(d_q.sum()))
if variant != "update_noclose":
    fx.close()
print("probe done", variant)
