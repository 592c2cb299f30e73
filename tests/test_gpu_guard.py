"""-m gpu: no kernel stores outside its buffers (guard zones).

Round 3 had ONE abort on the GPU box that left no message behind (gpurun_out/r3_t2.log: pytest's fd-level capture swallowed what
the HIP runtime wrote; the counting of the progress dots puts it in the il_twolayer / CG parameter of test_il_decks_match_oracle,
not in il_onelayer).  What can be done about a fault that does not recur is to make the class of cause visible: with CONP_GUARD=1
every device buffer of the library sits between two 4-KB zones of a known byte pattern, and conp_debug_check_guards() reads them
all back.  This test drives every deck shape of the reference through setup, updates, a re-neighbour, the post-force correction,
the CG solver and the mesh path in a child process with guard zones on, and wants every zone intact."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, {pkg!r}); sys.path.insert(0, {root!r})
from conp_amd import FixConp, neighbor, systems, capi
lib = capi.load_library()
lib.conp_debug_check_guards.restype = int
assert lib.conp_debug_check_guards() == 0, "guard zones are off"

def check(where):
    bad = lib.conp_debug_check_guards()
    assert bad == 0, (where, bad, lib.conp_last_error().decode())

cases = [("dilute", "ffield", (), "conp", None), ("il_onelayer", "ffield", (), "conp", None), ("il_onelayer", "slab", (), "conp", None),
         ("il_twolayer", "ffield", ("cg",), "conp", None),            # the shape round 3's abort happened at (Ne = 1664, CG, etypes)
         ("il_twolayer", "slab", (), "conq", None), ("cond2", "ffield", (), "conp", None),        # rough electrodes: general projection
         ("il_onelayer", "ffield", ("pppm",), "conp", (40, 45, 180, 5))]
# a mid-size synthetic box whose planar vectors fill several bands of FIVE row fragments (straddling the plan's row tiles), projecting
# mode + the structure-factor getter's two-slot partial tiles, symmetric solve with Ne not a multiple of its tile
cases.append(("synthetic", "ffield", (), "conp", None))
for deck, mode, extra, style, mesh in cases:
    if deck == "synthetic":
        s = systems.synthetic_fast(n_cells_x=22, n_cells_y=13, lz=120.0, n_elyte=2048, cutoff=10.0, accuracy_relative=1e-5, g_ewald=0.30)
    else:
        s = systems.deck(deck, mode, etypes=(deck != "dilute"), shuffle_seed=3)
    at, alist, blist = neighbor.build_lists(s)
    kw = dict(pppm_mesh=mesh[:3], pppm_order=mesh[3]) if mesh else dict()
    fx = FixConp(s, extra_args=list(extra), style=style, **kw)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    arg = 0.35 if style == "conq" else s.potdiff
    fx.setup_pre_force(at, 0, arg)
    check((deck, mode, extra, "setup"))
    fx.post_force_step(at, 0)
    fx.post_neighbor(at)                      # a re-neighbour with the same lists: every per-neighbour buffer is rebuilt
    fx.pre_force(at, 1, arg)
    fx.sfac(); fx.matrix() if "cg" not in extra else None; fx.ele_trig()
    check((deck, mode, extra, "update"))
    assert np.isfinite(at.q).all()
    fx.close()
print("GUARD_OK", len(cases))
'''


def test_no_kernel_stores_outside_its_buffers(tmp_path):
    script = tmp_path / "guard_child.py"
    script.write_text(CHILD.format(pkg=os.path.join(ROOT, "lammps-user-conp2_amd"), root=ROOT))
    env = dict(os.environ, CONP_GUARD="1")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "GUARD_OK" in p.stdout
