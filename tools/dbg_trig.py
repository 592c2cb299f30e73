import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-user-conp2_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_py
from conp_amd import FixConp, neighbor, systems
from helpers import OracleRun
lib = oracle_py.load()
for deck, mode in (("dilute", "ffield"), ("dilute", "slab"), ("il_onelayer", "ffield")):
    s = systems.deck(deck, mode, etypes=(deck != "dilute"))
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s); fx.init_lists(alist, blist); fx.setup_post_neighbor(at); fx.a_cal(at)
    o = OracleRun(lib, s, at, alist, blist); o.fx.lib.orc_fix_a_cal(o.fx.h)
    cg, sg = fx.ele_trig(); co, so = o.fx.trig()
    d = fx.info().kcount_dims
    bad = np.argwhere((cg != co) | (sg != so))
    print(deck, mode, "dims", list(d), "kflat", cg.shape, "mismatches", len(bad))
    if len(bad):
        cols = sorted(set(int(b[1]) for b in bad))
        print("  columns:", cols[:40], "...", len(cols))
        for i, f in bad[:6]:
            print("   atom", i, "flat", f, "c", cg[i, f], co[i, f], cg[i, f] - co[i, f], "s", sg[i, f], so[i, f], sg[i, f] - so[i, f])
    fx.close(); o.fx.close()
