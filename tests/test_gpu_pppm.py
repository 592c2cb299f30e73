"""-m gpu: PPPM k-space b vector (`pppm` keyword, SURVEY row a20 / BASELINE configs[3]) against the oracle's restatement of
pppm_conp.cpp's b_cal chain, and both against the Ewald b within the PPPM accuracy (the only cross-check available:
LAMMPS' PPPM internals are not part of the reference repository -> parity unpinned at that boundary)."""
import numpy as np
import pytest

import oracle_py
from conp_amd import FixConp, neighbor, systems
from helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("deck,mode,mesh,order,acc", [
    ("dilute", "ffield", (27, 24, 144), 5, 1e-4),       # the mesh LAMMPS chose in tests/dilute/persist.log:113-114
    ("dilute", "slab", (27, 24, 432), 5, 1e-4),
    ("il_onelayer", "ffield", (40, 45, 180), 5, 1e-4),  # BASELINE configs[3]
    ("il_onelayer", "ffield", (36, 40, 150), 4, 1e-3),  # even order: the other rounding convention (shift/shiftone)
    ("dilute", "ffield", (28, 22, 128), 5, 1e-3),       # 7 and 11 are not radix-2/3/5: x, y lines take the plain-DFT fallback
    ("dilute", "ffield", (32, 25, 160), 7, 1e-4),       # radix 4/2/5 only; highest stencil order LAMMPS allows
])
def test_pppm_b_matches_oracle_and_ewald(oracle, deck, mode, mesh, order, acc):
    s = systems.deck(deck, mode, etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s, extra_args=["pppm"], pppm_mesh=mesh, pppm_order=order)
    assert fx.args.pppm == 1
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    b_pppm = fx.km_b_cal(at)                                  # k-space part only (+ slab), eleall order
    m = fx.maps()
    loc = {int(t): i for i, t in enumerate(at.tag[:at.nlocal])}
    xele = np.array([at.x[loc[int(t)]] for t in m["eleall2tag"]])
    pp = oracle_py.Pppm(oracle, s, mesh, order)
    b_o = pp.b_cal(at.x, at.q, at.echeck, at.nlocal, xele)
    assert rel_err(b_pppm, b_o) < 1e-11                        # same mesh arithmetic (atomic-add order aside)
    ks = oracle_py.KSpace.from_system(oracle, s)
    sr, si = ks.sincos_b(at.x, at.q, at.echeck, at.nlocal)
    csk, snk = ks.ele_trig(xele)
    b_ew = ks.bbb(csk, snk, sr, si)
    if s.slabflag:
        oracle.orc_slabcorr(ks.h, at.nlocal, np.ascontiguousarray(at.x), at.q, at.echeck, len(xele), np.ascontiguousarray(xele), b_ew)
    assert rel_err(b_pppm, b_ew) < acc                         # PPPM vs Ewald within the mesh accuracy
    pp.close(); ks.close(); fx.close()


def test_pppm_full_update_close_to_ewald():
    """whole charge update with the pppm keyword (A matrix from the Ewald provider as in pppm_conp.cpp:91-101)"""
    s = systems.deck("dilute", "ffield", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    res = {}
    for name, kw in (("ewald", {}), ("pppm", dict(extra_args=["pppm"], pppm_mesh=(27, 24, 144), pppm_order=5))):
        at.q[:] = s.q[at.owner]
        fx = FixConp(s, **kw)
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        fx.setup_pre_force(at, 0, s.potdiff)
        res[name] = at.q[at.echeck != 0].copy()
        loc = slice(0, at.nlocal)
        assert abs(at.q[loc][at.echeck[loc] != 0].sum()) < 1e-12
        fx.close()
    assert rel_err(res["pppm"], res["ewald"]) < 2e-3


def test_pppm_keyword_without_mesh_is_the_reference_error():
    from conp_amd import ConpError
    s = systems.deck("dilute", "ffield", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s, extra_args=["pppm"])
    fx.init_lists(alist, blist)
    with pytest.raises(ConpError) as e:
        fx.setup_post_neighbor(at)
    assert "couldn't detect a pppm/conp kspace style" in str(e.value)      # fix_conp.cpp:404
    fx.close()
