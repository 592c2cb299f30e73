"""-m gpu: PPPM k-space b vector (`pppm` keyword, SURVEY row a20 / BASELINE configs[3]) against the oracle's restatement of
pppm_conp.cpp's b_cal chain, and both against the Ewald b within the PPPM accuracy (the only cross-check available:
LAMMPS' PPPM internals are not part of the reference repository -> parity unpinned at that boundary)."""
import numpy as np
import pytest

import oracle_py
from conp_amd import FixConp, capi, neighbor, systems
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


def test_pppm_spread_inside_the_forward_pass_equals_the_spreading_launch(monkeypatch):
    """deck-sized systems spread the charges in the forward xy pass's own workgroups (pppm_fft_xy_kernel, real_in 3); the density
    brick + spreading launch stay for large systems and as CONP_PATH_PPPM_SPREAD_LAUNCH: same b up to the order in which a mesh point's
    contributions arrive (unordered in both forms), slab geometry included (the slab sum is cut differently)"""
    for mode, mesh in (("ffield", (40, 45, 180)), ("slab", (40, 45, 540))):
        s = systems.deck("il_onelayer", mode, etypes=True)
        at, alist, blist = neighbor.build_lists(s)
        out = {}
        for sep in (False, True):
            if sep:
                capi.load_library().conp_debug_set_paths(capi.PATH_PPPM_SPREAD_LAUNCH)
            fx = FixConp(s, extra_args=["pppm"], pppm_mesh=mesh, pppm_order=5)
            fx.init_lists(alist, blist)
            fx.setup_post_neighbor(at)
            out[sep] = fx.km_b_cal(at).copy()
            fx.close()
            if sep:
                capi.load_library().conp_debug_set_paths(0)
        assert rel_err(out[False], out[True]) < 1e-12


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


def _pppm_handle(deck, mode, mesh, order):
    s = systems.deck(deck, mode, etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s, extra_args=["pppm"], pppm_mesh=mesh, pppm_order=order)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)               # electrode atoms carry their solved charges from here on
    return s, at, alist, blist, fx


@pytest.mark.parametrize("deck,mode,mesh,order", [("dilute", "ffield", (27, 24, 144), 5), ("il_onelayer", "ffield", (36, 40, 150), 4)])
def test_pppm_density_bricks_match_oracle(oracle, deck, mode, mesh, order):
    """ele_make_rho (pppm_conp.cpp:385-426) and the make_rho override (:434-450): electrode brick, electrolyte brick, their sum"""
    s, at, alist, blist, fx = _pppm_handle(deck, mode, mesh, order)
    n = mesh[0] * mesh[1] * mesh[2]
    d, e, l = fx.pppm_make_rho(at, n)
    pp = oracle_py.Pppm(oracle, s, mesh, order)
    d_o, e_o, l_o = pp.make_rho(mesh, at.x, at.q, at.echeck, at.nlocal)
    assert np.abs(e_o).max() > 0 and np.abs(l_o).max() > 0
    assert rel_err(e, e_o) < 1e-12 and rel_err(l, l_o) < 1e-12 and rel_err(d, d_o) < 1e-12
    dv = (s.prd[0] / mesh[0]) * (s.prd[1] / mesh[1]) * (s.prd[2] * s.slab_volfactor / mesh[2])
    ele = at.echeck[:at.nlocal] != 0
    assert e.sum() * dv == pytest.approx(at.q[:at.nlocal][ele].sum(), abs=1e-10)         # the stencil weights sum to one
    assert l.sum() * dv == pytest.approx(at.q[:at.nlocal][~ele].sum(), abs=1e-10)
    pp.close(); fx.close()


@pytest.mark.parametrize("deck,mode,mesh,order,acc", [("dilute", "ffield", (27, 24, 144), 5, 2e-4),
                                                      ("dilute", "slab", (27, 24, 432), 5, 2e-4),
                                                      ("il_onelayer", "ffield", (40, 45, 180), 5, 2e-4)])
def test_pppm_potentials_match_oracle_and_ewald(oracle, deck, mode, mesh, order, acc):
    """compute_group_potential / compute_particle_potential (pppm_conp.cpp:452-534) against the oracle's restatement, and against
    the k-space potential summed directly over the reference's k list with the structure factor of ALL charges (mesh accuracy)"""
    s, at, alist, blist, fx = _pppm_handle(deck, mode, mesh, order)
    n = at.nlocal
    rng = np.random.default_rng(4)
    sel = (rng.random(n) < 0.3).astype(np.int32)
    sel[np.nonzero(at.echeck[:n] != 0)[0][:5]] = 1                      # some electrode atoms among the probes
    got = fx.pppm_group_potential(at, sel)
    pp = oracle_py.Pppm(oracle, s, mesh, order)
    want = pp.group_potential(at.x, at.q, at.echeck, n, sel)
    pick = sel != 0
    assert rel_err(got[pick], want[pick]) < 1e-10
    i0 = int(np.nonzero(pick)[0][3])
    up = fx.pppm_particle_potential(at, i0)
    assert up == pytest.approx(pp.group_potential(at.x, at.q, at.echeck, n, sel, particle=True)[i0], rel=1e-10)
    assert up - got[i0] == pytest.approx(2 * s.g_ewald * at.q[i0] / np.sqrt(np.pi), rel=1e-9, abs=1e-14)
    # direct k sum: - sum_k 2 ug_k [cos(k r_i) Re S + sin(k r_i) Im S],  S over every charged atom (electrodes included)
    kt = fx.ktables()
    info = fx.info()
    uk = np.array(info.unitk)
    kv = np.stack([kt["kxvecs"], kt["kyvecs"], kt["kzvecs"]], 1) * uk
    ph = at.x[:n] @ kv.T
    S = (at.q[:n, None] * np.exp(1j * ph)).sum(axis=0)
    probes = np.nonzero(pick)[0]
    ew = -(2 * kt["ug"] * (np.cos(ph[probes]) * S.real + np.sin(ph[probes]) * S.imag)).sum(axis=1)
    assert rel_err(got[probes], ew) < acc
    pp.close(); fx.close()


@pytest.mark.parametrize("deck,mode,mesh,eta", [("dilute", "ffield", (27, 24, 144), 0.0), ("dilute", "slab", (27, 24, 432), 1.979),
                                                ("il_onelayer", "ffield", (36, 40, 150), 1.979)])
def test_compute_potential_atom_matches_oracle(oracle, deck, mode, mesh, eta):
    """`compute potential/atom` (compute_potential_atom.cpp:120-345): pair part over the pair style's half list (with and without
    the Gaussian `eta` correction), k-space part through the PPPM provider, slab correction, volts"""
    order = 5 if mesh[0] != 36 else 4
    s, at, alist, blist, fx = _pppm_handle(deck, mode, mesh, order)
    s_gen = systems.deck(deck, mode, etypes=False)                      # the pair style's own list: generic half list
    at_g, plist, _ = neighbor.build_lists(s_gen)
    assert at_g.nlocal == at.nlocal and np.array_equal(at_g.tag, at.tag)
    nall = at.nlocal + at.nghost
    rng = np.random.default_rng(8)
    sel = (rng.random(nall) < 0.5).astype(np.int32)
    sel[at.nlocal:] = sel[at.owner[at.nlocal:]]                          # ghosts carry their owner's group membership
    etasel = (at.echeck != 0).astype(np.int32)                           # eta_check: the electrode molecules
    pp = oracle_py.Pppm(oracle, s, mesh, order)
    for kw in (dict(pair=True, kspace=True, qsum=True), dict(pair=True, kspace=False, qsum=True), dict(pair=False, kspace=True, qsum=False)):
        got = fx.compute_potential_atom(at, plist, sel, etasel, eta=eta, **kw)
        want = pp.compute_potential_atom(s, at, plist, sel, etasel, eta=eta, **kw)
        assert np.abs(want[:at.nlocal]).max() > 0
        assert rel_err(got[:at.nlocal], want[:at.nlocal]) < 1e-10, kw
    # the constant-potential property seen by the diagnostic: with the eta correction the potential on the electrode atoms of one
    # electrode is flat (that is what the charges were solved for), up to the mesh accuracy
    if eta:
        sel_all = np.ones(nall, np.int32)
        pot = fx.compute_potential_atom(at, plist, sel_all, etasel, eta=eta)
        for sign in (1, -1):
            v = pot[:at.nlocal][at.echeck[:at.nlocal] == sign]
            assert v.std() < 5e-3 * max(abs(v.mean()), s.potdiff)
    pp.close(); fx.close()


def test_per_atom_potential_reads_the_cached_brick_and_the_kept_electrolyte_brick_is_reused(oracle):
    """(i) conp_pppm_compute (collective under ranks) forms the mesh potential of the total density once; compute_particle_potential is
    then a stencil gather from that brick -- any number of calls, no further mesh solve (the electrolyte spread counter stands
    still), each equal to the group entry's value + 2 g q / sqrt(pi).  (ii) With conp_pppm_keep_density the brick b_cal made stays
    on the device and make_rho adds the electrode brick to it: the same bricks as a fresh spread, no second spread; an update
    (or conp_pre_force's call) drops it."""
    s = systems.deck("dilute", "ffield", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    mesh, order = (27, 24, 144), 5
    fx = FixConp(s, extra_args=["pppm"], pppm_mesh=mesh, pppm_order=order)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    n = mesh[0] * mesh[1] * mesh[2]
    sel = (at.echeck[:at.nlocal] == 1).astype(np.int32)
    grp = fx.pppm_group_potential(at, sel)
    n0 = fx.info().pppm_elyte_spreads
    g = s.g_ewald
    for i in np.nonzero(sel)[0][:5]:
        u = fx.pppm_particle_potential(at, int(i))
        assert u == pytest.approx(grp[i] + 2.0 * g * at.q[i] / np.sqrt(np.pi), rel=1e-12, abs=1e-14)
    assert fx.info().pppm_elyte_spreads == n0               # five per-atom calls, not one more spread / mesh solve
    fx.pppm_compute(at)
    assert fx.info().pppm_elyte_spreads == n0 + 1
    # (ii)
    d0, e0, l0 = fx.pppm_make_rho(at, n)                    # nothing kept yet: a fresh spread
    fx.pppm_keep_density(True)
    b1 = fx.km_b_cal(at).copy()
    n1 = fx.info().pppm_elyte_spreads
    d1, e1, l1 = fx.pppm_make_rho(at, n)
    assert fx.info().pppm_elyte_spreads == n1               # the electrolyte brick of b_cal was reused
    assert rel_err(l1, l0) < 1e-12 and rel_err(e1, e0) < 1e-12 and rel_err(d1, d0) < 1e-12
    fx.pppm_keep_density(True)                              # conp_pre_force of the next step: the brick is dropped
    d2, e2, l2 = fx.pppm_make_rho(at, n)
    assert fx.info().pppm_elyte_spreads == n1 + 1 and rel_err(d2, d0) < 1e-12
    b2 = fx.km_b_cal(at)
    assert rel_err(b2, b1) < 1e-12
    fx.close()
