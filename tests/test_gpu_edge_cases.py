"""-m gpu: ragged and degenerate inputs of the charge update, each against the CPU oracle through the C ABI.

The reference has no special cases for these -- its loops simply run zero or few times (km_ewald.cpp:686 filters the
electrolyte by `electrode_check == 0 && q != 0`, fix_conp.cpp:1326-1334 filters pairs) -- so the library must not have any
either: no electrolyte at all, fewer charged atoms than one 16-atom MFMA chunk, electrolyte atoms that are all neutral,
electrode sizes that are not multiples of any tile, and one group serving as both electrodes (fix_conp.cpp:295)."""
import numpy as np
import pytest

from conp_amd import FixConp, neighbor, systems
from helpers import OracleRun, rel_err

pytestmark = pytest.mark.gpu


def _trim_electrolyte(s, keep_charged):
    """keep only the first `keep_charged` charged non-electrode atoms (and every electrode atom)"""
    sol = np.nonzero((s.echeck == 0) & (s.q != 0))[0]
    drop = np.zeros(len(s.q), dtype=bool)
    drop[sol[keep_charged:]] = True
    drop |= (s.echeck == 0) & (s.q == 0)
    keep = ~drop
    s.x, s.q, s.type, s.echeck = s.x[keep].copy(), s.q[keep].copy(), s.type[keep].copy(), s.echeck[keep].copy()
    s.tag = np.arange(1, keep.sum() + 1, dtype=np.int32)
    return s


def _run_both(oracle, s, **okw):
    at, alist, blist = neighbor.build_lists(s)
    o = OracleRun(oracle, s, at, alist, blist, **okw)
    o.setup()
    o.pre_force(s.potdiff)
    fx = FixConp(s, one_electrode=bool(okw.get("one_electrode", False)))
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    return at, o, fx


@pytest.mark.parametrize("n_charged", [0, 1, 3, 15, 17])
def test_few_or_no_electrolyte_charges(oracle, n_charged):
    s = _trim_electrolyte(systems.small_random(ne_side=4, n_elyte=96, lz=60.0), n_charged)
    assert int(((s.echeck == 0) & (s.q != 0)).sum()) == n_charged
    at, o, fx = _run_both(oracle, s)
    assert fx.info().n_elyte_charged == n_charged
    b_o, q_o, sq_o = o.fx.vectors()
    b_g, q_g, sq_g = fx.vectors()
    scale = max(np.abs(b_o).max(), 1e-300)
    assert np.abs(b_g - b_o).max() <= 1e-10 * scale + 1e-14
    if n_charged == 0:
        assert not b_g.any()                                   # b = 0: the charges are dV * elesetq exactly
    ele = at.echeck != 0
    assert rel_err(at.q[ele], o.q[ele]) < 1e-8
    assert fx.compute_scalar() == pytest.approx(o.fx.scalars()["scalar_output"], rel=1e-7, abs=1e-12)
    fx.close(); o.fx.close()


def test_neutral_electrolyte_atoms_are_ignored(oracle):
    """atoms with q == 0 between the electrodes are not electrolyte (km_ewald.cpp:686) but are in the neighbour lists"""
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0)
    sol = np.nonzero(s.echeck == 0)[0]
    s.q[sol[::2]] = 0.0
    at, o, fx = _run_both(oracle, s)
    assert fx.info().n_elyte_charged == int(((s.echeck == 0) & (s.q != 0)).sum())
    assert rel_err(fx.vectors()[0], o.fx.vectors()[0]) < 1e-10
    ele = at.echeck != 0
    assert rel_err(at.q[ele], o.q[ele]) < 1e-8
    fx.close(); o.fx.close()


@pytest.mark.parametrize("drop", [1, 7, 29])
def test_electrode_sizes_off_every_tile_boundary(oracle, drop):
    """Ne = 64 - 2*drop: not a multiple of 16 / 64 / 128 and, with vacancies, no longer a perfect lattice"""
    s = systems.small_random(ne_side=4, n_elyte=64, lz=60.0)
    left = np.nonzero(s.echeck == 1)[0][:drop]
    right = np.nonzero(s.echeck == -1)[0][-drop:]
    keep = np.ones(len(s.q), dtype=bool); keep[left] = False; keep[right] = False
    s.x, s.q, s.type, s.echeck = s.x[keep].copy(), s.q[keep].copy(), s.type[keep].copy(), s.echeck[keep].copy()
    s.tag = np.arange(1, keep.sum() + 1, dtype=np.int32)
    at, o, fx = _run_both(oracle, s)
    assert fx.info().elenum_all == int((s.echeck != 0).sum())
    assert rel_err(fx.matrix(), o.fx.matrix()) < 1e-8
    ele = at.echeck != 0
    assert rel_err(at.q[ele], o.q[ele]) < 1e-8
    assert abs(at.q[:at.nlocal][at.echeck[:at.nlocal] != 0].sum()) < 1e-11
    fx.close(); o.fx.close()


def test_one_group_as_both_electrodes(oracle):
    """`fix ID g conp 1 g ...`: groupbit == jgroupbit (fix_conp.cpp:295) -- every electrode atom is in group 1, the
    projection is applied after get_setq instead of after the inverse (:958, :1115)"""
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0)
    s.echeck = np.where(s.echeck != 0, 1, 0).astype(np.int32)
    at, o, fx = _run_both(oracle, s, one_electrode=True)
    b_o, q_o, sq_o = o.fx.vectors()
    b_g, q_g, sq_g = fx.vectors()
    assert rel_err(b_g, b_o) < 1e-10 and rel_err(sq_g, sq_o) < 1e-8
    ele = at.echeck != 0
    assert rel_err(at.q[ele], o.q[ele]) < 1e-8
    assert fx.compute_scalar() == pytest.approx(o.fx.scalars()["scalar_output"], rel=1e-7, abs=1e-12)
    fx.close(); o.fx.close()


@pytest.mark.parametrize("seed", range(8))
def test_seeded_sweep_of_geometries_and_parameters(oracle, seed):
    """eight seeded draws of box shape, layer count, cut-off, Ewald splitting, accuracy, eta, boundary mode and solver: full
    chain (k tables bit-exact, b, charges, scalar) against the oracle -- k-table edge cases (kxmax != kymax, several kz column
    tiles on a tall box, few planar vectors on a narrow one) come out of the draw rather than out of a hand-picked list"""
    rng = np.random.default_rng(1000 + seed)
    nx, ny = int(rng.integers(2, 7)), int(rng.integers(1, 4))
    layers = int(rng.integers(1, 3))
    lz = float(rng.uniform(45.0, 140.0))
    mode = ["slab", "ffield"][int(rng.integers(0, 2))]
    s = systems.synthetic(n_cells_x=nx, n_cells_y=ny, lz=lz, n_elyte=int(rng.integers(5, 40)) * 4, layers=layers,
                          cutoff=float(rng.uniform(5.0, 9.0)), accuracy_relative=float(10 ** rng.uniform(-6.5, -4.0)),
                          g_ewald=float(rng.uniform(0.25, 0.5)), mode=mode, seed=seed, min_dist=1.5,
                          potdiff=float(rng.uniform(0.2, 3.0)), name=f"sweep{seed}")
    s.eta = float(rng.uniform(1.2, 2.4))
    cg = bool(rng.integers(0, 2))
    at, alist, blist = neighbor.build_lists(s)
    o = OracleRun(oracle, s, at, alist, blist, minimizer=0 if cg else 1)
    o.setup()
    o.pre_force(s.potdiff)
    fx = FixConp(s, extra_args=["cg"] if cg else [])
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    kt = fx.ktables()
    for name in ("kxvecs", "kyvecs", "kzvecs", "ug"):
        assert np.array_equal(kt[name], getattr(o.fx.ks, name)), (seed, name)
    b_o, q_o, sq_o = o.fx.vectors()
    b_g, q_g, sq_g = fx.vectors()
    assert rel_err(b_g, b_o) < 1e-10, seed
    tol = 1e-6 if cg else 1e-8
    ele = at.echeck != 0
    assert rel_err(at.q[ele], o.q[ele]) < tol, seed
    assert fx.compute_scalar() == pytest.approx(o.fx.scalars()["scalar_output"], rel=10 * tol, abs=1e-10), seed
    fx.close(); o.fx.close()


def test_charge_switched_on_between_reneighbourings(oracle):
    """km_ewald.cpp:685-686 tests `q != 0` at EVERY step: an electrolyte atom whose charge goes from exactly 0 to non-zero (and
    another one back to 0) without a re-neighbour must enter / leave the structure factors at once.  The library keeps a compact
    list on the device; the host-buffer hooks re-check it against atom->q at every update."""
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0, mode="slab")
    sol = np.nonzero((s.echeck == 0) & (s.q != 0))[0]
    saved = s.q[sol[:4]].copy()
    s.q[sol[:4]] = 0.0                                  # four neutral atoms at setup ...
    at, o, fx = _run_both(oracle, s)
    n0 = fx.info().n_elyte_charged
    loc = {int(t): i for i, t in enumerate(at.tag[:at.nlocal])}
    for a, qv in zip(sol[:4], saved):                   # ... that carry charge from the next step on (owned copies and ghosts)
        for arr in (at.q, o.q):
            arr[at.tag == s.tag[a]] = qv
    for a in sol[4:6]:                                  # and two that lose theirs
        for arr in (at.q, o.q):
            arr[at.tag == s.tag[a]] = 0.0
    o.pre_force(s.potdiff)
    fx.pre_force(at, 1, s.potdiff)
    assert fx.info().n_elyte_charged == n0 + 4 - 2
    b_o, q_o, _ = o.fx.vectors()
    b_g, q_g, _ = fx.vectors()
    assert np.abs(b_g - b_o).max() <= 1e-10 * np.abs(b_o).max()
    ele = at.echeck != 0
    assert rel_err(at.q[ele], o.q[ele]) < 1e-8
    fx.close(); o.fx.close()


def test_page_locked_host_arrays_give_the_same_update():
    """conp_fix_pin_host_arrays: x, q uploaded by asynchronous DMA straight out of the host's arrays instead of the staged copy --
    bitwise the same charges, also after the arrays were unpinned again, and other arrays than the pinned ones fall back"""
    s = systems.deck("il_onelayer", "ffield", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    q_ref = at.q.copy()
    ele = at.echeck != 0
    at.q[ele] = 0.0
    fx.pin_host_arrays(at)
    fx.pre_force(at, 1, s.potdiff)
    assert np.array_equal(at.q, q_ref)
    other = neighbor.Atoms(nlocal=at.nlocal, nghost=at.nghost, x=at.x.copy(), q=at.q.copy(), type=at.type, tag=at.tag, echeck=at.echeck,
                           owner=at.owner)
    other.q[ele] = 0.0
    fx.pre_force(other, 2, s.potdiff)                   # not the pinned arrays: the staged copy
    assert np.array_equal(other.q, q_ref)
    fx.unpin_host_arrays()
    at.q[ele] = 0.0
    fx.pre_force(at, 3, s.potdiff)
    assert np.array_equal(at.q, q_ref)
    fx.close()
