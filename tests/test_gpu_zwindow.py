"""The z-window form of the structure-factor contraction (conp_zn.hip, round 5): planar electrodes, large systems.  The class table
Hc = sum_j A_rj K_rc(theta_j) is evaluated by interpolation from an oversampled z grid (type-2 NUFFT: 15-tap window, projection
table made once per run) instead of through all 2 nz columns of G.  Same b vector as the full kernels to ~1e-13 of its largest entry
(the parity bar for b is 1e-11); the headline-size oracle comparisons of tests/test_gpu_decks.py run on this path."""
import numpy as np
import pytest

from conp_amd import FixConp, capi, neighbor, systems
from helpers import rel_err

pytestmark = pytest.mark.gpu


def _medium(mode, seed=7):
    # 1024 electrode / 16384 electrolyte atoms: large enough for the z-window path (>= 8192 charged atoms), small enough for seconds
    return systems.synthetic_fast(n_cells_x=16, n_cells_y=8, lz=300.0, n_elyte=16384, cutoff=12.0, accuracy_relative=1e-6,
                                  g_ewald=0.26, mode=mode, seed=seed, name=f"medium {mode}")


def _b(s, at, alist, blist, mask):
    with capi.test_paths(mask):
        fx = FixConp(s)
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        fx.b_cal(at)
        b = fx.vectors()[0].copy()
        cols = fx.info().zn_cols
        fx.close()
    return b, cols


@pytest.mark.parametrize("mode", ["ffield", "slab"])
def test_z_window_b_equals_the_full_contraction(mode):
    s = _medium(mode)
    at, alist, blist = neighbor.build_lists(s)
    b_zn, cols_zn = _b(s, at, alist, blist, 0)
    b_cl, cols_cl = _b(s, at, alist, blist, capi.PATH_SK_CLASSIC)
    assert cols_zn in (32, 48) and cols_cl == 0            # the default handle really took the z-window path
    assert np.abs(b_cl).max() > 0
    assert rel_err(b_zn, b_cl) < 1e-11


def test_z_window_survives_drift_between_list_builds_and_repeats_an_update_that_overflows():
    """the list is z-ordered at a re-neighbouring; until the next one the atoms drift by less than the neighbour skin, which the
    window margins absorb (same b as the full kernels after +-1.5 A of z motion); an atom that jumps further raises the flag and the
    host-buffer update is repeated with the full kernels (same charges, one line in the message buffer)."""
    s = _medium("ffield", seed=11)
    at, alist, blist = neighbor.build_lists(s)
    rng = np.random.default_rng(5)
    res = {}
    for mask in (0, capi.PATH_SK_CLASSIC):
        with capi.test_paths(mask):
            at_k, _, _ = neighbor.build_lists(s)
            fx = FixConp(s)
            fx.init_lists(alist, blist)
            fx.setup_post_neighbor(at_k)
            fx.setup_pre_force(at_k, 0, s.potdiff)
            fx.mesg_drain()
            sol = at_k.echeck == 0
            r = np.random.default_rng(5)
            at_k.x[sol, 2] += r.uniform(-1.5, 1.5, size=int(sol.sum()))      # drift, no re-neighbouring
            fx.pre_force(at_k, 1, s.potdiff)
            q1 = at_k.q.copy()
            m1 = fx.mesg_drain()
            j = int(np.nonzero(sol)[0][17])
            at_k.x[j, 2] += 60.0                                             # one atom jumps 60 A: outside every window
            fx.pre_force(at_k, 2, s.potdiff)
            q2 = at_k.q.copy()
            m2 = fx.mesg_drain()
            res[mask] = (q1, q2, m1, m2)
            fx.close()
    ele = at.echeck != 0
    scale = np.abs(res[0][0][ele]).max()
    assert np.abs(res[0][0][ele] - res[capi.PATH_SK_CLASSIC][0][ele]).max() < 1e-9 * scale
    assert np.abs(res[0][1][ele] - res[capi.PATH_SK_CLASSIC][1][ele]).max() < 1e-9 * scale
    assert "z-window" not in res[0][2] and "z-window" in res[0][3]          # the jump was noticed, the drift was not a problem


@pytest.mark.parametrize("world", [2, 8])
def test_z_window_rank_shards_add_up(world):
    """several ranks (replicated atoms, one rank per GPU): rank r contracts the r-th share of the z-ordered list for all rows; the b
    vectors of the `world` rank handles, built one after the other on this GPU, sum to the one-rank vector (what the all-reduce of b
    does), each on the z-window path"""
    s = _medium("ffield", seed=3)
    at, alist, blist = neighbor.build_lists(s)

    def b_of(rank, nranks):
        fx = FixConp(s, rank=rank, nranks=nranks)
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        fx.b_cal(at)
        b = fx.vectors()[0].copy()
        cols = fx.info().zn_cols
        fx.close()
        return b, cols

    b1, c1 = b_of(0, 1)
    assert c1 in (32, 48)
    total = np.zeros_like(b1)
    for r in range(world):
        b, c = b_of(r, world)
        assert c in (32, 48)
        total += b
    assert rel_err(total, b1) < 1e-12


def test_z_window_ragged_list_across_the_periodic_wrap():
    """a list whose length is not a multiple of the 16-atom chunk (padding atoms in the last chunk) in a box translated along z so that
    the electrolyte straddles the periodic boundary: the z ordering starts behind the longest empty run of cells, no range straddles
    the wrap, and the result equals the full contraction's"""
    s = _medium("ffield", seed=19)
    lo, hi = s.boxlo[2], s.boxlo[2] + s.prd[2]
    s.x[:, 2] = lo + np.mod(s.x[:, 2] - lo + 0.37 * s.prd[2], s.prd[2])     # electrodes and liquid move together, wrapped into the box
    assert s.x[:, 2].min() >= lo and s.x[:, 2].max() < hi
    sol = np.nonzero((s.echeck == 0) & (s.q != 0))[0]
    s.q[sol[[3, 500, 7001, 7002, 16000]]] = 0.0                              # five atoms leave the charged list: 16379 = 16 * 1023 + 11
    at, alist, blist = neighbor.build_lists(s)
    b_zn, cols_zn = _b(s, at, alist, blist, 0)
    b_cl, cols_cl = _b(s, at, alist, blist, capi.PATH_SK_CLASSIC)
    assert cols_zn in (32, 48) and cols_cl == 0
    assert rel_err(b_zn, b_cl) < 1e-11


def _rough(mode="ffield", seed=23):
    s = _medium(mode, seed=seed)
    ele = s.echeck != 0
    s.x[ele, 2] += np.random.default_rng(seed).uniform(-0.4, 0.4, size=int(ele.sum()))      # every electrode atom its own z: no classes
    s.name = "medium, rough electrodes"
    return s


@pytest.mark.parametrize("mode", ["ffield", "slab"])          # slab: three column tiles of kz
def test_z_window_rough_electrodes_equal_the_full_contraction(mode):
    """rough electrodes have no z classes: the ranges' raw windows are summed on the z grid and transformed to the structure-factor
    matrix G (type-1 transform), which the general projection consumes as it consumes sk_reduce's; b and the structure factors equal
    the full contraction's"""
    s = _rough(mode)
    at, alist, blist = neighbor.build_lists(s)
    out = {}
    for mask in (0, capi.PATH_SK_CLASSIC):
        with capi.test_paths(mask):
            fx = FixConp(s)
            fx.init_lists(alist, blist)
            fx.setup_post_neighbor(at)
            fx.b_cal(at)
            info = fx.info()
            out[mask] = (fx.vectors()[0].copy(), fx.sfac(), info.zn_cols, info.n_zclasses)
            fx.close()
    b_zn, (sr_zn, si_zn), cols_zn, ncls = out[0]
    b_cl, (sr_cl, si_cl), cols_cl, _ = out[capi.PATH_SK_CLASSIC]
    assert ncls == 0 and cols_zn in (32, 48) and cols_cl == 0
    assert rel_err(b_zn, b_cl) < 1e-11
    scale = max(np.abs(sr_cl).max(), np.abs(si_cl).max())
    assert np.abs(sr_zn - sr_cl).max() < 1e-11 * scale and np.abs(si_zn - si_cl).max() < 1e-11 * scale


def test_device_resident_update_reports_a_window_overflow_on_the_next_call():
    """conp_fix_pre_force_device does not synchronise: an atom that left its window is seen by the call AFTER the one that used it
    (CONP_ERR_NUMERIC, conp_hip.h); from then on the handle uses the full kernels and its charges equal theirs"""
    import torch
    s = _medium("ffield", seed=29)
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s)
    fx.init_lists(alist, blist); fx.setup_post_neighbor(at); fx.linalg_setup(at)
    x = at.x.copy()
    j = int(np.nonzero((at.echeck == 0) & (at.q != 0))[0][123])
    x[j, 2] += 60.0
    d_x = torch.from_numpy(np.ascontiguousarray(x)).cuda(); d_q = torch.from_numpy(at.q.copy()).cuda()
    assert fx.info().zn_cols in (32, 48)
    fx.pre_force_device(d_x.data_ptr(), d_q.data_ptr(), s.potdiff)          # uses the window; the flag is raised on the device
    torch.cuda.synchronize()
    with pytest.raises(capi.ConpError) as e:
        fx.pre_force_device(d_x.data_ptr(), d_q.data_ptr(), s.potdiff)
    assert e.value.code == -4 and "z-window" in e.value.msg
    assert fx.info().zn_cols == 0                                           # the full kernels from now on
    fx.pre_force_device(d_x.data_ptr(), d_q.data_ptr(), s.potdiff)
    torch.cuda.synchronize()
    q_dev = d_q.cpu().numpy()
    fx.close()
    with capi.test_paths(capi.PATH_SK_CLASSIC):
        at2, _, _ = neighbor.build_lists(s)
        fc = FixConp(s)
        fc.init_lists(alist, blist); fc.setup_post_neighbor(at2); fc.linalg_setup(at2)
        d_q2 = torch.from_numpy(at2.q.copy()).cuda()
        fc.pre_force_device(d_x.data_ptr(), d_q2.data_ptr(), s.potdiff)
        torch.cuda.synchronize()
        q_ref = d_q2.cpu().numpy()
        fc.close()
    ele = at.echeck != 0
    assert np.abs(q_dev[ele] - q_ref[ele]).max() < 1e-9 * np.abs(q_ref[ele]).max()
