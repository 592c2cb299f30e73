"""-m gpu: the HIP library against the CPU oracle on identical inputs, through the C ABI.

Tolerances (north_star: "charges match the reference CPU fix_conp to a stated floating-point tolerance;
electroneutrality projection and electrode-index bookkeeping bit-exact"):
  * integer tables / index maps: bit-exact;
  * structure factors, b vector: 1e-11 of the largest entry (different summation order over <= 3e4 atoms and an
    angle-addition recurrence of up to kzmax steps on both sides);
  * A matrix: 1e-11 of the largest entry;  projected inverse, elesetq, charges: 1e-8 relative to the largest entry
    (error amplified by cond(A); the library's blocked Gauss-Jordan vs the oracle's plain LU);
  * inv_project on a GIVEN matrix: bit-exact.
"""
import numpy as np
import pytest

from conp_amd import FixConp, capi, neighbor, systems
from helpers import OracleRun, rel_err

pytestmark = pytest.mark.gpu

TOL_S = 1e-11
TOL_A = 1e-11
TOL_Q = 1e-8

CASES = {
    "dilute_ffield": lambda: systems.deck("dilute", "ffield", etypes=True, shuffle_seed=3),
    "dilute_slab": lambda: systems.deck("dilute", "slab", etypes=True),
    "dilute_slab_generic_list": lambda: systems.deck("dilute", "slab", etypes=False, shuffle_seed=5),
    "small_ffield": lambda: systems.small_random(ne_side=4, n_elyte=96, lz=60.0),
    "small_slab": lambda: systems.small_random(ne_side=4, n_elyte=96, lz=60.0, mode="slab"),
    # rough electrodes: every atom has its own z -> the general (non z-class) projection kernel
    "small_rough_ffield": lambda: rough(systems.small_random(ne_side=4, n_elyte=96, lz=60.0)),
    "dilute_rough_slab": lambda: rough(systems.deck("dilute", "slab", etypes=True)),
    # tall slab box: kzmax > 160 -> more than one kz column tile, in the planar fast path and in the general projection kernel
    "small_tall_slab": lambda: systems.small_random(ne_side=4, n_elyte=96, lz=400.0, mode="slab"),
    "small_rough_tall_slab": lambda: rough(systems.small_random(ne_side=4, n_elyte=96, lz=400.0, mode="slab")),
}


def rough(s, amp=0.05, seed=4):
    rng = np.random.default_rng(seed)
    ele = s.echeck != 0
    s.x[ele, 2] += rng.uniform(-amp, amp, size=int(ele.sum()))
    return s


def run_pair(oracle, s, extra=(), okw=None, special_frac=0.0):
    at, alist, blist = neighbor.build_lists(s, special_frac=special_frac)
    o = OracleRun(oracle, s, at, alist, blist, **(okw or {}))
    o.setup()
    fx = FixConp(s, extra_args=extra)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    return at, o, fx


@pytest.mark.parametrize("case", list(CASES))
def test_full_chain_matches_oracle(oracle, case):
    s = CASES[case]()
    at, o, fx = run_pair(oracle, s, special_frac=0.05)
    info = fx.info()
    # ---- k tables and index maps: bit-exact
    kt = fx.ktables()
    ks = o.fx.ks
    assert info.kcount == ks.kcount and info.kcount_flat == ks.kcount_flat and info.kcount_expand == ks.kcount_expand
    assert list(info.kcount_dims) == list(ks.kcount_dims)
    for name in ("kxvecs", "kyvecs", "kzvecs", "kxy_list", "kz_list"):
        assert np.array_equal(kt[name], getattr(ks, name)), name
    assert np.array_equal(kt["ug"], ks.ug)           # same expression, same libm
    mo, mg = o.fx.maps(), fx.maps()
    for name in mo:
        if name != "elecheck_eleall":      # filled by b_setq_cal during setup (fix_conp.cpp:631)
            assert np.array_equal(mo[name], mg[name]), name
    # ---- A matrix
    fx.a_cal(at)
    A_g = fx.matrix()
    o2 = OracleRun(oracle, s, at, *neighbor.build_lists(s, special_frac=0.05)[1:])
    o2.fx.lib.orc_fix_a_cal(o2.fx.h)
    A_o = o2.fx.matrix()
    assert rel_err(A_g, A_o) < TOL_A
    assert np.array_equal(A_g, A_g.T)
    c_g, s_g = fx.ele_trig()
    c_o, s_o = o2.fx.trig()
    assert np.array_equal(c_g, c_o) and np.array_equal(s_g, s_o)
    o2.fx.close()
    fx.close()
    # ---- full setup + one charge update through the hooks
    fx = FixConp(s)
    alist, blist = neighbor.build_lists(s, special_frac=0.05)[1:]
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    q_before = at.q.copy()
    fx.setup_pre_force(at, 0, s.potdiff)
    o.pre_force(s.potdiff)
    sr_o, si_o = ks.sincos_b(at.x, q_before, at.echeck, at.nlocal)
    sr_g, si_g = fx.sfac()
    scale = max(np.abs(sr_o).max(), np.abs(si_o).max())
    assert np.abs(sr_g - sr_o).max() / scale < TOL_S
    assert np.abs(si_g - si_o).max() / scale < TOL_S
    b_o, q_o, sq_o = o.fx.vectors()
    b_g, q_g, sq_g = fx.vectors()
    assert rel_err(b_g, b_o) < 1e-10
    S_o, S_g = o.fx.matrix(), fx.matrix()
    assert rel_err(S_g, S_o) < TOL_Q
    assert rel_err(sq_g, sq_o) < TOL_Q
    assert rel_err(q_g, q_o) < TOL_Q
    ele = at.echeck != 0
    assert rel_err(at.q[ele], o.q[ele]) < TOL_Q            # owned AND ghost electrode atoms
    assert np.array_equal(at.q[~ele], q_before[~ele])       # electrolyte untouched
    assert abs(at.q[:at.nlocal][at.echeck[:at.nlocal] != 0].sum()) < 1e-12   # electroneutral
    assert np.array_equal(o.fx.maps()["elecheck_eleall"], fx.maps()["elecheck_eleall"])
    assert (fx.info().n_zclasses == 0) == ("rough" in case)      # planar decks take the z-class fast path
    if "tall" in case:
        assert fx.info().kzmax > 160                              # really more than one column tile
    sc_o = o.fx.scalars()
    assert fx.compute_scalar() == pytest.approx(sc_o["scalar_output"], rel=1e-7, abs=1e-12)
    assert fx.info().totsetq == pytest.approx(sc_o["totsetq"], rel=1e-7)
    fx.close()
    o.fx.close()


@pytest.mark.parametrize("deck,mode", [("il_twolayer", "ffield"), ("il_onelayer", "slab"), ("dilute", "ffield")])
def test_planar_a_matrix_factorisation_equals_the_general_contraction(deck, mode, monkeypatch):
    """planar electrodes: the k-space part of A is contracted over the planar rows only (z classes, like the projection's fast
    path) instead of over every (planar, kz) pair -- the same matrix to rounding (CONP_PATH_A_GENERAL forces the general kernel)"""
    s = systems.deck(deck, mode, etypes=(deck != "dilute"))
    at, alist, blist = neighbor.build_lists(s)
    mats = []
    for general in (False, True):
        if general:
            capi.load_library().conp_debug_set_paths(capi.PATH_A_GENERAL)
        fx = FixConp(s)
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        fx.a_cal(at)
        assert fx.info().n_zclasses in (2, 4, 6)             # dilute: three-layer electrodes
        mats.append(fx.matrix())
        fx.close()
    a_zc, a_gen = mats
    assert np.array_equal(a_zc, a_zc.T)
    assert np.abs(a_zc - a_gen).max() <= 1e-12 * np.abs(a_gen).max()


def test_dilute_step0_charge_matches_persist_log():
    """the reference's own known answer (tests/dilute/persist.log:143) straight from the GPU path"""
    import json, os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dilute_persist.json")))
    s = systems.deck("dilute", "ffield", etypes=True, g_ewald=gold["g_ewald"])
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, 1.0)
    loc = slice(0, at.nlocal)
    qleft = at.q[loc][at.echeck[loc] == 1].sum()
    qright = at.q[loc][at.echeck[loc] == -1].sum()
    assert qleft == pytest.approx(gold["thermo"][0][3], rel=5e-7)
    assert qright == pytest.approx(gold["thermo"][0][4], rel=5e-7)
    assert abs(qleft + qright) < 1e-14
    fx.close()


def test_inv_project_bit_exact(oracle):
    """electroneutrality projection: same matrix in -> bitwise identical matrix out (fix_conp.cpp:982-1067)"""
    rng = np.random.default_rng(11)
    s = systems.small_random()
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    for n in (7, 64, 333):
        m = rng.normal(size=(n, n)); a = m @ m.T / n + np.eye(n)
        ainv = np.linalg.inv(a)
        z = rng.uniform(-1, 1, size=n)
        for zneutr in (False, True):
            ref = ainv.copy()
            tot_o = oracle.orc_inv_project(n, ref, 1, int(zneutr), np.ascontiguousarray(z), 0.0)
            got, tot_g = fx.inv_project(ainv, True, zneutr, z, 0.0)
            assert tot_g == tot_o
            assert np.array_equal(got, ref), (n, zneutr)
        # nonneutral keyword: matrix untouched, <e,e> still reported
        got, tot_g = fx.inv_project(ainv, False, False, z, 0.0)
        assert np.array_equal(got, ainv)
    fx.close()


def test_inverse_matches_numpy():
    """the in-place blocked Gauss-Jordan inverse (stands for dgetrf_/dgetri_): general matrices that need row
    pivoting, sizes that are not multiples of the 64-column block, and a singular matrix -> the reference's "Inversion failed!" error"""
    from conp_amd import ConpError
    rng = np.random.default_rng(3)
    s = systems.small_random()
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    for n in (5, 64, 130, 257, 700):
        a = rng.normal(size=(n, n))                      # no diagonal dominance: pivoting is exercised
        inv = fx.invert(a)
        ref = np.linalg.inv(a)
        assert np.abs(inv - ref).max() / np.abs(ref).max() < 1e-9 * max(1.0, np.linalg.cond(a) / 1e4), n
        assert np.abs(inv @ a - np.eye(n)).max() < 1e-8 * max(1.0, np.linalg.cond(a) / 1e4), n
    # the panel factorisation runs on many cooperating workgroups: more rows than one workgroup holds, more than 64 rows per
    # workgroup (forced by capping the workgroup count), and the single-workgroup fallback -- same pivots, same inverse
    import os
    a = rng.normal(size=(2500, 2500))
    ref = np.linalg.inv(a)
    results = []
    for env in ({}, {"CONP_PANEL_MAXG": "5"}, {"CONP_PANEL_SINGLE": "1"}):
        os.environ.update(env)
        try:
            inv = fx.invert(a)
        finally:
            for k in env:
                del os.environ[k]
        assert np.abs(inv - ref).max() / np.abs(ref).max() < 1e-9 * max(1.0, np.linalg.cond(a) / 1e4), env
        results.append(inv)
    assert np.abs(results[0] - results[1]).max() <= 1e-12 * np.abs(ref).max()
    assert np.abs(results[0] - results[2]).max() <= 1e-12 * np.abs(ref).max()
    # the panel's grid barrier gives up (forced: zero polls allowed): info = -7, the matrix is restored from its copy and the
    # one-workgroup panel repeats the inverse -- the bits of the CONP_PANEL_SINGLE run, and the fall-back says so
    fx.mesg_drain()
    os.environ["CONP_PANEL_SPIN"] = "0"
    try:
        inv = fx.invert(a)
    finally:
        del os.environ["CONP_PANEL_SPIN"]
    assert "timed out at its grid barrier" in fx.mesg_drain()
    assert np.array_equal(inv, results[2])
    m = rng.normal(size=(300, 300)); spd = m @ m.T / 300 + np.eye(300)
    inv = fx.invert(spd)
    assert np.abs(inv - np.linalg.inv(spd)).max() / np.abs(inv).max() < 1e-12
    # An exactly symmetric, positive definite matrix (the electrode matrix is one) is eliminated on its own diagonal blocks --
    # no pivot search, no grid barrier; the same inverse as the pivoted elimination (CONP_PATH_INV_PIVOTED) to rounding.  A symmetric
    # matrix that is NOT positive definite is noticed (a pivot <= 0), restored and pivoted; a general matrix never tries.
    assert fx.info().inverse_path == 1
    m = rng.normal(size=(1500, 1500)); spd = m @ m.T / 1500 + 0.5 * np.eye(1500)
    inv_spd = fx.invert(spd)
    assert fx.info().inverse_path == 1
    capi.load_library().conp_debug_set_paths(capi.PATH_INV_PIVOTED)
    try:
        inv_gen = fx.invert(spd)
    finally:
        capi.load_library().conp_debug_set_paths(0)
    assert fx.info().inverse_path == 2
    ref = np.linalg.inv(spd)
    assert np.abs(inv_spd - ref).max() / np.abs(ref).max() < 1e-12 and np.abs(inv_gen - ref).max() / np.abs(ref).max() < 1e-12
    assert np.abs(inv_spd @ spd - np.eye(1500)).max() < 1e-10
    indef = spd.copy(); indef[700:, 700:] *= -1.0; indef = (indef + indef.T) / 2        # symmetric, indefinite, well conditioned
    inv = fx.invert(indef)
    assert fx.info().inverse_path == 2
    assert np.abs(inv @ indef - np.eye(1500)).max() < 1e-9
    fx.invert(a)                                                                         # the general matrix from above
    assert fx.info().inverse_path == 2
    sing = rng.normal(size=(80, 80)); sing[17] = 0.0
    with pytest.raises(ConpError) as e:
        fx.invert(sing)
    assert "Inversion failed" in str(e.value) and e.value.code == -4
    # an exact zero pivot BEFORE the last column (a zero row is pivoted to the end and never gets there): a zero column, and
    # duplicated electrode atoms (two equal rows and columns) -- in the multi-workgroup panel, past the first panel, and in the
    # one-workgroup panel.  All must end in the reference's error, never in an out-of-range row swap.
    for n, col in ((80, 17), (200, 130), (700, 333)):
        zc = rng.normal(size=(n, n)); zc[:, col] = 0.0
        with pytest.raises(ConpError) as e:
            fx.invert(zc)
        assert "Inversion failed" in str(e.value) and e.value.code == -4, (n, col)
    import os
    os.environ["CONP_PANEL_SINGLE"] = "1"
    try:
        zc = rng.normal(size=(200, 200)); zc[:, 130] = 0.0
        with pytest.raises(ConpError):
            fx.invert(zc)
    finally:
        del os.environ["CONP_PANEL_SINGLE"]
    m = rng.normal(size=(150, 150)); dup = m @ m.T + 150 * np.eye(150)
    dup[40] = dup[7]; dup[:, 40] = dup[:, 7]; dup[40, 40] = dup[7, 7]            # atom 40 is a copy of atom 7
    # (numerically singular, but the eliminated pivot is a rounding residue, not an exact zero: dgetrf_ would not flag it either.
    #  Whatever comes back -- the error or a huge inverse -- the call must return.)
    try:
        fx.invert(dup)
    except ConpError as e:
        assert e.code == -4
    nanm = rng.normal(size=(100, 100)); nanm[3, 5] = np.nan
    with pytest.raises(ConpError):
        fx.invert(nanm)
    good = rng.normal(size=(90, 90))                       # the handle is still usable afterwards
    assert np.abs(fx.invert(good) @ good - np.eye(90)).max() < 1e-8 * max(1.0, np.linalg.cond(good) / 1e4)
    fx.close()


@pytest.mark.parametrize("mode", ["slab", "ffield"])
def test_kspace_provider_surface_matches_oracle(oracle, mode):
    """INTEGRATION.md mode B: the reference's FixConp stays on the CPU and only the k-space provider is replaced.  The three
    calls KSpaceModuleHip makes (conp_km_conp_setup / a_cal / b_cal = km_ewald.cpp:63-132, 147-151, 153-167) against the
    oracle's provider functions: k-space A incl. its diagonal and slab term, one orientation per pair; k-space b incl. slab"""
    import oracle_py
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0, mode=mode)
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    n = at.nlocal
    fx.km_conp_setup(float((at.q[:n] ** 2).sum()), n)
    a_g = fx.km_a_cal(at)
    b_g = fx.km_b_cal(at)
    ks = oracle_py.KSpace.from_system(oracle, s)
    for name in ("kxvecs", "kyvecs", "kzvecs", "ug"):
        assert np.array_equal(fx.ktables()[name], getattr(ks, name)), name
    m = fx.maps()
    loc = {int(t): i for i, t in enumerate(at.tag[:n])}
    xele = np.array([at.x[loc[int(t)]] for t in m["eleall2tag"]])
    csk, snk = ks.ele_trig(xele)
    a_o = ks.aaa(csk, snk, xele)
    # either orientation of a pair may carry the value (fix_conp.cpp:826-831 symmetrises): compare the symmetrised matrices
    sym = lambda a: np.tril(a, -1) + np.tril(a, -1).T + np.triu(a, 1) + np.triu(a, 1).T + np.diag(np.diag(a))
    assert rel_err(sym(a_g), sym(a_o)) < 1e-11
    sr, si = ks.sincos_b(at.x, at.q, at.echeck, n)
    b_o = ks.bbb(csk, snk, sr, si)
    if s.slabflag:
        oracle.orc_slabcorr(ks.h, n, np.ascontiguousarray(at.x), at.q, at.echeck, len(xele), xele, b_o)
    assert rel_err(b_g, b_o) < 1e-11
    ks.close(); fx.close()


def test_cg_solver_matches_oracle(oracle):
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0)
    at, alist, blist = neighbor.build_lists(s)
    o = OracleRun(oracle, s, at, alist, blist, minimizer=0)
    o.setup()
    o.pre_force(s.potdiff)
    fx = FixConp(s, extra_args=["cg"])
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    b_o, q_o, sq_o = o.fx.vectors()
    b_g, q_g, sq_g = fx.vectors()
    # CG stops at (r.p)/Ne < 1e-6: both sides agree far below the solver's own tolerance
    assert rel_err(sq_g, sq_o) < 1e-6
    assert rel_err(q_g, q_o) < 1e-6
    assert abs(fx.info().cg_iterations - o.fx.sizes()["cg_iters"]) <= 1
    ele = at.echeck != 0
    assert rel_err(at.q[ele], o.q[ele]) < 1e-6
    fx.close(); o.fx.close()


@pytest.mark.parametrize("split", [False, True])
def test_log_file_lines_inverse_solver(split, monkeypatch):
    """what the reference prints to its log file (fix_conp.cpp:787, 857, 564-566), in the same order and format.  By default the
    pair sums share a launch with the k-space phases (Coulomb time ~ 0, all of b_cal under Kspace); CONP_PATH_TIME_SPLIT (read when
    the handle is created) launches the halves separately so that each gets its own figure."""
    import re
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0)
    at, alist, blist = neighbor.build_lists(s)
    if split:
        capi.load_library().conp_debug_set_paths(capi.PATH_TIME_SPLIT)
    fx = FixConp(s, extra_args=[])
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    lines = fx.log_drain().splitlines()
    assert lines[0] == "A matrix calculating ..."
    m = re.fullmatch(r"A matrix calculation time  = (\S+)", lines[1])
    assert m and 0.0 < float(m.group(1)) < 60.0
    assert fx.log_drain() == ""                      # drained
    mesg = fx.mesg_drain().splitlines()              # utils::logmesg lines :1006-1009, :458-461
    assert mesg[0].startswith("conp output: <e,e> = ") and mesg[1].startswith("conp output: <d,d> = ") and len(mesg) == 2
    inf = fx.info()
    assert float(mesg[0].split("=")[1]) == pytest.approx(inf.totinve * systems.EVSCALE, rel=1e-7)
    assert float(mesg[1].split("=")[1]) == pytest.approx(-inf.totsetq, rel=1e-7)
    for step in range(1, 4):
        fx.pre_force(at, step, s.potdiff)
    assert fx.log_drain() == ""                      # nothing is printed per step
    fx.write_timing()
    lines = fx.log_drain().splitlines()
    names = ["B vector calculation time = ", "Coulomb calculation time = ", "Kspace calculation time = "]
    assert [l[:len(n)] for l, n in zip(lines, names)] == names and len(lines) == 3
    tb, tc, tk = (float(l.split("=")[1]) for l in lines)
    assert tb > 0 and tk > 0 and (tc > 0 if split else tc >= 0) and abs(tb - (tc + tk)) <= 1e-9 + 1e-6 * tb   # three b_cal calls, seconds
    assert tb < 5.0
    fx.close()


def test_cg_one_launch_per_iteration_gives_the_bits_of_the_two_launch_form(monkeypatch):
    """cg_step_kernel (every workgroup repeats the vector update, then multiplies its rows) against the round-1 form (matvec
    kernel + one-workgroup update kernel per iteration, CONP_PATH_CG_TWO_LAUNCH, read when the handle is created): same iteration
    count, same residual history, bitwise the same charges -- also when the first batch is too short (maxiter path: 'tol' small
    enough for several read-backs) and when convergence falls inside a batch"""
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0)
    for extra in (["cg"], ["cg", "tol", "1e-14", "maxiter", "60"], ["cg", "maxiter", "3"]):
        got = []
        # three forms: one launch per iteration (cg_step_kernel, the default), ONE persistent launch per solve with a grid barrier
        # per iteration (round 5, cg_persist_kernel: measured slower, a test path), two launches per iteration (round 1)
        for form in (0, capi.PATH_CG_PERSIST, capi.PATH_CG_TWO_LAUNCH):
            capi.load_library().conp_debug_set_paths(form)
            at, alist, blist = neighbor.build_lists(s)
            fx = FixConp(s, extra_args=extra)
            fx.init_lists(alist, blist)
            fx.setup_post_neighbor(at)
            fx.setup_pre_force(at, 0, s.potdiff)
            for step in (1, 2, 3):
                fx.pre_force(at, step, 0.9 * s.potdiff)
            lines = [l for l in fx.log_drain().splitlines() if l.startswith(("Iteration", "*****"))]
            got.append((fx.info().cg_iterations, lines, at.q.copy()))
            fx.close()
        for other in (1, 2):
            assert got[0][0] == got[other][0], extra
            assert got[0][1] == got[other][1], extra
            assert np.array_equal(got[0][2], got[other][2]), extra


def test_log_file_lines_cg(oracle):
    """CG residual lines (fix_conp.cpp:919-928): one per iteration, the last one with the net charge"""
    import re
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0)
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s, extra_args=["cg"])
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    lines = [l for l in fx.log_drain().splitlines() if not l.startswith("A matrix")]
    n = fx.info().cg_iterations
    # two solves during setup: the preset vector (linalg_setup -> equation_solve, :456) and the first pre_force (:390)
    first_end = next(k for k, l in enumerate(lines) if l.startswith("*****"))
    lines = lines[first_end + 1:]
    assert n >= 2 and len(lines) == n
    res = []
    for k, l in enumerate(lines[:-1], start=1):
        m = re.fullmatch(r"Iteration (\d+): res = (\S+)", l)
        assert m and int(m.group(1)) == k
        res.append(float(m.group(2)))
    m = re.fullmatch(r"\*\*\*\*\* Converged at iteration (\d+)\. res = (\S+) netcharge = (\S+)", lines[-1])
    assert m and int(m.group(1)) == n
    assert float(m.group(2)) / fx.info().elenum_all < 1e-6 <= res[-1] / fx.info().elenum_all    # the stop rule :915
    assert abs(float(m.group(3))) < 1e-9                                                         # neutrality constraint
    fx.close()


def test_pre_force_respects_nevery_and_reneighbor(oracle):
    """Nevery gate (fix_conp.cpp:546) and a re-neighbour with re-ordered atoms: permanent numbering survives"""
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0)
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s, extra_args=[])
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    maps0 = fx.maps()
    q_ref = at.q.copy()
    # re-order the owned atoms (LAMMPS sorts), move the electrolyte a little, rebuild lists
    rng = np.random.default_rng(5)
    perm = rng.permutation(s.natoms)
    s2 = s.copy()
    s2.x, s2.q, s2.type, s2.tag, s2.echeck = s.x[perm].copy(), at.q[:s.natoms][perm].copy(), s.type[perm].copy(), s.tag[perm].copy(), s.echeck[perm].copy()
    mob = s2.echeck == 0
    s2.x[mob] += rng.normal(scale=0.05, size=(mob.sum(), 3))
    at2, alist2, blist2 = neighbor.build_lists(s2)
    fx.init_lists(alist2, blist2)
    fx.post_neighbor(at2)
    maps1 = fx.maps()
    assert np.array_equal(maps0["eleall2tag"], maps1["eleall2tag"])       # permanent numbering
    assert np.array_equal(maps0["tag2eleall"], maps1["tag2eleall"])
    assert not np.array_equal(maps0["ele2eleall"], maps1["ele2eleall"])   # volatile numbering did change
    fx.pre_force(at2, 1, s.potdiff)
    # oracle on the re-ordered system with the SAME permanent numbering: rebuild from scratch and compare per tag
    o = OracleRun(oracle, s2, at2, alist2, blist2)
    o.setup(); o.pre_force(s.potdiff)
    ele = at2.echeck != 0
    assert rel_err(at2.q[ele], o.q[ele]) < TOL_Q
    fx.close(); o.fx.close()


def test_device_update_graph_replay_matches_direct_launches():
    """with CONP_GRAPH=1 conp_fix_pre_force_device replays the update as a HIP graph from its second call on: same charges as
    the host-buffer path (direct launches) for moving atoms, a changed potential difference and across a re-neighbour"""
    import os
    import torch
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0)
    at, alist, blist = neighbor.build_lists(s)
    os.environ["CONP_GRAPH"] = "1"                     # read when the handle is created
    try:
        fx = FixConp(s)
    finally:
        del os.environ["CONP_GRAPH"]
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    ref = FixConp(s)                                   # second handle: host-buffer hooks only (never captured)
    ref.init_lists(alist, blist)
    ref.setup_post_neighbor(at)
    ref.setup_pre_force(at, 0, s.potdiff)
    rng = np.random.default_rng(5)
    sol = at.echeck == 0
    ele = at.echeck != 0
    d_x = torch.from_numpy(np.ascontiguousarray(at.x)).cuda()
    d_q = torch.from_numpy(at.q.copy()).cuda()
    dv = s.potdiff
    for step in range(1, 9):
        if step == 5:
            dv = 0.5 * s.potdiff                       # new potential difference: the graph is re-captured
        if step == 7:
            fx.post_neighbor(at); ref.post_neighbor(at)    # any other ABI call drops the graph
        at.x[sol] += rng.normal(scale=0.02, size=(int(sol.sum()), 3))      # well inside the skin
        d_x.copy_(torch.from_numpy(np.ascontiguousarray(at.x)))
        fx.pre_force_device(d_x.data_ptr(), d_q.data_ptr(), dv)
        torch.cuda.synchronize()
        q_dev = d_q.cpu().numpy()
        ref.pre_force(at, step, dv)
        assert np.array_equal(q_dev[ele], at.q[ele]), step      # same kernels, same order -> bitwise equal
    fx.close(); ref.close()


@pytest.mark.parametrize("seed", range(10))
def test_random_geometries_match_the_oracle(oracle, seed):
    """a sweep the decks do not cover: random box shapes, sheet counts, ion numbers (electrolyte counts that are no multiple of
    the 16-atom chunk), Ewald parameters, boundary modes, shuffled atom order -- and in every other draw electrodes whose atoms
    have been moved off their planes (rough electrodes: hundreds of distinct z values, the general projection kernels).  Charges,
    b, the matrix and the structure factors against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    ne_side = int(rng.choice([4, 5, 6, 8]))                 # box edges stay above the cutoff
    layers = int(rng.choice([1, 2]))
    n_elyte = int(rng.choice([24, 52, 100, 212, 404]))
    lz = float(rng.choice([44.0, 60.0, 88.0]))
    mode = str(rng.choice(["ffield", "slab"]))
    g_ewald = float(rng.choice([0.28, 0.35, 0.45]))
    cutoff = float(rng.choice([6.0, 8.0]))
    s = systems.synthetic(n_cells_x=ne_side, n_cells_y=max(1, ne_side // 2), lz=lz, n_elyte=n_elyte, layers=layers, cutoff=cutoff,
                          accuracy_relative=float(rng.choice([1e-4, 1e-5, 1e-6])), g_ewald=g_ewald, mode=mode, seed=50 + seed,
                          min_dist=1.5, shuffle_seed=seed if seed % 3 else None, name=f"sweep{seed}:{mode}")
    if seed % 2:
        ele = s.echeck != 0
        s.x[ele, 2] += rng.normal(scale=0.15, size=int(ele.sum()))          # rough electrodes
    at, alist, blist = neighbor.build_lists(s)
    o = OracleRun(oracle, s, at, alist, blist)
    o.setup()
    fx = FixConp(s)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    q0 = at.q.copy()
    fx.setup_pre_force(at, 0, s.potdiff)
    o.pre_force(s.potdiff)
    sr_o, si_o = o.fx.ks.sincos_b(at.x, q0, at.echeck, at.nlocal)
    sr_g, si_g = fx.sfac()
    scale = max(np.abs(sr_o).max(), np.abs(si_o).max())
    assert max(np.abs(sr_g - sr_o).max(), np.abs(si_g - si_o).max()) / scale < 1e-11
    b_o, q_o, sq_o = o.fx.vectors()
    b_g, q_g, sq_g = fx.vectors()
    assert rel_err(b_g, b_o) < 1e-10
    assert rel_err(fx.matrix(), o.fx.matrix()) < 1e-7
    ele = at.echeck != 0
    assert rel_err(at.q[ele], o.q[ele]) < 1e-7
    # a second update with moved ions
    sol = at.echeck == 0
    at.x[sol] += rng.normal(scale=0.03, size=(int(sol.sum()), 3))
    fx.pre_force(at, 1, 0.7 * s.potdiff)
    o.pre_force(0.7 * s.potdiff)
    assert rel_err(at.q[ele], o.q[ele]) < 1e-7
    fx.close(); o.fx.close()


def test_ghost_image_mode_uploads_owned_atoms_only_and_gives_the_same_bits():
    """conp_env.ghost_images (what the LAMMPS glue sets): ghosts are rebuilt on the device from their owners and image shifts
    instead of being uploaded -- identical charges, also after the atoms moved and after a re-neighbour; a ghost that is NOT an
    exact image makes the handle fall back to full uploads"""
    s = systems.deck("dilute", "slab", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    assert at.nghost > 0
    prd = s.prd
    img = np.rint((at.x[at.nlocal:] - at.x[at.owner[at.nlocal:]]) / prd)
    def refresh_ghosts():
        at.x[at.nlocal:] = at.x[at.owner[at.nlocal:]] + img * prd        # what forward communication does
        at.q[at.nlocal:] = at.q[at.owner[at.nlocal:]]
    refresh_ghosts()
    fg = FixConp(s, ghost_images=True)
    ff = FixConp(s)
    rng = np.random.default_rng(4)
    loc_sol = np.nonzero(at.echeck[:at.nlocal] == 0)[0]
    qa = at.q.copy()
    for fx in (fg, ff):
        at.q[:] = qa
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        fx.setup_pre_force(at, 0, s.potdiff)
    q_first = at.q.copy()
    for step in range(1, 5):
        at.x[loc_sol] += rng.normal(scale=0.02, size=(len(loc_sol), 3))
        refresh_ghosts()
        res = []
        for fx in (fg, ff):
            if step == 3:
                fx.post_neighbor(at)
            fx.pre_force(at, step, s.potdiff)
            res.append(at.q.copy())
        assert np.array_equal(res[0], res[1]), step
        assert not np.array_equal(res[0], q_first)
    # stale ghost arrays are fine in image mode (the library does not read them between re-neighbours) ...
    x_keep = at.x.copy()
    at.x[at.nlocal:] += 0.3
    fg.pre_force(at, 5, s.potdiff); qg = at.q.copy()
    at.x[:] = x_keep
    ff.pre_force(at, 5, s.potdiff)
    assert np.array_equal(qg, at.q)
    # ... and a ghost that is not an image at a re-neighbour switches the handle to full uploads
    at.x[at.nlocal] += 1e-3
    fg.post_neighbor(at); ff.post_neighbor(at)
    fg.pre_force(at, 6, s.potdiff); qg = at.q.copy()
    ff.pre_force(at, 6, s.potdiff)
    assert np.array_equal(qg, at.q)
    fg.close(); ff.close()


@pytest.mark.parametrize("deck,world", [("il_onelayer", 4), ("dilute", 3), ("cond2", 8)])
def test_rank_shards_of_b_add_up_to_the_single_rank_vector(deck, world):
    """k-shard (row tiles dealt by cost) + row-shard of the real-space term: the b vectors of the `world` rank handles, built
    one after the other on this GPU, sum to the unsharded b.  il_onelayer has 2 row tiles -> two of the four ranks own none;
    cond2 takes the general projection kernel (rough electrodes)."""
    s = systems.deck(deck, "ffield", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    def b_of(rank, nranks):
        fx = FixConp(s, rank=rank, nranks=nranks)
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        fx.b_cal(at)
        b = fx.vectors()[0].copy()
        r0, r1 = fx.row_range()
        fx.close()
        return b, (r0, r1)
    b1, rr = b_of(0, 1)
    assert rr == (0, len(b1))
    total = np.zeros_like(b1)
    edges = []
    for r in range(world):
        b, (r0, r1) = b_of(r, world)
        total += b
        edges.append((r0, r1))
    assert edges[0][0] == 0 and edges[-1][1] == len(b1) and all(edges[k][1] == edges[k + 1][0] for k in range(world - 1))
    assert rel_err(total, b1) < 1e-12


@pytest.mark.parametrize("deck,newton", [("il_onelayer", False), ("il_onelayer", True), ("dilute", False), ("cond2", False)])
def test_device_row_regrouping_keeps_the_list_order(deck, newton, monkeypatch):
    """the electrode rows of the real-space b are regrouped from the half list on the device (count / scan / emit / stable radix
    sort); the pairs of a row must come out in list order like the host counting sort's -- then both give the same bits"""
    s = systems.deck(deck, "ffield", etypes=(deck != "dilute"))
    s.newton = newton
    at, alist, blist = neighbor.build_lists(s)
    res = []
    for host in (False, True):
        if host:
            capi.load_library().conp_debug_set_paths(capi.PATH_ROWS_HOST)
        fx = FixConp(s)
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        fx.b_cal(at)
        res.append((fx.vectors()[0].copy(), fx.info().n_blist_pairs))
        fx.close()
    assert res[0][1] == res[1][1] > 0
    assert np.array_equal(res[0][0], res[1][0])


@pytest.mark.parametrize("deck,mode", [("il_onelayer", "ffield"), ("il_twolayer", "ffield"), ("dilute", "slab"),
                                       ("small_tall_slab", None)])
def test_projecting_epilogue_equals_the_partial_tile_path(deck, mode, monkeypatch):
    """planar electrodes: a segment of sk_gemm projects its partial tile on the z classes before it leaves the registers and the
    pieces are added afterwards; the comparison path (CONP_PATH_PARTIAL_TILES) adds the partial tiles and projects the sum -- the same
    terms re-associated.  The structure factors, which the projecting update never forms, are re-formed on request through the
    partial-tile kernels (the two paths cut the atom axis into different shares: equal to rounding, not bit for bit).
    small_tall_slab: two kz column tiles -- the comparison path is then sk_reduce + b_hc instead of the fused sk_reduce_hc."""
    s = CASES[deck]() if mode is None else systems.deck(deck, mode, etypes=(deck != "dilute"))
    at, alist, blist = neighbor.build_lists(s)
    res = []
    for partials in (False, True):
        if partials:
            capi.load_library().conp_debug_set_paths(capi.PATH_PARTIAL_TILES)
        fx = FixConp(s)
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        fx.b_cal(at)
        b = fx.vectors()[0].copy()
        sr, si = fx.sfac()
        res.append((b, sr.copy(), si.copy(), fx.info().n_zclasses))
        fx.close()
    assert res[0][3] == res[1][3] > 0
    assert rel_err(res[0][0], res[1][0]) < 1e-12
    assert rel_err(res[0][1], res[1][1]) < 1e-12 and rel_err(res[0][2], res[1][2]) < 1e-12


def _gpu_shard_worker(rank, world, port, out):
    import os, sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "lammps-user-conp2_amd"))
    from conp_amd import FixConp, capi, neighbor, systems
    from conp_amd.distributed import sharded_update
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)     # both ranks drive cuda:0; collectives over gloo
    s = systems.deck("dilute", "slab", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s, device=0, rank=rank, nranks=world)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.linalg_setup(at)
    ne = fx.info().elenum_all
    d_x = torch.from_numpy(np.ascontiguousarray(at.x)).cuda()
    d_q = torch.from_numpy(at.q.copy()).cuda()
    d_b = torch.zeros(ne, dtype=torch.float64, device="cuda")
    d_sol = torch.zeros(ne, dtype=torch.float64, device="cuda")
    fx.bind_device_buffers(d_b.data_ptr(), d_sol.data_ptr())
    r0, r1 = fx.row_range()

    class Backend:
        def b_local(self):
            fx.b_cal_device(d_x.data_ptr(), d_q.data_ptr()); torch.cuda.synchronize()
            return d_b.cpu()
        def solve_rows(self, b):
            d_b.copy_(b); fx.solve_device(s.potdiff); torch.cuda.synchronize()
            return d_sol[r0:r1].cpu()
        def finish(self, q_all):
            d_sol.copy_(q_all); fx.scatter_device(d_q.data_ptr(), s.potdiff); torch.cuda.synchronize()

    b, q_all = sharded_update(Backend(), ne, rank, world)
    out[rank] = (b.numpy().copy(), q_all.numpy().copy(), d_q.cpu().numpy().copy(), (r0, r1))
    fx.close()
    dist.destroy_process_group()


def test_two_rank_sharding_on_one_gpu_matches_single_rank():
    """k-shard + row-shard device path (conp_fix_b_cal_device / solve_device / scatter_device) with world_size 2:
    both ranks use cuda:0, the two collectives run over gloo; result == the unsharded update"""
    import torch.multiprocessing as mp
    s = systems.deck("dilute", "slab", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    b1, q1, _ = fx.vectors()
    fx.close()
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_gpu_shard_worker, args=(2, 29871, out), nprocs=2, join=True)
    for rank in (0, 1):
        b, q_all, q_atoms, rr = out[rank]
        assert rel_err(b, b1) < 1e-12 and rel_err(q_all, q1) < 1e-11
        ele = at.echeck != 0
        assert rel_err(q_atoms[ele], at.q[ele]) < 1e-11
    assert out[0][3][1] == out[1][3][0]


def test_conq_matches_oracle(oracle):
    """fix conq (fix_conq.cpp:41-90): prescribed electrode charge; the fix scalar is the potential difference"""
    s = systems.deck("dilute", "ffield", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    QR = 0.037
    o = OracleRun(oracle, s, at, alist, blist)
    o.setup()
    dv_o = o.fx.pre_force_conq(QR)
    fx = FixConp(s, style="conq")
    assert fx.args.conq == 1
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, QR)
    ele = at.echeck != 0
    assert rel_err(at.q[ele], o.q[ele]) < TOL_Q
    assert fx.compute_scalar() == pytest.approx(dv_o, rel=1e-9)
    loc = slice(0, at.nlocal)
    assert at.q[loc][at.echeck[loc] == -1].sum() == pytest.approx(QR, rel=1e-9)     # group2 carries +QR
    assert at.q[loc][at.echeck[loc] == 1].sum() == pytest.approx(-QR, rel=1e-9)
    fx.close(); o.fx.close()


def test_post_force_matches_oracle(oracle):
    """force_cal + blist_coul_cal_post_force (fix_conp.cpp:1163-1201, 1368-1444).  The Gaussian correction only acts below
    ~1.2 A (eta^2 r^2 < 5.8), so a few electrolyte atoms are pushed onto electrode atoms to exercise it."""
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0, mode="slab")
    rng = np.random.default_rng(9)
    ele_idx = np.nonzero(s.echeck != 0)[0]; sol_idx = np.nonzero((s.echeck == 0) & (s.q != 0))[0]
    for k in range(6):
        d = rng.normal(size=3); d *= rng.uniform(0.4, 1.1) / np.linalg.norm(d)
        s.x[sol_idx[k]] = s.x[ele_idx[3 * k]] + d
    s.x[:, :2] = s.boxlo[:2] + np.mod(s.x[:, :2] - s.boxlo[:2], s.prd[:2])
    for newton in (False, True):
        s.newton = newton
        at, alist, blist = neighbor.build_lists(s)
        o = OracleRun(oracle, s, at, alist, blist)
        o.setup(); o.pre_force(s.potdiff)
        fx = FixConp(s)
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        fx.setup_pre_force(at, 0, s.potdiff)
        f_o, out_o = o.fx.post_force()
        f_g, ek, ec, vir = fx.post_force(at)
        assert np.abs(f_o).max() > 1.0                                 # the correction really is exercised
        assert rel_err(f_g, f_o) < 1e-9
        assert ek == pytest.approx(out_o[0], rel=1e-9)
        assert ec == pytest.approx(out_o[1], rel=1e-9)
        assert np.allclose(vir, out_o[2:8], rtol=1e-9, atol=1e-9 * np.abs(out_o[2:8]).max())
        assert np.all(f_g[at.echeck != 0] == 0.0)                      # electrode atoms receive no force from this term
        fx.close(); o.fx.close()


def test_post_force_of_the_same_step_reuses_the_device_copy(oracle):
    """conp_fix_post_force_step: with the step number of the pre_force that just ran, positions and charges are not uploaded
    again -- same forces as the uploading call; a different step, or moved atoms with Nevery > 1, upload"""
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0, mode="slab")
    rng = np.random.default_rng(9)
    ele_idx = np.nonzero(s.echeck != 0)[0]; sol_idx = np.nonzero((s.echeck == 0) & (s.q != 0))[0]
    for k in range(6):
        d = rng.normal(size=3); d *= rng.uniform(0.4, 1.1) / np.linalg.norm(d)
        s.x[sol_idx[k]] = s.x[ele_idx[3 * k]] + d
    s.x[:, :2] = s.boxlo[:2] + np.mod(s.x[:, :2] - s.boxlo[:2], s.prd[:2])
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    fx.pre_force(at, 1, s.potdiff)
    ref = fx.post_force(at)                      # always uploads
    got = fx.post_force_step(at, 1)              # resident: step 1 is the step of the last pre_force
    assert np.abs(ref[0]).sum() > 0
    # (the energy / virial accumulators and multiply-hit force entries are f64 atomic sums: equal up to the order of the additions)
    close = lambda a, b: np.allclose(a, b, rtol=1e-12, atol=1e-12 * np.abs(ref[0]).max())
    assert close(got[0], ref[0]) and close(got[1], ref[1]) and close(got[2], ref[2]) and close(got[3], ref[3])
    # atoms move, no pre_force for the new step (Nevery > 1): the stale device copy must not be used
    at.x[sol_idx[:6]] += 0.05
    ref2 = fx.post_force(at)
    got2 = fx.post_force_step(at, 2)
    assert close(got2[0], ref2[0]) and not close(ref2[0], ref[0])
    fx.close()


@pytest.mark.parametrize("case,extra,okw", [
    ("noslab_zneutr", (), {}),                          # doubled antisymmetric cell, second neutrality constraint (fix_conp.cpp:1027-1060)
    ("qinit", ("qinit",), dict(qinit=True)),           # initial electrode charges kept as an offset (:1107-1114, 1156)
    ("nonneutral", ("nonneutral",), dict(nullneutral=False)),   # projection skipped, <e,e> still computed (:1011)
    ("newton_on", (), {}),                              # ghost contributions routed through newtonbuf (:1345-1361)
])
def test_keyword_variants_match_oracle(oracle, case, extra, okw):
    if case == "noslab_zneutr":
        s = systems.deck("dilute", "noslab_zneutr", etypes=True)
    else:
        s = systems.deck("dilute", "slab", etypes=True)
    if case == "qinit":
        rng = np.random.default_rng(2)
        ele = s.echeck != 0
        s.q[ele] = rng.normal(scale=1e-3, size=int(ele.sum()))
    if case == "newton_on":
        s.newton = True
    at, alist, blist = neighbor.build_lists(s)
    o = OracleRun(oracle, s, at, alist, blist, **okw)
    o.setup(); o.pre_force(s.potdiff)
    fx = FixConp(s, extra_args=list(extra))
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    b_o, q_o, sq_o = o.fx.vectors()
    b_g, q_g, sq_g = fx.vectors()
    assert rel_err(b_g, b_o) < 1e-10 and rel_err(q_g, q_o) < TOL_Q and rel_err(sq_g, sq_o) < TOL_Q
    ele = at.echeck != 0
    assert rel_err(at.q[ele], o.q[ele]) < TOL_Q
    assert fx.info().totinve == pytest.approx(o.fx.scalars()["totinve"], rel=1e-7)      # the logged <e,e> / evscale
    loc = slice(0, at.nlocal)
    qe = at.q[loc][at.echeck[loc] != 0]
    if case == "noslab_zneutr":
        zpos = at.x[loc][at.echeck[loc] != 0][:, 2] > 0.5 * (s.boxlo[2] + s.boxhi[2])
        assert abs(qe[zpos].sum()) < 1e-11 and abs(qe[~zpos].sum()) < 1e-11    # each half of the doubled cell is neutral
    elif case == "nonneutral":
        assert abs(qe.sum()) > 1e-6                                              # no projection -> not neutral
    elif case == "qinit":
        assert abs(qe.sum() - s.q[s.echeck != 0].sum()) < 1e-11                  # solved part neutral, offset kept
    else:
        assert abs(qe.sum()) < 1e-12
    fx.close(); o.fx.close()


def test_equivalent_formulations_give_the_same_charges():
    """the reference's own test idea (tests/*/compare.gnu): slab-corrected p p f, finite-field p p p and the
    doubled antisymmetric cell are formulations of the same physical system -> same electrode charge on the dilute deck"""
    tot = {}
    for mode in ("slab", "ffield", "noslab_zneutr"):
        s = systems.deck("dilute", mode, etypes=True)
        at, alist, blist = neighbor.build_lists(s)
        fx = FixConp(s)
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        fx.setup_pre_force(at, 0, 1.0)
        loc = slice(0, at.nlocal)
        sel = at.echeck[loc] == 1
        if mode == "noslab_zneutr":   # first copy of the cell only
            sel &= at.tag[loc] <= 432
        tot[mode] = at.q[loc][sel].sum()
        fx.close()
    assert tot["slab"] == pytest.approx(tot["ffield"], rel=2e-3)
    assert tot["noslab_zneutr"] == pytest.approx(tot["ffield"], rel=2e-2)


def test_matrix_files_round_trip(tmp_path):
    """`matout` / `org F` / `inv F` (fix_conp.cpp:721-773, 833-849, 960-977): the reference's text layouts, a permuted
    tag row re-defining the permanent numbering, and the reference's error messages"""
    import os
    from conp_amd import ConpError
    s = systems.deck("dilute", "ffield", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    cwd = os.getcwd(); os.chdir(tmp_path)
    try:
        fx = FixConp(s, extra_args=["matout"])
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        fx.setup_pre_force(at, 0, 1.0)
        q_ref = at.q.copy()
        S = fx.matrix()
        tags = fx.maps()["eleall2tag"]
        fx.close()
        # layout checks: tag row of %20d, rows of %20.12f / %20.10f
        lines = open("amatrix").read().split("\n")
        assert lines[0] == " " + "".join("%20d" % t for t in tags)
        assert len(lines[1]) == 1 + 20 * len(tags)
        A_file = np.loadtxt("amatrix", skiprows=1)
        inv_lines = open("inv_a_matrix").read().split("\n")
        assert inv_lines[1].startswith("%20.10f" % S[0, 0]) and inv_lines[1].split()[1] == ("%.10f" % S[0, 1])
        # `org amatrix`: numbering from the file, inverse recomputed -> same charges (file precision 1e-12)
        for kw, fname, tol in (("org", "amatrix", 1e-6), ("inv", "inv_a_matrix", 1e-4)):
            at.q[:] = s.q[at.owner]
            fy = FixConp(s, extra_args=[kw, fname])
            fy.init_lists(alist, blist)
            fy.setup_post_neighbor(at)
            fy.setup_pre_force(at, 0, 1.0)
            ele = at.echeck != 0
            assert rel_err(at.q[ele], q_ref[ele]) < tol, kw
            fy.close()
        # a permuted tag row: the permanent numbering follows the file (a_read :753-759), charges per atom unchanged
        perm = np.random.default_rng(0).permutation(len(tags))
        with open("amatrix_perm", "w") as fh:
            fh.write(" " + "".join("%20d" % t for t in tags[perm]) + "\n")
            for i in perm:
                fh.write(" " + "".join("%20.12f" % v for v in A_file[i][perm]) + "\n")
        at.q[:] = s.q[at.owner]
        fz = FixConp(s, extra_args=["org", "amatrix_perm"])
        fz.init_lists(alist, blist)
        fz.setup_post_neighbor(at)
        fz.setup_pre_force(at, 0, 1.0)
        assert np.array_equal(fz.maps()["eleall2tag"], tags[perm])
        ele = at.echeck != 0
        assert rel_err(at.q[ele], q_ref[ele]) < 1e-6
        fz.close()
        # error paths
        with open("short", "w") as fh:
            fh.write(" ".join(str(t) for t in tags) + "\n1.0 2.0\n")
        for fname, msg in (("short", "Too few entries in A matrix file"), ("missing_file", "Cannot open A matrix file")):
            fe = FixConp(s, extra_args=["org", fname])
            fe.init_lists(alist, blist)
            fe.setup_post_neighbor(at)
            with pytest.raises(ConpError) as e:
                fe.setup_pre_force(at, 0, 1.0)
            assert msg in str(e.value)
            fe.close()
        with open("long", "w") as fh:
            fh.write(open("amatrix").read() + " 1.0\n")
        fe = FixConp(s, extra_args=["org", "long"])
        fe.init_lists(alist, blist)
        fe.setup_post_neighbor(at)
        with pytest.raises(ConpError) as e:
            fe.setup_pre_force(at, 0, 1.0)
        assert "Too many entries in A matrix file" in str(e.value)
        fe.close()
    finally:
        os.chdir(cwd)


def _ehgo_setup(oracle, s, kappa, coeffs, special=None):
    """coeffs: list of (type, eta, u0_eV or 'auto')"""
    at, alist, blist = neighbor.build_lists(s)
    o = OracleRun(oracle, s, at, alist, blist)
    nt1 = s.ntypes + 1
    eta_i = np.zeros(nt1); u0 = np.zeros(nt1)
    fx = FixConp(s, extra_args=["ehgo"])
    assert fx.modify_param("ehgo", "kappa", kappa) == 3
    for t, eta, u in coeffs:
        assert fx.modify_param("ehgo", "coeff", t, eta, u) == 5
        lo, hi = (int(v) if v else d for v, d in zip(str(t).split("*") if "*" in str(t) else (t, t), (1, s.ntypes)))
        for ty in range(lo, hi + 1):
            eta_i[ty] = float(eta)
            u0[ty] = np.sqrt(2.0) / np.sqrt(np.pi) * float(eta) / systems.EVSCALE if u == "auto" else float(u)
    assert o.fx.set_ehgo(float(kappa), eta_i, u0) == 1
    o.setup(); o.pre_force(s.potdiff)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    return at, o, fx


def test_ehgo_pair_mode(oracle):
    """EHGO (fix_conp.cpp:1482-1598).  (a) the reference deck's usage (tests/il_onelayer/input:103-106: kappa 0, coeff etype
    1.979 auto) reduces to the plain eta model; (b) general coefficients against the oracle, incl. the post-force term"""
    from conp_amd import ConpError
    s = systems.deck("dilute", "slab", etypes=True)
    # (a)
    at, o, fx = _ehgo_setup(oracle, s, 0, [(3, 1.979, "auto")])
    q_ehgo = at.q.copy()
    at2, al2, bl2 = neighbor.build_lists(s)
    fp = FixConp(s)
    fp.init_lists(al2, bl2); fp.setup_post_neighbor(at2); fp.setup_pre_force(at2, 0, s.potdiff)
    ele = at.echeck != 0
    assert rel_err(q_ehgo[ele], at2.q[ele]) < 1e-12
    assert rel_err(q_ehgo[ele], o.q[ele]) < TOL_Q
    fp.close(); fx.close(); o.fx.close()
    # (b) Gaussian electrolyte sites with their own widths, explicit self-interaction u0, kappa 1
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0, mode="slab")
    rng = np.random.default_rng(9)
    ele_idx = np.nonzero(s.echeck != 0)[0]; sol_idx = np.nonzero((s.echeck == 0) & (s.q != 0))[0]
    for k in range(6):        # close contacts so that the post-force correction is non-zero
        d = rng.normal(size=3); d *= rng.uniform(0.5, 1.1) / np.linalg.norm(d)
        s.x[sol_idx[k]] = s.x[ele_idx[3 * k]] + d
    s.x[:, :2] = s.boxlo[:2] + np.mod(s.x[:, :2] - s.boxlo[:2], s.prd[:2])
    at, o, fx = _ehgo_setup(oracle, s, 1.0, [(5, 1.979, 11.0), ("1*2", 1.2, 6.5), (4, 0.9, "auto")])
    ele = at.echeck != 0
    b_o, q_o, sq_o = o.fx.vectors(); b_g, q_g, sq_g = fx.vectors()
    assert rel_err(b_g, b_o) < 1e-10 and rel_err(sq_g, sq_o) < TOL_Q and rel_err(at.q[ele], o.q[ele]) < TOL_Q
    assert rel_err(fx.matrix(), o.fx.matrix()) < TOL_Q
    f_o, out_o = o.fx.post_force()
    f_g, ek, ec, vir = fx.post_force(at)
    assert np.abs(f_o).max() > 0.1 and rel_err(f_g, f_o) < 1e-9
    assert ek == pytest.approx(out_o[0], rel=1e-9) and ec == pytest.approx(out_o[1], rel=1e-9)
    assert np.allclose(vir, out_o[2:8], rtol=1e-9, atol=1e-9 * np.abs(out_o[2:8]).max())
    fx.close(); o.fx.close()
    # error paths of modify_param
    fe = FixConp(s)
    with pytest.raises(ConpError) as e:
        fe.modify_param("ehgo", "kappa", 1.0)
    assert "Can't fix_modify conp parameters in basic pair mode" in str(e.value)
    fe.close()
    fe = FixConp(s, extra_args=["ehgo"])
    for toks, msg in ((("ehgo", "kappa"), "Invalid number of inputs"), (("ehgo", "coeff", "1", "2.0"), "Invalid number of inputs"),
                      (("ehgo", "frob", "1"), "Invalid entry for EHGO coeff setting")):
        with pytest.raises(ConpError) as e:
            fe.modify_param(*toks)
        assert msg in str(e.value)
    fe.close()


def test_cond_matches_oracle(oracle):
    """fix cond (fix_cond.cpp:46-126): prescribed charge, potential difference from the electrolyte dipole (ffield geometry)"""
    s = systems.deck("dilute", "ffield", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    Q = 0.021
    o = OracleRun(oracle, s, at, alist, blist)
    o.fx.lib.orc_fix_a_cal(o.fx.h); o.fx.lib.orc_fix_b_setq_cal(o.fx.h)
    setz = o.fx.vectors()[0] / systems.EVSCALE                         # cond_setup: preset vector / evscale
    assert o.fx.lib.orc_fix_equation_solve(o.fx.h) == 0
    o.fx.lib.orc_fix_get_setq(o.fx.h)
    dv_o = o.fx.pre_force_cond(Q, setz)
    fx = FixConp(s, style="cond")
    assert fx.args.cond == 1 and fx.args.conq == 0
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, Q)
    loc = slice(0, at.nlocal)
    ele = at.echeck[loc] != 0
    assert rel_err(at.q[loc][ele], o.q[loc][ele]) < TOL_Q
    assert fx.compute_scalar() == pytest.approx(dv_o, rel=1e-8)
    fx.close(); o.fx.close()


def test_ehgo_fix_modify_after_setup_reaches_the_device_tables():
    """`fix_modify ID ehgo coeff ...` issued AFTER the first setup: the reference rebuilds its per-type tables in FixConp::init()
    of every run (fix_conp.cpp:296-299), so the new widths must act on the next b vector (the A matrix stays: it is built once)."""
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0, mode="slab")
    at, alist, blist = neighbor.build_lists(s)

    def handle(coeffs):
        fx = FixConp(s, extra_args=["ehgo"])
        fx.modify_param("ehgo", "kappa", 1.0)
        for c in coeffs:
            fx.modify_param("ehgo", "coeff", *c)
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        return fx

    c1 = [(5, 1.979, 11.0), ("1*2", 1.2, 6.5)]
    c2 = [(5, 1.979, 11.0), ("1*2", 0.7, 3.0)]
    fa = handle(c1)
    fa.b_cal(at)
    b1 = fa.vectors()[0].copy()
    for c in c2:
        fa.modify_param("ehgo", "coeff", *c)              # after setup_post_neighbor: tables are re-uploaded
    fa.b_cal(at)
    b_mod = fa.vectors()[0].copy()
    fb = handle(c2)
    fb.b_cal(at)
    b2 = fb.vectors()[0].copy()
    assert np.abs(b2 - b1).max() > 1e-6 * np.abs(b1).max()           # the change matters ...
    assert np.array_equal(b_mod, b2)                                  # ... and reaches the kernels, bit for bit
    fa.close(); fb.close()
