"""-m gpu: the reference's ionic-liquid decks (BASELINE configs[1], [2]) and the headline-size box.

il_onelayer / il_twolayer: full parity against the CPU oracle (Ne = 832 / 1664 -- the oracle needs tens of seconds for
its A matrix, so these share one oracle run per deck).  Headline size (4096 / 32768): the oracle's A build would take
hours, so the checks are size-independent properties plus an oracle comparison of the per-step b vector."""
import numpy as np
import pytest

from conp_amd import FixConp, capi, neighbor, systems
from helpers import OracleRun, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("deck,mode,extra", [
    ("il_onelayer", "ffield", ()),
    ("il_onelayer", "slab", ()),
    ("il_twolayer", "ffield", ("cg",)),        # BASELINE configs[2]: CG solver + etypes split lists
])
def test_il_decks_match_oracle(oracle, deck, mode, extra):
    s = systems.deck(deck, mode, etypes=True, shuffle_seed=11)
    at, alist, blist = neighbor.build_lists(s)
    cg = "cg" in extra
    o = OracleRun(oracle, s, at, alist, blist, minimizer=0 if cg else 1)
    o.setup()
    fx = FixConp(s, extra_args=list(extra))
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    q0 = at.q.copy()
    fx.setup_pre_force(at, 0, s.potdiff)
    o.pre_force(s.potdiff)
    info = fx.info()
    assert info.elenum_all == (832 if deck == "il_onelayer" else 1664)
    for name, v in o.fx.maps().items():
        assert np.array_equal(v, fx.maps()[name]), name
    kt = fx.ktables()
    for name in ("kxvecs", "kyvecs", "kzvecs", "kxy_list", "kz_list", "ug"):
        assert np.array_equal(kt[name], getattr(o.fx.ks, name)), name
    sr_o, si_o = o.fx.ks.sincos_b(at.x, q0, at.echeck, at.nlocal)
    sr_g, si_g = fx.sfac()
    scale = max(np.abs(sr_o).max(), np.abs(si_o).max())
    assert max(np.abs(sr_g - sr_o).max(), np.abs(si_g - si_o).max()) / scale < 1e-11
    b_o, q_o, sq_o = o.fx.vectors()
    b_g, q_g, sq_g = fx.vectors()
    assert rel_err(b_g, b_o) < 1e-10
    tol = 1e-6 if cg else 1e-8
    assert rel_err(sq_g, sq_o) < tol and rel_err(q_g, q_o) < tol
    ele = at.echeck != 0
    assert rel_err(at.q[ele], o.q[ele]) < tol
    assert abs(at.q[:at.nlocal][at.echeck[:at.nlocal] != 0].sum()) < 1e-11
    if not cg:
        assert rel_err(fx.matrix(), o.fx.matrix()) < 1e-8
    assert fx.compute_scalar() == pytest.approx(o.fx.scalars()["scalar_output"], rel=1e-6, abs=1e-10)
    fx.close(); o.fx.close()


def test_heavily_split_tiles_use_the_two_level_partial_sum(oracle, monkeypatch):
    """multi-GPU shards cut one tile into hundreds of sk_gemm segments; forced here with 256 workgroups on il_onelayer's two
    tiles (80 segments each -> sk_reduce level 1 + level 2): same structure factors and b vector"""
    capi.load_library().conp_debug_set_sk_workgroups(256)
    s = systems.deck("il_onelayer", "ffield", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    q0 = at.q.copy()
    fx.b_cal(at)
    import oracle_py
    ks = oracle_py.KSpace.from_system(oracle, s)
    sr_o, si_o = ks.sincos_b(at.x, q0, at.echeck, at.nlocal)
    ks.close()
    sr_g, si_g = fx.sfac()
    scale = max(np.abs(sr_o).max(), np.abs(si_o).max())
    assert max(np.abs(sr_g - sr_o).max(), np.abs(si_g - si_o).max()) / scale < 1e-11
    capi.load_library().conp_debug_set_sk_workgroups(0)
    fy = FixConp(s)                                  # default schedule (32 workgroups, one-level sum)
    fy.init_lists(alist, blist)
    fy.setup_post_neighbor(at)
    fy.b_cal(at)
    assert rel_err(fx.vectors()[0], fy.vectors()[0]) < 1e-12
    fx.close(); fy.close()


@pytest.mark.parametrize("deck,mode,Q", [
    ("cond", "slab", 0.35),       # tests/cond/input N = 0 (conp, dv 2) and N = 1 (conq, Q 0.35)
    ("cond", "ffield", 0.35),     # N = 2 (conp ffield), 3 (conq ffield), 4 (cond ffield)
    ("cond2", "ffield", 50.0),    # tests/cond2/input: rough 2 x 1248-atom electrodes, Q = 50
])
def test_cond_decks_all_three_fix_styles(deck, mode, Q):
    """the reference's tests/cond and tests/cond2 decks: one A matrix, then the conp / conq / cond charge rules on it"""
    import oracle_py
    lib = oracle_py.load(fast=True)                  # OpenMP build: the A matrix of cond2 (Ne 2496) takes a minute otherwise
    s = systems.deck(deck, mode, etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    o = OracleRun(lib, s, at, alist, blist)
    o.fx.lib.orc_fix_a_cal(o.fx.h); o.fx.lib.orc_fix_b_setq_cal(o.fx.h)
    setz = o.fx.vectors()[0] / systems.EVSCALE       # FixCond::cond_setup (fix_cond.cpp:46-56)
    assert o.fx.lib.orc_fix_equation_solve(o.fx.h) == 0
    o.fx.lib.orc_fix_get_setq(o.fx.h)
    loc = slice(0, at.nlocal)
    ele = at.echeck[loc] != 0
    q_start = at.q.copy()
    styles = ["conp", "conq"] + (["cond"] if mode == "ffield" else [])
    for style in styles:
        at.q[:] = q_start; o.q[:] = q_start
        arg = 2.0 if style == "conp" else Q
        if style == "conp":
            o.pre_force(arg); want = o.fx.scalars()["scalar_output"]
        elif style == "conq":
            want = o.fx.pre_force_conq(arg)
        else:
            want = o.fx.pre_force_cond(arg, setz)
        fx = FixConp(s, style=style)
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        fx.setup_pre_force(at, 0, arg)
        assert fx.info().elenum_all == (832 if deck == "cond" else 2496)
        assert rel_err(fx.vectors()[0], o.fx.vectors()[0]) < 1e-10, style
        assert rel_err(at.q[loc][ele], o.q[loc][ele]) < 1e-7, style
        assert fx.compute_scalar() == pytest.approx(want, rel=1e-7, abs=1e-10), style
        if style == "conq":                          # prescribed charge on group 2 (fix_conq.cpp:63-76)
            assert at.q[loc][at.echeck[loc] == -1].sum() == pytest.approx(Q, rel=1e-8)
        fx.close()
    o.fx.close()


def test_zmirror_deck_conp_and_conq():
    """tests/zmirror/input N = 0 (conp 2 V) and N = 3 (conq 0.7): doubled, mirrored il cell with `noslab zneutr`"""
    import oracle_py
    lib = oracle_py.load(fast=True)
    s = systems.deck("il_onelayer", "zmirror", etypes=True)
    assert s.zneutr and s.ff_flag == 2
    at, alist, blist = neighbor.build_lists(s)
    o = OracleRun(lib, s, at, alist, blist)
    o.setup()
    loc = slice(0, at.nlocal)
    ele = at.echeck[loc] != 0
    upper = at.x[loc][:, 2] > 0
    q_start = at.q.copy()
    for style, arg in (("conp", 2.0), ("conq", 0.7)):
        at.q[:] = q_start; o.q[:] = q_start
        if style == "conp":
            o.pre_force(arg); want = o.fx.scalars()["scalar_output"]
        else:
            want = o.fx.pre_force_conq(arg)
        fx = FixConp(s, style=style)
        assert fx.args.zneutr == 1 and fx.args.ff_flag == 2
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        fx.setup_pre_force(at, 0, arg)
        assert fx.info().elenum_all == 1664
        assert rel_err(fx.vectors()[0], o.fx.vectors()[0]) < 1e-10, style
        assert rel_err(at.q[loc][ele], o.q[loc][ele]) < 1e-7, style
        assert fx.compute_scalar() == pytest.approx(want, rel=1e-7, abs=1e-10), style
        # zneutr (fix_conp.cpp:1027-1060): each half of the doubled cell is neutral on its own
        qe = at.q[loc][ele]
        assert abs(qe[upper[ele]].sum()) < 1e-9 and abs(qe[~upper[ele]].sum()) < 1e-9, style
        fx.close()
    o.fx.close()


@pytest.fixture(scope="module")
def headline():
    s = systems.synthetic_fast()          # 4096 electrode / 32768 electrolyte, ffield (bench.py's workload)
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.linalg_setup(at)
    yield s, at, alist, blist, fx
    fx.close()


def test_headline_b_vector_matches_oracle(oracle, headline):
    """k-space + real-space b at the full benchmark size against the oracle's sincos_b / bbb / blist loops"""
    import oracle_py
    s, at, alist, blist, fx = headline
    fx.b_cal(at)
    b_g, _, _ = fx.vectors()
    o = OracleRun(oracle, s, at, alist, blist)          # post_neighbor only: no A matrix needed for b
    m = o.fx.maps()
    loc = {int(t): i for i, t in enumerate(at.tag[:at.nlocal])}
    xele = np.array([at.x[loc[int(t)]] for t in m["eleall2tag"]])
    fast = oracle_py.load(fast=True); fast.orc_set_threads(16)    # same arithmetic, OpenMP over k rows
    ks = oracle_py.KSpace.from_system(fast, s)
    sr, si = ks.sincos_b(at.x, at.q, at.echeck, at.nlocal)
    csk, snk = ks.ele_trig(xele)
    b_o = ks.bbb(csk, snk, sr, si)
    breal = np.zeros(len(b_o)); breal[m["ele2eleall"]] = o.fx.blist_only()
    b_o += breal
    sr_g, si_g = fx.sfac()
    scale = max(np.abs(sr).max(), np.abs(si).max())
    assert max(np.abs(sr_g - sr).max(), np.abs(si_g - si).max()) / scale < 1e-11
    assert rel_err(b_g, b_o) < 1e-10
    info = fx.info()
    assert (info.elenum_all, info.n_elyte_charged) == (4096, 32768) and info.kcount == ks.kcount
    ks.close(); o.fx.close()


def test_headline_charges_are_the_product_of_the_projected_inverse_with_b(headline):
    """Ne = 4096: the fused solve multiplies with the projected inverse taken as a SYMMETRIC matrix (packed lower-triangle tiles,
    half the bytes).  q_ele must equal S b + dV S d computed on the host from the full matrix the library hands out -- to the
    asymmetry of the computed S (~1e-16 relative) times the condition of the sum -- and so must the row-by-row product
    (the comparison is against numpy, not a second handle)."""
    import torch
    s, at, alist, blist, fx = headline
    S = fx.matrix()
    assert np.abs(S - S.T).max() <= 1e-12 * np.abs(S).max()
    d_x = torch.from_numpy(np.ascontiguousarray(at.x)).cuda(); d_q = torch.from_numpy(at.q.copy()).cuda()
    fx.pre_force_device(d_x.data_ptr(), d_q.data_ptr(), s.potdiff)
    torch.cuda.synchronize()
    b, y, setq = fx.vectors()
    want = S @ b
    assert np.abs(y - want).max() <= 1e-11 * np.abs(want).max()
    q = d_q.cpu().numpy()
    m = fx.maps()
    loc = {int(t): i for i, t in enumerate(at.tag[:at.nlocal])}
    qe = np.array([q[loc[int(t)]] for t in m["eleall2tag"]])
    assert np.abs(qe - (want + s.potdiff * setq)).max() <= 1e-11 * np.abs(qe).max()
    assert abs(qe.sum()) < 1e-9


def test_headline_inverse_really_inverts(headline):
    """Ne = 4096: the matrix the GEMV streams IS the projected inverse of the Ewald matrix, not merely symmetric with zero row
    sums.  A is rebuilt by a second handle (a_cal only, before any inverse); with P = A^-1 e e^T / (e^T A^-1 e) the projection
    gives S A = I - P, hence  S (A v) = v  for every v with sum(v) = 0, and  S A e' = e' - A^-1 e (e^T e')/(e^T A^-1 e).
    Checked with random probe vectors (O(n^2) each) and, through conp_invert on the same A, the plain inverse:  inv (A v) = v."""
    s, at, alist, blist, fx = headline
    S = fx.matrix()
    fa = FixConp(s)
    fa.init_lists(alist, blist)
    fa.setup_post_neighbor(at)
    fa.a_cal(at)
    A = fa.matrix()
    n = A.shape[0]
    assert np.array_equal(A, A.T) and np.all(np.diag(A) > 0)
    rng = np.random.default_rng(11)
    V = rng.normal(size=(n, 6))
    V0 = V - V.mean(axis=0)                                  # zero-sum probes
    assert np.abs(S @ (A @ V0) - V0).max() < 1e-8 * np.abs(V0).max()
    inv = fa.invert(A)                                       # the blocked Gauss-Jordan on the device, Ne = 4096
    assert np.abs(inv @ (A @ V) - V).max() < 1e-8 * np.abs(V).max()
    ainve = inv.sum(axis=1)
    Sref_V = inv @ V - np.outer(ainve, ainve @ V) / ainve.sum()      # fix_conp.cpp:1011-1020 applied to the probes
    assert np.abs(S @ V - Sref_V).max() < 1e-8 * np.abs(Sref_V).max()
    fa.close()


def test_headline_properties(headline):
    s, at, alist, blist, fx = headline
    S = fx.matrix()
    n = S.shape[0]
    # projected inverse: symmetric, S e = 0 (electroneutrality for any b)
    assert np.abs(S - S.T).max() / np.abs(S).max() < 1e-9
    assert np.abs(S.sum(axis=1)).max() / np.abs(S).max() < 1e-9
    # one update: charges neutral, scalar consistent, electrolyte untouched
    q_before = at.q.copy()
    fx.pre_force(at, 0, s.potdiff)
    ele = at.echeck != 0
    assert abs(at.q[:at.nlocal][ele[:at.nlocal]].sum()) < 1e-9
    assert np.array_equal(at.q[~ele], q_before[~ele])
    b, q, setq = fx.vectors()
    assert rel_err(S @ b, q) < 1e-10                       # the GEMV really is S b
    left = fx.maps()["elecheck_eleall"] == 1
    assert fx.compute_scalar() == pytest.approx(s.potdiff * setq[left].sum() + q[left].sum(), rel=1e-9)
    # linearity in the electrolyte charges: b(2 q) = 2 b(q)  (exact in binary floating point up to the last bits)
    at2_q = at.q.copy(); at2_q[~ele] *= 2.0
    at2 = neighbor.Atoms(nlocal=at.nlocal, nghost=at.nghost, x=at.x, q=at2_q, type=at.type, tag=at.tag, echeck=at.echeck, owner=at.owner)
    fx.b_cal(at2)
    b2, _, _ = fx.vectors()
    assert rel_err(b2, 2.0 * b) < 1e-13
    # idempotence: the same inputs give bitwise the same b (fixed summation trees, no atomics)
    fx.b_cal(at2)
    b3, _, _ = fx.vectors()
    assert np.array_equal(b2, b3)


def test_large_box_structure_factors_against_direct_sums():
    """BASELINE configs[4] geometry (16384 electrode / 262144 electrolyte atoms, K ~ 1.0e6, three kz column tiles): a sample of
    structure factors and k-space b entries against direct numpy sums  S(k) = sum_j q_j exp(i k.r_j),
    b_i = - sum_k 2 ug_k Re(conj(e^{i k.r_i}) S_k)  -- independent of the oracle and of the (planar, kz) factorisation"""
    s = systems.synthetic_fast(n_cells_x=64, n_cells_y=32, lz=1200.0, n_elyte=262144, cutoff=12.0, accuracy_relative=1e-6,
                               g_ewald=0.2554)
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    info = fx.info()
    assert info.elenum_all == 16384 and info.n_elyte_charged == 262144 and info.kcount > 1_000_000
    b_g = fx.km_b_cal(at)
    sr, si = fx.sfac()
    kt = fx.ktables()
    unitk = np.array(list(info.unitk))
    sol = (at.echeck[:at.nlocal] == 0) & (at.q[:at.nlocal] != 0)
    xs, qs = at.x[:at.nlocal][sol], at.q[:at.nlocal][sol]
    rng = np.random.default_rng(5)
    pick = np.concatenate([rng.integers(0, info.kcount, 60), [0, info.kcount_flat, info.kcount - 1]])
    kv = np.stack([kt["kxvecs"][pick], kt["kyvecs"][pick], kt["kzvecs"][pick]], 1) * unitk
    ph = xs @ kv.T
    S_ref = (qs[:, None] * np.exp(1j * ph)).sum(0)
    scale = np.abs(sr).max()
    assert np.abs(sr[pick] - S_ref.real).max() / scale < 1e-10
    assert np.abs(si[pick] - S_ref.imag).max() / scale < 1e-10
    # b for a few electrode atoms from ALL structure factors the GPU produced
    m = fx.maps()
    loc = {int(t): i for i, t in enumerate(at.tag[:at.nlocal])}
    kall = np.stack([kt["kxvecs"], kt["kyvecs"], kt["kzvecs"]], 1) * unitk
    for ia in (0, 5000, 16383):
        r = at.x[loc[int(m["eleall2tag"][ia])]]
        e = np.exp(1j * (kall @ r))
        b_ref = -np.sum(2.0 * kt["ug"] * (e.real * sr + e.imag * si))
        assert b_g[ia] == pytest.approx(b_ref, rel=1e-9)
    fx.close()


@pytest.mark.parametrize("mode", ["ffield", "slab"])
def test_small_system_fused_phase_equals_the_separate_launch(monkeypatch, mode):
    """il_onelayer takes the small-system form of sk_gemm (SkFuse: the phase tables of a segment's atoms and the pair sums inside the
    launch, no elyte_phase launch).  The tables and the pair sums are the stand-alone kernels' arithmetic value for value: the b
    vector and the charges must come out bit for bit as with the stand-alone phase launch, CONP_PATH_PHASE_LAUNCH (ffield); in slab mode the slab sum is added in
    another order (per segment, not per block of the phase kernel): equal to rounding."""
    s = systems.deck("il_onelayer", mode, etypes=True, shuffle_seed=5)
    at0, alist, blist = neighbor.build_lists(s)
    out = {}
    for sep in (False, True):
        if sep:
            capi.load_library().conp_debug_set_paths(capi.PATH_PHASE_LAUNCH)
        at, _, _ = neighbor.build_lists(s)
        fx = FixConp(s)
        fx.init_lists(alist, blist)
        fx.setup_post_neighbor(at)
        fx.setup_pre_force(at, 0, s.potdiff)
        fx.profile(1)
        fx.pre_force(at, 1, s.potdiff)
        names = set(fx.profile_read())
        fx.profile(0)
        b, q, _ = fx.vectors()
        out[sep] = (b.copy(), q.copy(), at.q.copy(), names)
        fx.close()
        if sep:
            capi.load_library().conp_debug_set_paths(0)
    assert "elyte_phase" not in out[False][3] and "elyte_phase" in out[True][3]
    if mode == "ffield":
        for k in range(3):
            assert np.array_equal(out[False][k], out[True][k])
    else:
        for k in range(3):
            assert rel_err(out[False][k], out[True][k]) < 1e-13


def test_one_launch_tail_with_the_in_launch_hand_off_gives_the_bits_of_the_two_launches(headline):
    """round 5 (review item 1): the pieces' sums, the real-space pair sums and the per-atom dot as ONE launch -- a few workgroups add
    the pieces and hand the class table to the dot workgroups through write-through stores and a ticket, no fence
    (CONP_PATH_HC_FUSED).  Measured slower than hc_sum + b_zc_final (profiles/r05_tail_handoff_ab.txt), so not the default; the
    results are the default path's bit for bit, also when every dot workgroup's bounded wait runs out and it adds the pieces itself
    (CONP_PATH_HC_NO_WAIT)."""
    s, at, alist, blist, fx = headline
    fx.b_cal(at)
    b0 = fx.vectors()[0].copy()
    for mask in (capi.PATH_HC_FUSED, capi.PATH_HC_FUSED | capi.PATH_HC_NO_WAIT):
        with capi.test_paths(mask):
            fy = FixConp(s)
            fy.init_lists(alist, blist)
            fy.setup_post_neighbor(at)
            fy.b_cal(at)
            b1 = fy.vectors()[0].copy()
            fy.b_cal(at)
            assert np.array_equal(b1, fy.vectors()[0])
            fy.close()
        assert np.array_equal(b0, b1)
