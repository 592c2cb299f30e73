"""-m gpu: the C++ LAMMPS glue (FixConpHip) EXECUTED -- not only type-checked -- by lammps_glue/glue_driver, a miniature
host on top of the interface mock that calls the hooks in LAMMPS' order (constructor, fix_modify, init, init_list,
setup_post_neighbor, setup_pre_force, [post_neighbor], pre_force, post_force, end_of_step, compute_scalar).
The numbers must equal what the ctypes harness gets from the same library on the same inputs (bitwise: same kernels), the log
file must carry the reference's lines, and error->all must carry the reference's messages."""
import os
import subprocess

import numpy as np
import pytest

from conp_amd import FixConp, neighbor, systems
from conp_amd.capi import fix_command_for
from helpers import rel_err

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "lammps-user-conp2_amd", "lammps_glue", "glue_driver")
BIT = {"all": 1, "eleleft": 2, "eleright": 4}


def write_case(path, s, at, lists, tokens, steps, variable=("-", 0.0), modify=(), mesh=(0, 0, 0, 0)):
    out = []
    w = lambda *a: out.append(" ".join(repr(float(v)) if isinstance(v, (float, np.floating)) else str(v) for v in a))
    w(s.ntypes, at.nlocal, at.nghost, s.natoms)
    w(*[float(v) for v in s.prd], *[float(v) for v in s.boxlo])
    w(float(s.g_ewald), float(s.accuracy), float(s.slab_volfactor), int(s.slabflag), *mesh)
    w(systems.QQRD2E, systems.QQR2E, systems.QE2F, 1.0, int(s.newton), float(s.cutoff))
    w(*[float(v) for v in s.cutsq_table().ravel()])
    w(len(BIT)); [w(k, v) for k, v in BIT.items()]
    for i in range(at.nall):
        mask = 1 | (2 if at.echeck[i] == 1 else 0) | (4 if at.echeck[i] == -1 else 0)
        w(int(at.tag[i]), int(at.type[i]), mask, float(at.q[i]), *[float(v) for v in at.x[i]])
    w(len(lists))
    for L in lists:
        w(L.inum)
        for i in L.ilist[:L.inum]:
            n = int(L.numneigh[i]); f = int(L.first[i])
            w(int(i), n, *[int(j) for j in L.neigh[f:f + n]])
    w(variable[0], float(variable[1]))
    w(len(modify)); [w(len(m), *m) for m in modify]
    w(len(tokens), *tokens)
    w(len(steps))
    for (ts, pd, reneigh, x) in steps:
        w(ts, float(pd), int(reneigh), 0 if x is None else 1)
        if x is not None:
            w(*[float(v) for v in np.asarray(x).ravel()])
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")


ENV_EXTRA = None


def run_driver(case, cwd, *extra):
    """case: one case file, or a list of them = one per rank thread (`glue_driver ranks N ...`)"""
    if not os.path.exists(DRIVER):            # normally built by __graft_entry__.build(); g++ is on the GPU box too
        subprocess.check_call(["make", "-C", os.path.dirname(DRIVER), "driver"])
    cmd = [DRIVER, case, *extra] if isinstance(case, str) else [DRIVER, "ranks", str(len(case)), *case, *extra]
    p = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=300, env=dict(os.environ, **(ENV_EXTRA or {})))
    res = {"scalar": {}, "q": {}, "f": {}, "error": None, "screen": [], "rc": p.returncode, "scalar_by_rank": {}}
    for line in p.stdout.splitlines():
        rank = 0
        if line.startswith("rank "):                 # several ranks: "rank R <line>"; the charges of all ranks form one table
            _, r, line = line.split(" ", 2)
            rank = int(r)
        t = line.split()
        if line.startswith("scalar "):
            res["scalar_by_rank"].setdefault(rank, {})[int(t[1])] = float(t[2])
        if line.startswith("scalar "):
            res["scalar"][int(t[1])] = float(t[2])
        elif line.startswith("q "):
            res["q"].setdefault(int(t[1]), {})[int(t[2])] = float(t[3])
        elif line.startswith("f "):
            res["f"][int(t[1])] = [float(v) for v in t[2:6]] + [float(t[7]), float(t[9])]
        elif line.startswith("ERROR: "):
            res["error"] = line[len("ERROR: "):]
        else:
            res["screen"].append(line)
    return res, p


@pytest.mark.parametrize("etypes", [True, False])
def test_hook_sequence_gives_the_same_charges_as_the_ctypes_path(tmp_path, etypes):
    s = systems.deck("dilute", "ffield", etypes=etypes)
    at, alist, blist = neighbor.build_lists(s)
    rng = np.random.default_rng(2)
    sol = at.echeck == 0
    frames = []
    x = at.x.copy()
    for step in (1, 2, 3):
        x = x.copy(); x[sol] += rng.normal(scale=0.02, size=(int(sol.sum()), 3))
        frames.append(x)
    steps = [(0, s.potdiff, 0, None), (1, s.potdiff, 0, frames[0]), (2, 0.5, 1, frames[1]), (3, 0.5, 0, frames[2])]
    tokens = fix_command_for(s)
    tokens[6] = "v_dv"                                   # the potential difference as an equal-style variable (:266-272, :1143)
    lists = [alist] if alist is blist else [alist, blist]
    case = str(tmp_path / "case.txt")
    write_case(case, s, at, lists, tokens, steps, variable=("dv", s.potdiff))
    res, proc = run_driver(case, str(tmp_path))
    assert res["rc"] == 0 and res["error"] is None, proc.stdout[-2000:] + proc.stderr[-2000:]

    fx = FixConp(s)                                      # same library through ctypes
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    ele = np.nonzero(at.echeck[:at.nlocal] != 0)[0]
    def check(ts):
        for i in ele:
            assert res["q"][ts][int(at.tag[i])] == at.q[i], (ts, i)
        assert res["scalar"][ts] == fx.compute_scalar()
    check(0)
    for ts, pd, reneigh, xs in steps[1:]:
        at.x[:] = xs
        if reneigh:
            fx.post_neighbor(at)
        fx.pre_force(at, ts, pd)
        check(ts)
    fx.close()
    # the reference's screen and log-file lines (fix_conp.cpp:458-461, 1006-1009, 787, 857, 564-566)
    screen = "\n".join(res["screen"])
    assert "conp output: <e,e> = " in screen and "conp output: <d,d> = " in screen
    log = open(tmp_path / "log_conp").read().splitlines()
    assert log[0] == "A matrix calculating ..." and log[1].startswith("A matrix calculation time  = ")
    assert [l.split("=")[0] for l in log[2:]] == ["B vector calculation time ", "Coulomb calculation time ", "Kspace calculation time "]


def test_post_force_through_the_glue_matches_the_ctypes_path(tmp_path):
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0, mode="slab")
    rng = np.random.default_rng(9)
    ele_idx = np.nonzero(s.echeck != 0)[0]; sol_idx = np.nonzero((s.echeck == 0) & (s.q != 0))[0]
    for k in range(6):                                    # a few electrolyte atoms inside the Gaussian overlap range
        d = rng.normal(size=3); d *= rng.uniform(0.4, 1.1) / np.linalg.norm(d)
        s.x[sol_idx[k]] = s.x[ele_idx[3 * k]] + d
    s.x[:, :2] = s.boxlo[:2] + np.mod(s.x[:, :2] - s.boxlo[:2], s.prd[:2])
    at, alist, blist = neighbor.build_lists(s)
    lists = [alist] if alist is blist else [alist, blist]
    case = str(tmp_path / "case.txt")
    write_case(case, s, at, lists, fix_command_for(s), [(0, s.potdiff, 0, None)])
    res, proc = run_driver(case, str(tmp_path))
    assert res["rc"] == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    fx = FixConp(s)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    f, ek, ec, vir = fx.post_force(at)
    f = f.reshape(-1, 3)[:at.nlocal]
    got = res["f"][0]
    assert np.abs(f).sum() > 0
    assert np.allclose(got[:3], f.sum(axis=0), rtol=0, atol=1e-9 * np.abs(f).sum())
    assert got[3] == pytest.approx(np.abs(f).sum(), rel=1e-12)
    assert got[4] == pytest.approx(ec, rel=1e-12)         # what Pair::ev_tally would have accumulated
    fx.close()


@pytest.mark.parametrize("tokens,modify,message", [
    (["e", "eleleft", "conp", "1", "eleright", "1.979", "1.0", "log_conp", "bogus"], (), "unknown option"),
    (["e", "eleleft", "conp", "1", "nogroup", "1.979", "1.0", "log_conp"], (), "Fix conp group ID does not exist"),
    (["e", "eleleft", "conp", "1", "eleright", "1.979", "v_nope", "log_conp"], (), "potential difference variable does not exist"),
    (["e", "eleleft", "conp", "1", "eleright", "1.979", "1.0", "log_conp", "pppm"], (), "pppm"),
    (["e", "eleleft", "conp", "1", "eleright", "1.979", "1.0", "log_conp"], (["ehgo", "kappa", "1.0"],), "basic pair mode"),
])
def test_errors_reach_error_all_with_the_reference_messages(tmp_path, tokens, modify, message):
    s = systems.small_random(ne_side=4, n_elyte=32, lz=60.0)
    at, alist, blist = neighbor.build_lists(s)
    case = str(tmp_path / "case.txt")
    write_case(case, s, at, [alist] if alist is blist else [alist, blist], tokens, [(0, 1.0, 0, None)], modify=modify)
    res, proc = run_driver(case, str(tmp_path))
    assert res["rc"] == 2 and res["error"] is not None, proc.stdout[-1000:]
    assert message in res["error"], res["error"]


@pytest.mark.parametrize("mode", ["slab", "ffield"])
def test_kspace_provider_class_executed(tmp_path, mode):
    """INTEGRATION.md mode B: KSpaceModuleHip::{conp_setup, a_cal, b_cal} called like FixConp calls its provider
    (aaa[elenum][elenum_all] accumulated, bbb[elenum] overwritten, local electrode order) == the C-ABI provider calls"""
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0, mode=mode)
    at, alist, blist = neighbor.build_lists(s)
    case = str(tmp_path / "case.txt")
    write_case(case, s, at, [alist] if alist is blist else [alist, blist], fix_command_for(s), [(0, s.potdiff, 0, None)])
    res, proc = run_driver(case, str(tmp_path), "provider")
    assert res["rc"] == 0 and res["error"] is None, proc.stdout[-2000:] + proc.stderr[-2000:]
    a_loc, b_loc, e2ea = {}, {}, {}
    for line in proc.stdout.splitlines():
        t = line.split()
        if t[0] == "a":
            a_loc[(int(t[1]), int(t[2]))] = float(t[3])
        elif t[0] == "b":
            b_loc[int(t[1])] = float(t[2])
        elif t[0] == "m":
            e2ea[int(t[1])] = int(t[2])
    fx = FixConp(s)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    n = at.nlocal
    fx.km_conp_setup(float((at.q[:n] ** 2).sum()), n)
    a_full = fx.km_a_cal(at)
    b_all = fx.km_b_cal(at)
    ne = fx.info().elenum_all
    assert len(b_loc) == fx.info().elenum and len(a_loc) == len(b_loc) * ne
    assert np.array_equal(np.array([e2ea[i] for i in range(len(e2ea))]), fx.maps()["ele2eleall"])
    for i in range(len(b_loc)):
        assert b_loc[i] == b_all[e2ea[i]]
        assert np.array_equal(np.array([a_loc[(i, j)] for j in range(ne)]), a_full[e2ea[i]])
    fx.close()


def test_pppm_keyword_takes_the_mesh_from_the_kspace_style(tmp_path):
    """`pppm`: FixConpHip reads nx_pppm / ny_pppm / nz_pppm / order from force->kspace (the public KSpace members a PPPM style
    fills) -- same charges as the ctypes path given the same mesh"""
    s = systems.deck("dilute", "ffield", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    case = str(tmp_path / "case.txt")
    tokens = fix_command_for(s, extra=["pppm"])
    write_case(case, s, at, [alist, blist], tokens, [(0, s.potdiff, 0, None)], mesh=(27, 24, 144, 5))
    res, proc = run_driver(case, str(tmp_path))
    assert res["rc"] == 0 and res["error"] is None, proc.stdout[-2000:] + proc.stderr[-2000:]
    fx = FixConp(s, extra_args=["pppm"], pppm_mesh=(27, 24, 144), pppm_order=5)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    ele = np.nonzero(at.echeck[:at.nlocal] != 0)[0]
    got = np.array([res["q"][0][int(at.tag[i])] for i in ele])
    # the mesh spread uses f64 atomics: equal up to the order of the additions
    assert np.abs(got - at.q[ele]).max() <= 1e-12 * np.abs(at.q[ele]).max()
    fx.close()


@pytest.mark.parametrize("name,axis,nranks,pppm", [("small_slab", 0, 2, False), ("dilute_ffield_etypes", 2, 2, False),
                                                   ("small_ffield", 1, 3, False), ("dilute_ffield_etypes", 1, 2, True)])
def test_several_mpi_ranks_match_one_rank(tmp_path, name, axis, nranks, pppm):
    """FixConpHip on a spatially decomposed system (fix_conp.cpp:409, 641-648; km_ewald.cpp:782-786 are the reference's multi-rank
    sites): N rank threads, each with the owned atoms, ghosts and half lists of its slab, the collectives through the glue's
    MPI-backed conp_comm callbacks (here the MPI mock).  Charges per tag, and the fix scalar on every rank, equal the one-rank run.
    pppm: the `pppm` keyword under several ranks -- the mesh lives on rank 0, the ranks' electrolyte atoms are gathered per update."""
    mesh = (27, 24, 144, 5) if pppm else None
    kw = dict(extra_args=["pppm"], pppm_mesh=mesh[:3], pppm_order=mesh[3]) if pppm else {}
    s = {"small_slab": lambda: systems.small_random(ne_side=4, n_elyte=96, lz=60.0, mode="slab"),
         "small_ffield": lambda: systems.small_random(ne_side=4, n_elyte=96, lz=60.0),
         "dilute_ffield_etypes": lambda: systems.deck("dilute", "ffield", etypes=True)}[name]()
    rng = np.random.default_rng(5)
    disp = np.where((s.echeck == 0)[:, None], rng.normal(scale=0.02, size=s.x.shape), 0.0)     # per ATOM (by tag - 1)
    tag2row = {int(t): i for i, t in enumerate(s.tag)}

    def moved(at):
        return at.x + disp[[tag2row[int(t)] for t in at.tag]]

    # one rank, ctypes path
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s, **kw)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.setup_pre_force(at, 0, s.potdiff)
    ref = {0: ({int(t): q for t, q in zip(at.tag[:at.nlocal], at.q[:at.nlocal])}, fx.compute_scalar())}
    at.x[:] = moved(at)
    fx.pre_force(at, 1, 0.7)
    ref[1] = ({int(t): q for t, q in zip(at.tag[:at.nlocal], at.q[:at.nlocal])}, fx.compute_scalar())
    fx.post_neighbor(at)
    fx.pre_force(at, 2, 0.7)
    ref[2] = ({int(t): q for t, q in zip(at.tag[:at.nlocal], at.q[:at.nlocal])}, fx.compute_scalar())
    fx.close()

    parts = neighbor.build_lists_decomposed(s, nranks, axis=axis)
    assert all(p[0].nlocal > 0 for p in parts)
    cases = []
    for r, (atr, al, bl) in enumerate(parts):
        case = str(tmp_path / f"case{r}.txt")
        steps = [(0, s.potdiff, 0, None), (1, 0.7, 0, moved(atr)), (2, 0.7, 1, None)]
        tokens = fix_command_for(s, extra=["pppm"]) if pppm else fix_command_for(s)
        tokens[6] = "v_dv"                               # the potential difference changes from step to step: an equal-style variable
        write_case(case, s, atr, [al] if al is bl else [al, bl], tokens, steps, variable=("dv", s.potdiff),
                   **(dict(mesh=mesh) if pppm else {}))
        cases.append(case)
    res, proc = run_driver(cases, str(tmp_path))
    assert res["rc"] == 0 and res["error"] is None, proc.stdout[-3000:] + proc.stderr[-2000:]
    ele_tags = [int(t) for t, e in zip(s.tag, s.echeck) if e != 0]
    for ts in (0, 1, 2):
        qref, sc = ref[ts]
        scale = max(abs(qref[t]) for t in ele_tags)
        assert sorted(res["q"][ts]) == sorted(ele_tags)                       # every electrode atom reported by exactly its owner
        assert max(abs(res["q"][ts][t] - qref[t]) for t in ele_tags) < 1e-9 * scale, ts
        for r in range(nranks):
            assert res["scalar_by_rank"][r][ts] == pytest.approx(sc, rel=1e-9, abs=1e-12)


def test_pppm_conp_hip_kspace_style_executed(tmp_path):
    """`kspace_style pppm/conp/hip` (PPPMConpHip : PPPM, KSpaceModule): found by the dynamic_cast of fix_conp.cpp:402 when the fix
    carries the `pppm` keyword, then conp_setup / conp_post_neighbor / a_cal / b_cal, and the mesh potentials ComputePotentialAtom
    asks a provider for (compute_group_potential, compute_particle_potential) -- equal to the C-ABI calls through ctypes"""
    s = systems.deck("dilute", "ffield", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    mesh, order = (27, 24, 144), 5
    case = str(tmp_path / "case.txt")
    write_case(case, s, at, [alist, blist], fix_command_for(s, extra=["pppm"]), [(0, s.potdiff, 0, None)], mesh=(*mesh, order))
    res, proc = run_driver(case, str(tmp_path), "provider")
    assert res["rc"] == 0 and res["error"] is None, proc.stdout[-2000:] + proc.stderr[-2000:]
    b_loc, e2ea, u, up = {}, {}, {}, None
    for line in proc.stdout.splitlines():
        t = line.split()
        if t[0] == "b":
            b_loc[int(t[1])] = float(t[2])
        elif t[0] == "m":
            e2ea[int(t[1])] = int(t[2])
        elif t[0] == "u":
            u[int(t[1])] = float(t[2])
        elif t[0] == "up":
            up = (int(t[1]), float(t[2]))
    fx = FixConp(s, extra_args=["pppm"], pppm_mesh=mesh, pppm_order=order)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    b_all = fx.km_b_cal(at)
    scale = np.abs(b_all).max()
    assert len(b_loc) == fx.info().elenum
    for i in range(len(b_loc)):
        assert abs(b_loc[i] - b_all[e2ea[i]]) <= 1e-12 * scale          # f64 atomics in the spread: equal up to the order of additions
    sel = (at.echeck[:at.nlocal] == 1).astype(np.int32)
    want = fx.pppm_group_potential(at, sel)
    tags = at.tag[:at.nlocal]
    assert sorted(u) == sorted(int(t) for t in tags[sel != 0])
    for i in np.nonzero(sel)[0]:
        assert abs(u[int(tags[i])] - want[i]) <= 1e-12 * np.abs(want).max()
    i0 = int(np.nonzero(tags == up[0])[0][0])
    assert up[1] == pytest.approx(fx.pppm_particle_potential(at, i0), rel=1e-11)
    fx.close()


def test_pppm_compute_is_handed_the_cached_bricks_by_the_make_rho_override(tmp_path, oracle):
    """SURVEY 8f-2 / pppm_conp.cpp:428-450: after b_cal, LAMMPS' own PPPM::compute (the mock runs its first two virtual steps and the
    ghost sum) must be handed electrolyte brick + electrode brick by the provider's particle_map / make_rho overrides -- equal to the
    oracle's total brick --, the base class's own spread must not run, and the library must not spread the electrolyte a second
    time for it (the brick b_cal made for the step is kept: conp_info.pppm_elyte_spreads stands still across PPPM::compute);
    conp_pre_force of the next step drops the kept brick and b_cal comes out the same again."""
    global ENV_EXTRA
    import oracle_py
    s = systems.deck("dilute", "ffield", etypes=True)
    at, alist, blist = neighbor.build_lists(s)
    at.q[at.echeck == 1] = 0.013; at.q[at.echeck == -1] = -0.013      # (the deck's electrode atoms start uncharged: as if update_charge had run)
    mesh, order = (27, 24, 144), 5
    case = str(tmp_path / "case.txt")
    rho_path = str(tmp_path / "rho.bin")
    write_case(case, s, at, [alist, blist], fix_command_for(s, extra=["pppm"]), [(0, s.potdiff, 0, None)], mesh=(*mesh, order))
    ENV_EXTRA = {"GLUE_RHO_OUT": rho_path}
    try:
        res, proc = run_driver(case, str(tmp_path), "provider")
    finally:
        ENV_EXTRA = None
    assert res["rc"] == 0 and res["error"] is None, proc.stdout[-2000:] + proc.stderr[-2000:]
    calls = [ln.split() for ln in proc.stdout.splitlines() if ln.startswith("rho_calls ")]
    again = [ln.split() for ln in proc.stdout.splitlines() if ln.startswith("b_again ")]
    assert len(calls) == 1 and len(again) == 1
    base_map, base_rho, spreads_before, spreads_after = (int(v) for v in (calls[0][1], calls[0][2], calls[0][4], calls[0][5]))
    assert base_map == 0 and base_rho == 0                  # the base class never spread anything
    assert spreads_before >= 1 and spreads_after == spreads_before      # nor did the library, a second time
    rho = np.fromfile(rho_path, dtype=np.float64)
    assert rho.size == mesh[0] * mesh[1] * mesh[2]
    pp = oracle_py.Pppm(oracle, s, mesh, order)
    d_o, e_o, l_o = pp.make_rho(mesh, at.x, at.q, at.echeck, at.nlocal)
    pp.close()
    assert np.abs(e_o).max() > 0 and np.abs(l_o).max() > 0
    assert rel_err(rho, d_o.ravel()) < 1e-12
    assert float(again[0][1]) <= 1e-12 * 10.0               # the next step's b_cal: the same b (f64 atomics: to rounding)
