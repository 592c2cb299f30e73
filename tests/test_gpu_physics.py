"""-m gpu: the constant-potential property itself, checked against a brute-force Ewald sum written from the definitions
(complex exponentials over the k list, explicit periodic images) -- not against the oracle's restated loops.

After one charge update the electrostatic potential at every electrode atom, from ALL charges (electrolyte + the new
electrode charges, Gaussian width eta, fix_conp.cpp:1446-1475, km_ewald.cpp:584-666, 789-847), must equal the applied value
dV * d_i (the preset vector of b_setq_cal, fix_conp.cpp:609-637) up to ONE constant for all atoms -- the Lagrange multiplier
of the electroneutrality projection (:982-1067).  Only the k list / ug weights are taken from the library; they are pinned
bit-exact against the oracle elsewhere (tests/test_host_logic.py)."""
import numpy as np
import pytest

from conp_amd import FixConp, neighbor, systems

pytestmark = pytest.mark.gpu


def erfcr(a2r2):
    """erfc(sqrt(x))/sqrt(x) with the reference's 5-term polynomial and its 5.8 cut (fix_conp.cpp:53-60, 1446-1454)"""
    a2r2 = np.asarray(a2r2, dtype=float)
    ar = np.sqrt(a2r2)
    t = 1.0 / (1.0 + 0.3275911 * ar)
    poly = t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429))))
    with np.errstate(divide="ignore", invalid="ignore"):
        v = poly * np.exp(-a2r2) / ar
    return np.where(a2r2 < 5.8 * 5.8, v, 0.0)


def potential_at_electrodes(s, x, q, echeck, kt, unitk, volume):
    """brute force, e/Angstrom: row i = electrode atom i (order of np.nonzero(echeck))"""
    g, eta, rc2 = s.g_ewald, s.eta, s.cutoff ** 2
    rc2 = min(rc2, (5.8 / g) ** 2)                                   # fix_conp.cpp:1302-1305
    ele = np.nonzero(echeck != 0)[0]
    kv = np.stack([kt["kxvecs"] * unitk[0], kt["kyvecs"] * unitk[1], kt["kzvecs"] * unitk[2]], axis=1)
    charged = np.nonzero(q != 0)[0]
    S = np.exp(1j * x[charged] @ kv.T).T @ q[charged]                 # S_k over every charged atom
    phi = (2.0 * kt["ug"] * (np.exp(-1j * x[ele] @ kv.T) * S).real).sum(axis=1)      # includes ug_tot * q_i
    phi += q[ele] * (np.sqrt(2.0) * eta - 2.0 * g) / np.sqrt(np.pi)  # Gaussian self term, Ewald self correction
    if s.slabflag:
        phi += 4.0 * np.pi / volume * x[ele, 2] * (q * x[:, 2]).sum()  # EW3DC dipole term (km_ewald.cpp:647-665, 827-847)
    shifts = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in ((-1, 0, 1) if s.periodic[2] else (0,))]
    prd = s.prd
    for sh in shifts:
        d = x[ele][:, None, :] - (x[None, :, :] + np.array(sh) * prd)
        r2 = (d * d).sum(axis=2)
        ok = (r2 < rc2) & (r2 > 1e-12)
        r2s = np.where(ok, r2, 1.0)
        is_ele = (echeck != 0)[None, :]
        eta2 = np.where(is_ele, eta * eta / 2.0, eta * eta)          # ele-ele: eta/sqrt(2); ele-electrolyte: eta
        etaf = np.where(is_ele, eta / np.sqrt(2.0), eta)
        f = erfcr(g * g * r2s) * g - erfcr(eta2 * r2s) * etaf
        phi += (np.where(ok, f, 0.0) * q[None, :]).sum(axis=1)
    return ele, phi


@pytest.mark.parametrize("system,mode,solver", [("small", "slab", "inv"), ("small", "ffield", "inv"), ("dilute", "ffield", "inv"),
                                                ("dilute", "slab", "inv"), ("il_onelayer", "slab", "inv"),
                                                ("small", "slab", "cg"), ("dilute", "ffield", "cg")])
def test_electrode_potential_is_the_applied_one(system, mode, solver):
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0, mode=mode) if system == "small" else systems.deck(system, mode)
    at, alist, blist = neighbor.build_lists(s)
    # CG stops at (r.p)/Ne < tol (fix_conp.cpp:917): tighten it so that the residual potential is far below the check's bound
    # (not absurdly: once the residual is exactly zero the reference's recurrence divides 0 by 0, fix_conp.cpp:895-899)
    fx = FixConp(s, extra_args=["cg", "maxiter", "400", "tol", "1e-16"] if solver == "cg" else [])
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    dv = 1.7
    fx.setup_pre_force(at, 0, dv)
    info = fx.info()
    n = at.nlocal
    x, q, ec = at.x[:n], at.q[:n], at.echeck[:n]
    ele, phi = potential_at_electrodes(s, x, q, ec, fx.ktables(), np.array(info.unitk), info.volume)
    ev = systems.EVSCALE
    z = x[ele, 2]
    if mode == "ffield":
        zhalf = s.boxlo[2] + 0.5 * s.prd[2]
        d = np.where((ec[ele] == 1) & (z < zhalf), -ev * (z / s.prd[2] + 1.0), -ev * z / s.prd[2])
    else:
        d = -0.5 * ev * ec[ele]
    resid = phi - dv * d
    spread = resid.max() - resid.min()
    # (CG: the stop criterion (r.p)/Ne < 1e-16 leaves a residual potential of order sqrt(Ne * 1e-16))
    assert spread < (1e-9 if solver == "inv" else 1e-6) * dv * ev, (spread, dv * ev)
    assert abs(q[ele].sum()) < 1e-12                                  # and the electrodes are neutral together
    fx.close()
