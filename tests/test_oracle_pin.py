"""Pins the CPU oracle to the only known-answer vector the reference's tests hold for this path:
tests/dilute/persist.log (G vector, step-0 electrode charges; ffield etypes, dV = 1 V)."""
import json
import os

import numpy as np
import pytest

from conp_amd import neighbor, systems
import oracle_py

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def run_oracle(lib, s, **kw):
    at, alist, blist = neighbor.build_lists(s)
    fx = oracle_py.Fix(lib, s, **kw)
    fx.set_atoms(at)
    fx.set_lists(alist, blist)
    fx.post_neighbor()
    info = fx.linalg_setup()
    assert info == 0
    fx.pre_force(s.potdiff)
    return fx, at


def test_dilute_step0_charge_matches_persist_log(oracle):
    gold = json.load(open(os.path.join(GOLD, "dilute_persist.json")))
    s = systems.deck("dilute", "ffield", etypes=True, g_ewald=gold["g_ewald"])
    fx, at = run_oracle(oracle, s)
    qleft = at.q[:at.nlocal][at.echeck[:at.nlocal] == 1].sum()
    qright = at.q[:at.nlocal][at.echeck[:at.nlocal] == -1].sum()
    step0 = gold["thermo"][0]
    print("oracle qleft", qleft, "persist.log", step0[3])
    # the log prints 8 significant digits
    assert qleft == pytest.approx(step0[3], rel=5e-7)
    assert qright == pytest.approx(step0[4], rel=5e-7)
    assert abs(qleft + qright) < 1e-14
    fx.close()
