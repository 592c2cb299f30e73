"""shared drivers for the parity tests: the same System goes through the CPU oracle and through the HIP library"""
import numpy as np

from conp_amd import neighbor, systems
import oracle_py


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


class OracleRun:
    def __init__(self, lib, s, at, alist, blist, **kw):
        self.at = at
        self.q = at.q.copy()
        # the oracle writes charges into its own copy of q
        self.at_o = neighbor.Atoms(nlocal=at.nlocal, nghost=at.nghost, x=at.x, q=self.q, type=at.type, tag=at.tag,
                                   echeck=at.echeck, owner=at.owner)
        self.fx = oracle_py.Fix(lib, s, **kw)
        self.fx.set_atoms(self.at_o)
        self.fx.set_lists(alist, blist)
        self.fx.post_neighbor()

    def setup(self):
        assert self.fx.linalg_setup() == 0

    def pre_force(self, potdiff):
        self.fx.pre_force(potdiff)


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    den = max(np.max(np.abs(b)), 1e-300)
    return float(np.max(np.abs(a - b)) / den)
