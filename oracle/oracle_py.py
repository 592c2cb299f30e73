"""ctypes binding of the CPU oracle (oracle/conp_oracle.c).  TEST INFRASTRUCTURE ONLY:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False):
    so = os.path.join(HERE, "libconp_oracle.so")
    src = os.path.join(HERE, "conp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return so


def load(fast: bool = False):
    build()
    lib = C.CDLL(os.path.join(HERE, "libconp_oracle_fast.so" if fast else "libconp_oracle.so"))
    vp = C.c_void_p
    lib.orc_kspace_create.restype = vp
    lib.orc_kspace_create.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double,
                                      C.c_double, C.c_longlong, C.c_double, C.c_double]
    lib.orc_kspace_destroy.argtypes = [vp]
    lib.orc_kspace_info.argtypes = [vp, _ip, _dp]
    lib.orc_kspace_tables.argtypes = [vp, _ip, _ip, _ip, _dp, _ip, _ip]
    lib.orc_sincos_b.restype = C.c_int
    lib.orc_sincos_b.argtypes = [vp, C.c_int, _dp, _dp, _ip, _dp, _dp]
    lib.orc_ele_trig.argtypes = [vp, C.c_int, _dp, _dp, _dp]
    lib.orc_bbb_from_sincos_b.argtypes = [vp, C.c_int, _dp, _dp, _dp, _dp, _dp]
    lib.orc_slabcorr.restype = C.c_double
    lib.orc_slabcorr.argtypes = [vp, C.c_int, _dp, _dp, _ip, C.c_int, _dp, _dp]
    lib.orc_aaa_from_sincos_a.argtypes = [vp, C.c_int, _dp, _dp, _dp, _dp]
    lib.orc_erfcr_sqrt.restype = C.c_double
    lib.orc_erfcr_sqrt.argtypes = [C.c_double]
    lib.orc_lu_inverse.restype = C.c_int
    lib.orc_lu_inverse.argtypes = [C.c_int, _dp]
    lib.orc_inv_project.restype = C.c_double
    lib.orc_inv_project.argtypes = [C.c_int, _dp, C.c_int, C.c_int, _dp, C.c_double]
    lib.orc_cg.restype = C.c_int
    lib.orc_cg.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int, C.c_double]
    lib.orc_fix_create.restype = vp
    lib.orc_fix_create.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                   C.c_int, C.c_double, C.c_int, _dp, C.c_double, C.c_double, C.c_double, vp]
    lib.orc_fix_destroy.argtypes = [vp]
    lib.orc_fix_set_atoms.argtypes = [vp, C.c_int, C.c_int, _dp, _dp, _ip, _ip, _ip]
    lib.orc_fix_set_lists.argtypes = [vp, C.c_int, _ip, _ip, _ip, _ip, C.c_int, _ip, _ip, _ip, _ip]
    for name in ("orc_fix_post_neighbor", "orc_fix_a_cal", "orc_fix_b_setq_cal", "orc_fix_get_setq"):
        getattr(lib, name).argtypes = [vp]
    for name in ("orc_fix_inv", "orc_fix_equation_solve", "orc_fix_linalg_setup"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = C.c_int
    lib.orc_fix_b_cal.argtypes = [vp, C.c_int]
    lib.orc_fix_update_charge.argtypes = [vp, C.c_double]
    lib.orc_fix_pre_force.argtypes = [vp, C.c_double]
    lib.orc_fix_update_charge_conq.argtypes = [vp, C.c_double]
    lib.orc_fix_update_charge_conq.restype = C.c_double
    lib.orc_fix_post_force.argtypes = [vp, C.c_double, _dp, _dp]
    lib.orc_fix_b_cal.argtypes = [vp, C.c_int]
    lib.orc_fix_update_charge_cond.argtypes = [vp, C.c_double, C.c_double, C.c_double, _dp]
    lib.orc_fix_update_charge_cond.restype = C.c_double
    lib.orc_fix_set_ehgo.argtypes = [vp, C.c_double, _dp, _dp]
    lib.orc_fix_set_ehgo.restype = C.c_int
    lib.orc_fix_sizes.argtypes = [vp, _ip]
    lib.orc_fix_scalars.argtypes = [vp, _dp]
    lib.orc_fix_get_maps.argtypes = [vp, _ip, _ip, _ip, _ip, _ip, _ip, _ip]
    lib.orc_fix_get_matrix.argtypes = [vp, _dp]
    lib.orc_fix_set_matrix.argtypes = [vp, _dp, C.c_int]
    lib.orc_fix_get_vectors.argtypes = [vp, _dp, _dp, _dp]
    lib.orc_fix_get_trig.argtypes = [vp, _dp, _dp]
    lib.orc_fix_blist_only.argtypes = [vp, _dp]
    lib.orc_fix_alist_only.argtypes = [vp, _dp]
    lib.orc_set_threads.argtypes = [C.c_int]
    lib.orc_set_threads.restype = C.c_int
    lib.orc_gemv_rows.argtypes = [C.c_int, _dp, _dp, _dp]
    lib.orc_multirank_maps.argtypes = [C.c_int, _ip, _ip, _ip, _ip, C.c_int, _ip, _ip, _ip, _ip]
    return lib


class KSpace:
    """km_ewald.cpp conp_setup tables"""

    def __init__(self, lib, g_ewald, accuracy, slab_volfactor, slabflag, prd, qsqsum, natoms, qqrd2e=332.06371,
                 dielectric=1.0):
        self.lib = lib
        self.h = lib.orc_kspace_create(g_ewald, accuracy, slab_volfactor, int(slabflag), float(prd[0]), float(prd[1]),
                                       float(prd[2]), float(qsqsum), int(natoms), qqrd2e, dielectric)
        io = np.zeros(16, np.int32); do = np.zeros(8, np.float64)
        lib.orc_kspace_info(self.h, io, do)
        (self.kcount, self.kcount_flat, self.kcount_expand, self.kxmax, self.kymax, self.kzmax, self.kmax,
         self.kmax3d) = [int(v) for v in io[:8]]
        self.kcount_dims = io[8:15].copy()
        self.unitk = do[:3].copy(); self.volume = do[3]; self.gsqmx = do[4]; self.ug_tot = do[5]
        self.g_ewald = do[6]
        K, KE = self.kcount, self.kcount_expand
        self.kxvecs = np.zeros(K, np.int32); self.kyvecs = np.zeros(K, np.int32); self.kzvecs = np.zeros(K, np.int32)
        self.ug = np.zeros(K); self.kxy_list = np.zeros(max(KE, 1), np.int32); self.kz_list = np.zeros(max(KE, 1), np.int32)
        lib.orc_kspace_tables(self.h, self.kxvecs, self.kyvecs, self.kzvecs, self.ug, self.kxy_list, self.kz_list)
        self.kxy_list = self.kxy_list[:KE]; self.kz_list = self.kz_list[:KE]

    @classmethod
    def from_system(cls, lib, s):
        return cls(lib, s.g_ewald, s.accuracy, s.slab_volfactor, s.slabflag, s.prd, s.qsqsum, s.natoms)

    def sincos_b(self, x, q, echeck, nlocal=None):
        nlocal = len(q) if nlocal is None else nlocal
        sr = np.zeros(self.kcount); si = np.zeros(self.kcount)
        self.lib.orc_sincos_b(self.h, nlocal, np.ascontiguousarray(x), np.ascontiguousarray(q),
                              np.ascontiguousarray(echeck, np.int32), sr, si)
        return sr, si

    def ele_trig(self, xele):
        ne = len(xele)
        csk = np.zeros((ne, self.kcount_flat)); snk = np.zeros((ne, self.kcount_flat))
        self.lib.orc_ele_trig(self.h, ne, np.ascontiguousarray(xele), csk, snk)
        return csk, snk

    def bbb(self, csk, snk, sr, si):
        b = np.zeros(len(csk))
        self.lib.orc_bbb_from_sincos_b(self.h, len(csk), csk, snk, sr, si, b)
        return b

    def aaa(self, csk, snk, xele):
        ne = len(csk)
        a = np.zeros((ne, ne))
        self.lib.orc_aaa_from_sincos_a(self.h, ne, csk, snk, np.ascontiguousarray(xele), a)
        return a

    def close(self):
        if self.h:
            self.lib.orc_kspace_destroy(self.h); self.h = None


class Fix:
    """single-rank restatement of FixConp driven like LAMMPS drives the fix"""

    def __init__(self, lib, s, minimizer=1, maxiter=100, tolerance=1e-6, nullneutral=True, qinit=False, one_electrode=False):
        from conp_amd.systems import EVSCALE
        self.lib, self.s = lib, s
        self.ks = KSpace.from_system(lib, s)
        one_electrode = int(one_electrode)
        self.h = lib.orc_fix_create(s.eta, s.ff_flag, int(s.zneutr), int(nullneutral), minimizer, maxiter, tolerance,
                                    int(s.newton), int(qinit), one_electrode, EVSCALE, s.ntypes,
                                    np.ascontiguousarray(s.cutsq_table()), s.cutoff, float(s.boxlo[2]), float(s.prd[2]),
                                    self.ks.h)
        self._keep = []

    def set_atoms(self, at):
        self.at = at
        self._keep = [np.ascontiguousarray(at.x), at.q, at.type, at.tag, at.echeck]
        self.lib.orc_fix_set_atoms(self.h, at.nlocal, at.nghost, self._keep[0], at.q, at.type, at.tag, at.echeck)

    def set_lists(self, alist, blist):
        self._lists = (alist, blist)
        self.lib.orc_fix_set_lists(self.h, alist.inum, alist.ilist, alist.numneigh, alist.first, _nz(alist.neigh),
                                   blist.inum, blist.ilist, blist.numneigh, blist.first, _nz(blist.neigh))

    def post_neighbor(self):
        self.lib.orc_fix_post_neighbor(self.h)

    def linalg_setup(self):
        return self.lib.orc_fix_linalg_setup(self.h)

    def pre_force(self, potdiff):
        self.lib.orc_fix_pre_force(self.h, potdiff)

    def set_ehgo(self, kappa, eta_i, u0_ev):
        return self.lib.orc_fix_set_ehgo(self.h, kappa, np.ascontiguousarray(eta_i, np.float64), np.ascontiguousarray(u0_ev, np.float64))

    def pre_force_conq(self, rightcharge):
        self.lib.orc_fix_b_cal(self.h, 1)
        self.lib.orc_fix_equation_solve(self.h)
        return self.lib.orc_fix_update_charge_conq(self.h, rightcharge)

    def pre_force_cond(self, rightcharge, setzvec):
        self.lib.orc_fix_b_cal(self.h, 1)
        self.lib.orc_fix_equation_solve(self.h)
        return self.lib.orc_fix_update_charge_cond(self.h, rightcharge, float(self.s.prd[0]), float(self.s.prd[1]),
                                                   np.ascontiguousarray(setzvec))

    def post_force(self, qqrd2e=332.06371):
        fadd = np.zeros((self.at.nlocal + self.at.nghost, 3)); out = np.zeros(8)
        self.lib.orc_fix_post_force(self.h, qqrd2e, fadd, out)
        return fadd, out

    def sizes(self):
        o = np.zeros(8, np.int32); self.lib.orc_fix_sizes(self.h, o)
        return dict(elenum=int(o[0]), elenum_all=int(o[1]), elytenum=int(o[2]), maxtag_all=int(o[3]), runstage=int(o[4]),
                    cg_iters=int(o[5]))

    def scalars(self):
        o = np.zeros(4); self.lib.orc_fix_scalars(self.h, o)
        return dict(totsetq=o[0], scalar_output=o[1], totinve=o[2], slabcorr=o[3])

    def maps(self):
        sz = self.sizes(); n, na, mt = sz["elenum"], sz["elenum_all"], sz["maxtag_all"]
        m = dict(ele2tag=np.zeros(n, np.int32), ele2eleall=np.zeros(n, np.int32), eleall2tag=np.zeros(na, np.int32),
                 eleall2ele=np.zeros(na + 1, np.int32), elecheck_eleall=np.zeros(na, np.int32),
                 elebuf2eleall=np.zeros(na, np.int32), tag2eleall=np.zeros(mt + 1, np.int32))
        self.lib.orc_fix_get_maps(self.h, m["ele2tag"], m["ele2eleall"], m["eleall2tag"], m["eleall2ele"],
                                  m["elecheck_eleall"], m["elebuf2eleall"], m["tag2eleall"])
        return m

    def matrix(self):
        na = self.sizes()["elenum_all"]
        a = np.zeros((na, na)); self.lib.orc_fix_get_matrix(self.h, a)
        return a

    def vectors(self):
        na = self.sizes()["elenum_all"]
        b, q, sq = np.zeros(na), np.zeros(na), np.zeros(na)
        self.lib.orc_fix_get_vectors(self.h, b, q, sq)
        return b, q, sq

    def trig(self):
        na = self.sizes()["elenum_all"]
        c = np.zeros((na, self.ks.kcount_flat)); s = np.zeros((na, self.ks.kcount_flat))
        self.lib.orc_fix_get_trig(self.h, c, s)
        return c, s

    def blist_only(self):
        o = np.zeros(self.sizes()["elenum"]); self.lib.orc_fix_blist_only(self.h, o)
        return o

    def alist_only(self):
        sz = self.sizes()
        o = np.zeros((sz["elenum"], sz["elenum_all"])); self.lib.orc_fix_alist_only(self.h, o)
        return o

    def close(self):
        if self.h:
            self.lib.orc_fix_destroy(self.h); self.h = None
        self.ks.close()


def _nz(a):
    """ctypes ndpointer rejects zero-length arrays on some numpy versions; pad"""
    return a if a.size else np.zeros(1, np.int32)


class Pppm:
    """oracle restatement of the PPPM b vector (pppm_conp.cpp b_cal chain); parity unpinned at the LAMMPS boundary"""

    def __init__(self, lib, s, mesh, order=5):
        lib.orc_pppm_create.restype = C.c_void_p
        lib.orc_pppm_create.argtypes = [C.c_int] * 4 + [C.c_double, C.c_double, C.c_int, _dp, _dp]
        lib.orc_pppm_b_cal.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip, C.c_int, _dp, _dp, C.c_void_p]
        lib.orc_pppm_destroy.argtypes = [C.c_void_p]
        self.lib = lib
        self.h = lib.orc_pppm_create(mesh[0], mesh[1], mesh[2], order, s.g_ewald, s.slab_volfactor, int(s.slabflag),
                                     np.ascontiguousarray(s.boxlo), np.ascontiguousarray(s.prd))

    def b_cal(self, x, q, echeck, nlocal, xele):
        b = np.zeros(len(xele))
        self.lib.orc_pppm_b_cal(self.h, nlocal, np.ascontiguousarray(x), np.ascontiguousarray(q),
                                np.ascontiguousarray(echeck, np.int32), len(xele), np.ascontiguousarray(xele), b, None)
        return b

    # ---- PPPM coupling beyond b (pppm_conp.cpp:385-534) and compute potential/atom (compute_potential_atom.cpp:120-345)
    def _nfft(self, mesh):
        return int(mesh[0]) * int(mesh[1]) * int(mesh[2])

    def make_rho(self, mesh, x, q, echeck, nlocal):
        n = self._nfft(mesh)
        d, e, l = np.zeros(n), np.zeros(n), np.zeros(n)
        self.lib.orc_pppm_make_rho.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip, _dp, _dp, _dp]
        self.lib.orc_pppm_make_rho(self.h, nlocal, np.ascontiguousarray(x), np.ascontiguousarray(q),
                                   np.ascontiguousarray(echeck, np.int32), d, e, l)
        return d, e, l

    def group_potential(self, x, q, echeck, nlocal, sel, particle=False):
        out = np.zeros(nlocal)
        self.lib.orc_pppm_group_potential.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip, _ip, C.c_int, _dp]
        self.lib.orc_pppm_group_potential(self.h, nlocal, np.ascontiguousarray(x), np.ascontiguousarray(q),
                                          np.ascontiguousarray(echeck, np.int32), np.ascontiguousarray(sel, np.int32), int(particle), out)
        return out

    def compute_potential_atom(self, s, at, lst, sel, etasel, eta=0.0, pair=True, kspace=True, qsum=True):
        nall = at.nlocal + at.nghost
        pot = np.zeros(nall)
        neigh = lst.neigh if lst.neigh.size else np.zeros(1, np.int32)
        self.lib.orc_compute_potential_atom.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _ip, _ip, _ip, _ip, C.c_int, _ip, _ip, _ip,
                                                        _ip, C.c_int, C.c_int, _dp, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                                        C.c_int, C.c_int, C.c_double, C.c_double, _dp]
        vol = float(s.prd[0] * s.prd[1] * s.prd[2] * s.slab_volfactor)
        from conp_amd import systems as _sy
        self.lib.orc_compute_potential_atom(self.h, at.nlocal, at.nghost, np.ascontiguousarray(at.x), np.ascontiguousarray(at.q),
                                            np.ascontiguousarray(at.type, np.int32), np.ascontiguousarray(at.echeck, np.int32),
                                            np.ascontiguousarray(sel, np.int32), np.ascontiguousarray(etasel, np.int32), lst.inum,
                                            lst.ilist, lst.numneigh, lst.first, neigh, int(s.newton), s.ntypes,
                                            np.ascontiguousarray(s.cutsq_table().ravel()), float(s.cutoff), float(s.g_ewald), float(eta),
                                            int(pair), int(kspace), int(s.slabflag), int(qsum), vol, _sy.QQR2E / _sy.QE2F, pot)
        return pot

    def close(self):
        if self.h:
            self.lib.orc_pppm_destroy(self.h); self.h = None
