"""ctypes binding of include/conp_hip.h and a thin driver that calls the hooks in the order LAMMPS does."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import systems as _systems

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class ConpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[conp status {code}] {msg}")
        self.code = code
        self.msg = msg


def library_path():
    # CONP_LIB: a diagnostic build of the same ABI (tools/sk_stamp.py loads the s_memtime-stamped one); never set in tests / bench
    return os.environ.get("CONP_LIB") or os.path.join(HERE, "libconp_hip.so")


class conp_fix_args(C.Structure):
    _fields_ = [("everynum", C.c_int), ("eta", C.c_double), ("potdiff", C.c_double), ("potdiff_is_variable", C.c_int),
                ("ff_flag", C.c_int), ("zneutr", C.c_int), ("matout", C.c_int), ("pppm", C.c_int), ("split", C.c_int),
                ("qinit", C.c_int), ("lowmem", C.c_int), ("nullneutral", C.c_int), ("ehgo", C.c_int),
                ("a_matrix_f", C.c_int), ("a_matrix_file", C.c_char * 512), ("smartlist", C.c_int),
                ("eletypenum", C.c_int), ("eletypes", C.c_int * 32), ("minimizer", C.c_int), ("maxiter", C.c_int),
                ("tolerance", C.c_double), ("logfile", C.c_char * 512), ("group2", C.c_char * 128), ("potdiff_var", C.c_char * 128), ("conq", C.c_int), ("cond", C.c_int)]


class conp_env(C.Structure):
    _fields_ = [("qqrd2e", C.c_double), ("qqr2e", C.c_double), ("qe2f", C.c_double), ("dielectric", C.c_double),
                ("newton_pair", C.c_int), ("g_ewald", C.c_double), ("accuracy", C.c_double),
                ("slab_volfactor", C.c_double), ("slabflag", C.c_int), ("xprd", C.c_double), ("yprd", C.c_double),
                ("zprd", C.c_double), ("boxlo_z", C.c_double), ("boxlo_x", C.c_double), ("boxlo_y", C.c_double), ("ntypes", C.c_int), ("cutsq", C.POINTER(C.c_double)),
                ("cut_coul", C.c_double), ("one_electrode", C.c_int), ("device", C.c_int), ("rank", C.c_int),
                ("nranks", C.c_int), ("pppm_nx", C.c_int), ("pppm_ny", C.c_int), ("pppm_nz", C.c_int), ("pppm_order", C.c_int),
                ("ghost_images", C.c_int)]


class conp_atoms(C.Structure):
    _fields_ = [("nlocal", C.c_int), ("nghost", C.c_int), ("x", C.POINTER(C.c_double)), ("q", C.POINTER(C.c_double)),
                ("type", C.POINTER(C.c_int)), ("tag", C.POINTER(C.c_int)), ("echeck", C.POINTER(C.c_int))]


class conp_neighlist(C.Structure):
    _fields_ = [("inum", C.c_int), ("ilist", C.POINTER(C.c_int)), ("numneigh", C.POINTER(C.c_int)),
                ("first", C.POINTER(C.c_int)), ("neigh", C.POINTER(C.c_int)), ("nneigh", C.c_int64)]


_CB_SUM = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)
_CB_MAXI = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int)
_CB_GATHER_INT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int))
_CB_GATHERV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64))


class conp_comm(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int), ("nranks", C.c_int), ("allreduce_sum", _CB_SUM),
                ("allreduce_max_int", _CB_MAXI), ("allgather_int", _CB_GATHER_INT), ("allgatherv", _CB_GATHERV)]


class conp_potential_args(C.Structure):
    _fields_ = [("pairflag", C.c_int), ("kspaceflag", C.c_int), ("qsumflag", C.c_int), ("eta", C.c_double)]


class conp_info(C.Structure):
    _fields_ = [("elenum", C.c_int), ("elenum_all", C.c_int), ("elytenum", C.c_int), ("maxtag_all", C.c_int),
                ("runstage", C.c_int), ("kcount", C.c_int), ("kcount_flat", C.c_int), ("kcount_expand", C.c_int),
                ("kxmax", C.c_int), ("kymax", C.c_int), ("kzmax", C.c_int), ("kmax", C.c_int), ("kmax3d", C.c_int),
                ("kcount_dims", C.c_int * 7), ("cg_iterations", C.c_int), ("n_zclasses", C.c_int), ("unitk", C.c_double * 3),
                ("volume", C.c_double), ("gsqmx", C.c_double), ("ug_tot", C.c_double), ("totsetq", C.c_double),
                ("scalar_output", C.c_double), ("totinve", C.c_double), ("slabcorr", C.c_double),
                ("n_blist_pairs", C.c_int64), ("n_alist_pairs", C.c_int64), ("n_elyte_charged", C.c_int64),
                ("inverse_path", C.c_int), ("inverse_retries", C.c_int), ("pppm_elyte_spreads", C.c_int), ("zn_cols", C.c_int), ("zn_grid", C.c_int), ("zn_rows", C.c_int)]


# every symbol include/conp_hip.h declares (checked by tests/test_capi_symbols.py without a GPU)
SYMBOLS = [
    "conp_parse_fix_args", "conp_fix_create", "conp_fix_destroy", "conp_last_error", "conp_abi_version",
    "conp_fix_init_list", "conp_fix_setup_post_neighbor", "conp_fix_setup_pre_force", "conp_fix_post_neighbor",
    "conp_fix_pre_force", "conp_fix_compute_scalar", "conp_fix_post_force", "conp_fix_post_force_step", "conp_fix_modify_param", "conp_fix_linalg_setup", "conp_fix_a_cal", "conp_fix_b_cal",
    "conp_fix_equation_solve", "conp_fix_update_charge", "conp_km_conp_setup", "conp_km_a_cal", "conp_km_b_cal",
    "conp_fix_info", "conp_fix_get_ktables", "conp_fix_get_maps", "conp_fix_get_matrix", "conp_fix_set_matrix",
    "conp_fix_get_vectors", "conp_fix_get_sfac", "conp_fix_get_ele_trig", "conp_inv_project", "conp_invert",
    "conp_host_ktables", "conp_host_index", "conp_host_pair_rows", "conp_fix_write_matrix_file", "conp_fix_read_matrix_file",
    "conp_fix_set_stream",
    "conp_fix_bind_device_buffers", "conp_fix_row_range", "conp_fix_b_cal_device", "conp_fix_solve_device",
    "conp_fix_scatter_device", "conp_fix_pre_force_device", "conp_fix_profile", "conp_fix_profile_read", "conp_debug_check_guards",
    "conp_debug_set_paths", "conp_debug_set_sk_workgroups",
    "conp_fix_pin_host_arrays", "conp_fix_unpin_host_arrays", "conp_host_alloc", "conp_host_free",
    "conp_fix_write_timing", "conp_fix_log_drain", "conp_fix_mesg_drain",
    "conp_fix_set_comm", "conp_rccl_unique_id", "conp_fix_comm_init_rccl", "conp_rccl_available", "conp_fix_comm_destroy_rccl",
    "conp_pppm_make_rho", "conp_pppm_compute_group_potential", "conp_pppm_compute_particle_potential",
    "conp_pppm_keep_density", "conp_pppm_compute",
    "conp_compute_potential_atom",
]


# test hooks of the ABI (include/conp_hip.h, CONP_PATH_*): alternative code paths the parity tests compare the default ones with
PATH_PARTIAL_TILES, PATH_A_GENERAL, PATH_INV_PIVOTED, PATH_CG_TWO_LAUNCH, PATH_GEMV_ROWS = 1, 2, 4, 8, 16
PATH_PHASE_LAUNCH, PATH_PPPM_SPREAD_LAUNCH, PATH_ROWS_HOST, PATH_TIME_SPLIT, PATH_HC_NO_WAIT = 32, 64, 128, 256, 512
PATH_HC_FUSED = 1024
PATH_CG_PERSIST = 2048
PATH_SK_CLASSIC = 4096


class test_paths:
    """with capi.test_paths(capi.PATH_PARTIAL_TILES): ...   -- process-wide; handles created inside take the alternative path"""

    def __init__(self, mask=0, sk_workgroups=0):
        self.mask, self.nwg = mask, sk_workgroups

    def __enter__(self):
        lib = load_library()
        if not hasattr(lib, "conp_debug_set_paths"):
            raise RuntimeError("this library build has no test hooks (a comparison build of an earlier round?)")
        lib.conp_debug_set_paths(self.mask)
        lib.conp_debug_set_sk_workgroups(self.nwg)
        return self

    def __exit__(self, *a):
        lib = load_library()
        lib.conp_debug_set_paths(0)
        lib.conp_debug_set_sk_workgroups(0)
        return False


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(path)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
    lib.conp_last_error.restype = C.c_char_p
    lib.conp_parse_fix_args.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.POINTER(conp_fix_args)]
    lib.conp_fix_create.argtypes = [C.POINTER(conp_fix_args), C.POINTER(conp_env), C.POINTER(vp)]
    lib.conp_fix_destroy.argtypes = [vp]
    lib.conp_fix_destroy.restype = None
    if hasattr(lib, "conp_fix_pin_host_arrays"):      # (comparison builds of earlier rounds, loaded through CONP_LIB, lack these)
        lib.conp_fix_pin_host_arrays.argtypes = [vp, dp, dp, C.c_int]
        lib.conp_fix_unpin_host_arrays.argtypes = [vp]
        lib.conp_host_alloc.argtypes = [C.c_size_t]
        lib.conp_host_alloc.restype = C.c_void_p
        lib.conp_host_free.argtypes = [C.c_void_p]
        lib.conp_host_free.restype = None
    if hasattr(lib, "conp_debug_set_paths"):
        lib.conp_debug_set_paths.argtypes = [C.c_uint]
        lib.conp_debug_set_paths.restype = None
        lib.conp_debug_set_sk_workgroups.argtypes = [C.c_int]
        lib.conp_debug_set_sk_workgroups.restype = None
    lib.conp_fix_init_list.argtypes = [vp, C.c_int, C.POINTER(conp_neighlist)]
    for n in ("conp_fix_setup_post_neighbor", "conp_fix_post_neighbor", "conp_fix_linalg_setup", "conp_fix_a_cal",
              "conp_fix_b_cal"):
        getattr(lib, n).argtypes = [vp, C.POINTER(conp_atoms)]
    for n in ("conp_fix_setup_pre_force", "conp_fix_pre_force"):
        getattr(lib, n).argtypes = [vp, C.POINTER(conp_atoms), C.c_int64, C.c_double]
    lib.conp_fix_compute_scalar.argtypes = [vp]
    lib.conp_fix_compute_scalar.restype = C.c_double
    lib.conp_fix_post_force.argtypes = [vp, C.POINTER(conp_atoms), dp, dp, dp, dp]
    lib.conp_fix_post_force_step.argtypes = [vp, C.POINTER(conp_atoms), C.c_int64, dp, dp, dp, dp]
    lib.conp_fix_modify_param.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), ip]
    lib.conp_fix_equation_solve.argtypes = [vp]
    lib.conp_fix_update_charge.argtypes = [vp, C.POINTER(conp_atoms), C.c_double]
    lib.conp_km_conp_setup.argtypes = [vp, C.c_double, C.c_int64]
    lib.conp_km_a_cal.argtypes = [vp, C.POINTER(conp_atoms), dp]
    lib.conp_km_b_cal.argtypes = [vp, C.POINTER(conp_atoms), dp]
    lib.conp_fix_info.argtypes = [vp, C.POINTER(conp_info)]
    lib.conp_fix_get_ktables.argtypes = [vp, ip, ip, ip, dp, ip, ip]
    lib.conp_fix_get_maps.argtypes = [vp, ip, ip, ip, ip, ip, ip, ip]
    lib.conp_fix_get_matrix.argtypes = [vp, dp]
    lib.conp_fix_set_matrix.argtypes = [vp, dp, C.c_int]
    lib.conp_fix_get_vectors.argtypes = [vp, dp, dp, dp]
    lib.conp_fix_get_sfac.argtypes = [vp, dp, dp]
    lib.conp_fix_get_ele_trig.argtypes = [vp, dp, dp]
    lib.conp_inv_project.argtypes = [vp, C.c_int, dp, C.c_int, C.c_int, dp, C.c_double, dp]
    lib.conp_invert.argtypes = [vp, C.c_int, dp]
    lib.conp_host_ktables.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_int64, C.c_double, C.c_double, ip, ip, ip, ip, dp, ip, ip, ip, ip, ip]
    lib.conp_host_index.argtypes = [C.c_int, ip, ip, C.c_int, ip, ip, ip, ip, ip, ip, ip, ip, ip]
    lib.conp_host_pair_rows.argtypes = [C.c_int, C.POINTER(conp_neighlist), C.POINTER(conp_atoms), C.c_int, ip, ip, ip, ip]
    lib.conp_host_pair_rows.restype = C.c_int64
    lib.conp_fix_write_matrix_file.argtypes = [vp, C.c_char_p, C.c_int]
    lib.conp_fix_read_matrix_file.argtypes = [vp, C.POINTER(conp_atoms), C.c_char_p]
    lib.conp_fix_set_stream.argtypes = [vp, vp]
    lib.conp_fix_bind_device_buffers.argtypes = [vp, vp, vp]
    lib.conp_fix_row_range.argtypes = [vp, ip, ip]
    lib.conp_fix_b_cal_device.argtypes = [vp, vp, vp]
    lib.conp_fix_solve_device.argtypes = [vp, C.c_double]
    lib.conp_fix_scatter_device.argtypes = [vp, vp, C.c_double]
    lib.conp_fix_pre_force_device.argtypes = [vp, vp, vp, C.c_double]
    lib.conp_fix_profile.argtypes = [vp, C.c_int]
    lib.conp_fix_profile_read.argtypes = [vp, ip, C.POINTER(C.c_char_p), dp, ip]
    lib.conp_fix_write_timing.argtypes = [vp]
    lib.conp_fix_log_drain.argtypes = [vp]
    lib.conp_fix_log_drain.restype = C.c_char_p
    lib.conp_fix_mesg_drain.argtypes = [vp]
    lib.conp_fix_mesg_drain.restype = C.c_char_p
    lib.conp_pppm_make_rho.argtypes = [vp, C.POINTER(conp_atoms), dp, dp, dp]
    lib.conp_pppm_compute_group_potential.argtypes = [vp, C.POINTER(conp_atoms), ip, dp]
    lib.conp_pppm_compute_particle_potential.argtypes = [vp, C.POINTER(conp_atoms), C.c_int, dp]
    if hasattr(lib, "conp_pppm_compute"):
        lib.conp_pppm_compute.argtypes = [vp, C.POINTER(conp_atoms)]
        lib.conp_pppm_keep_density.argtypes = [vp, C.c_int]
    lib.conp_compute_potential_atom.argtypes = [vp, C.POINTER(conp_atoms), C.POINTER(conp_neighlist), ip, ip,
                                                C.POINTER(conp_potential_args), dp]
    lib.conp_fix_set_comm.argtypes = [vp, C.POINTER(conp_comm)]
    lib.conp_rccl_unique_id.argtypes = [C.c_void_p]
    lib.conp_fix_comm_init_rccl.argtypes = [vp, C.c_void_p]
    _LIB = lib
    return lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def parse_fix_command(tokens: Sequence[str], ntypes: int) -> conp_fix_args:
    """tokens = the whole `fix` command split on whitespace: ID group1 conp Nevery group2 eta DV logfile [keywords]"""
    lib = load_library()
    arr = (C.c_char_p * len(tokens))(*[t.encode() for t in tokens])
    out = conp_fix_args()
    rc = lib.conp_parse_fix_args(len(tokens), arr, ntypes, C.byref(out))
    if rc != 0:
        raise ConpError(rc, lib.conp_last_error().decode())
    return out


def fix_command_for(s: "_systems.System", extra: Sequence[str] = (), style: str = "conp") -> list:
    """the fix command the reference's decks would use for this system"""
    toks = ["e", "eleleft", style, "1", "eleright", repr(float(s.eta)), repr(float(s.potdiff)), "log_conp"]
    if s.eletypes is not None:
        toks += ["etypes", str(len(s.eletypes))] + [str(t) for t in s.eletypes]
    if s.ff_flag == 1:
        toks.append("ffield")
    elif s.ff_flag == 2:
        toks.append("noslab")
    if s.zneutr:
        toks.append("zneutr")
    return toks + list(extra)


class FixConp:
    """Drives one `fix conp` instance through the C ABI the way LAMMPS' Modify drives the reference fix:
    init_list -> setup_post_neighbor -> setup_pre_force -> [post_neighbor] -> pre_force ...."""

    def __init__(self, s: "_systems.System", extra_args: Sequence[str] = (), device: int = 0, rank: int = 0,
                 nranks: int = 1, one_electrode: bool = False, style: str = "conp", pppm_mesh=None, pppm_order: int = 5,
                 ghost_images: bool = False):
        self.lib = load_library()
        self.s = s
        self.args = parse_fix_command(fix_command_for(s, extra_args, style), s.ntypes)
        self._cutsq = np.ascontiguousarray(s.cutsq_table())
        env = conp_env(qqrd2e=_systems.QQRD2E, qqr2e=_systems.QQR2E, qe2f=_systems.QE2F, dielectric=1.0,
                       newton_pair=int(s.newton), g_ewald=s.g_ewald, accuracy=s.accuracy,
                       slab_volfactor=s.slab_volfactor, slabflag=s.slabflag, xprd=float(s.prd[0]), yprd=float(s.prd[1]),
                       zprd=float(s.prd[2]), boxlo_z=float(s.boxlo[2]), ntypes=s.ntypes, cutsq=_dptr(self._cutsq),
                       cut_coul=s.cutoff, one_electrode=int(one_electrode), device=device, rank=rank, nranks=nranks,
                       boxlo_x=float(s.boxlo[0]), boxlo_y=float(s.boxlo[1]),
                       pppm_nx=pppm_mesh[0] if pppm_mesh else 0, pppm_ny=pppm_mesh[1] if pppm_mesh else 0,
                       pppm_nz=pppm_mesh[2] if pppm_mesh else 0, pppm_order=pppm_order if pppm_mesh else 0,
                       ghost_images=int(ghost_images))
        self.h = C.c_void_p()
        self._check(self.lib.conp_fix_create(C.byref(self.args), C.byref(env), C.byref(self.h)))
        self._keep = {}

    def _check(self, rc):
        if rc != 0:
            raise ConpError(rc, self.lib.conp_last_error().decode())

    # -- views -----------------------------------------------------------------------------------
    def atoms_view(self, at) -> conp_atoms:
        x = np.ascontiguousarray(at.x, dtype=np.float64)
        self._keep["atoms"] = (x, at.q, at.type, at.tag, at.echeck)
        return conp_atoms(nlocal=at.nlocal, nghost=at.nghost, x=_dptr(x), q=_dptr(at.q), type=_iptr(at.type),
                          tag=_iptr(at.tag), echeck=_iptr(at.echeck))

    def init_list(self, which: int, lst):
        neigh = lst.neigh if lst.neigh.size else np.zeros(1, np.int32)
        self._keep[f"list{which}"] = (lst.ilist, lst.numneigh, lst.first, neigh)
        v = conp_neighlist(inum=lst.inum, ilist=_iptr(lst.ilist), numneigh=_iptr(lst.numneigh), first=_iptr(lst.first),
                           neigh=_iptr(neigh), nneigh=int(lst.neigh.size))
        self._check(self.lib.conp_fix_init_list(self.h, which, C.byref(v)))

    def init_lists(self, alist, blist):
        if alist is blist:
            self.init_list(2, alist)
        else:
            self.init_list(0, alist)
            self.init_list(1, blist)

    # -- hooks (same names as the reference's Fix methods) ---------------------------------------------
    def setup_post_neighbor(self, at):
        self._check(self.lib.conp_fix_setup_post_neighbor(self.h, C.byref(self.atoms_view(at))))

    def setup_pre_force(self, at, ntimestep=0, potdiff=None):
        pd = self.s.potdiff if potdiff is None else potdiff
        self._check(self.lib.conp_fix_setup_pre_force(self.h, C.byref(self.atoms_view(at)), ntimestep, pd))

    def linalg_setup(self, at):
        self._check(self.lib.conp_fix_linalg_setup(self.h, C.byref(self.atoms_view(at))))

    def post_neighbor(self, at):
        self._check(self.lib.conp_fix_post_neighbor(self.h, C.byref(self.atoms_view(at))))

    def pre_force(self, at, ntimestep=0, potdiff=None):
        pd = self.s.potdiff if potdiff is None else potdiff
        self._check(self.lib.conp_fix_pre_force(self.h, C.byref(self.atoms_view(at)), ntimestep, pd))

    def a_cal(self, at):
        self._check(self.lib.conp_fix_a_cal(self.h, C.byref(self.atoms_view(at))))

    def b_cal(self, at):
        self._check(self.lib.conp_fix_b_cal(self.h, C.byref(self.atoms_view(at))))

    def equation_solve(self):
        self._check(self.lib.conp_fix_equation_solve(self.h))

    def update_charge(self, at, potdiff=None):
        pd = self.s.potdiff if potdiff is None else potdiff
        self._check(self.lib.conp_fix_update_charge(self.h, C.byref(self.atoms_view(at)), pd))

    def modify_param(self, *tokens):
        """fix_modify ID <tokens>, e.g. modify_param("ehgo", "coeff", "5", "1.979", "auto")"""
        arr = (C.c_char_p * len(tokens))(*[str(t).encode() for t in tokens])
        n = C.c_int()
        self._check(self.lib.conp_fix_modify_param(self.h, len(tokens), arr, C.byref(n)))
        return n.value

    def post_force(self, at):
        f = np.zeros((at.nlocal + at.nghost, 3)); ek = C.c_double(); ec = C.c_double(); vir = np.zeros(6)
        self._check(self.lib.conp_fix_post_force(self.h, C.byref(self.atoms_view(at)), _dptr(f), C.byref(ek), C.byref(ec), _dptr(vir)))
        return f, ek.value, ec.value, vir

    def post_force_step(self, at, ntimestep):
        f = np.zeros((at.nlocal + at.nghost, 3)); ek = C.c_double(); ec = C.c_double(); vir = np.zeros(6)
        self._check(self.lib.conp_fix_post_force_step(self.h, C.byref(self.atoms_view(at)), C.c_int64(ntimestep), _dptr(f),
                                                      C.byref(ek), C.byref(ec), _dptr(vir)))
        return f, ek.value, ec.value, vir

    def compute_scalar(self):
        return float(self.lib.conp_fix_compute_scalar(self.h))

    # -- provider surface ------------------------------------------------------------------------
    def km_conp_setup(self, qsqsum, natoms):
        self._check(self.lib.conp_km_conp_setup(self.h, qsqsum, natoms))

    def km_b_cal(self, at):
        b = np.zeros(self.info().elenum_all)
        self._check(self.lib.conp_km_b_cal(self.h, C.byref(self.atoms_view(at)), _dptr(b)))
        return b

    def km_a_cal(self, at):
        ne = self.info().elenum_all
        a = np.zeros((ne, ne))
        self._check(self.lib.conp_km_a_cal(self.h, C.byref(self.atoms_view(at)), _dptr(a)))
        return a

    # -- read-back -------------------------------------------------------------------------------
    def info(self) -> conp_info:
        o = conp_info()
        self._check(self.lib.conp_fix_info(self.h, C.byref(o)))
        return o

    def ktables(self):
        i = self.info()
        K, E = i.kcount, max(i.kcount_expand, 1)
        kx, ky, kz = (np.zeros(K, np.int32) for _ in range(3))
        ug = np.zeros(K); kxy = np.zeros(E, np.int32); kzl = np.zeros(E, np.int32)
        self._check(self.lib.conp_fix_get_ktables(self.h, _iptr(kx), _iptr(ky), _iptr(kz), _dptr(ug), _iptr(kxy), _iptr(kzl)))
        return dict(kxvecs=kx, kyvecs=ky, kzvecs=kz, ug=ug, kxy_list=kxy[:i.kcount_expand], kz_list=kzl[:i.kcount_expand])

    def maps(self):
        i = self.info()
        n, na, mt = i.elenum, i.elenum_all, i.maxtag_all
        m = dict(ele2tag=np.zeros(n, np.int32), ele2eleall=np.zeros(n, np.int32), eleall2tag=np.zeros(na, np.int32),
                 eleall2ele=np.zeros(na + 1, np.int32), elecheck_eleall=np.zeros(na, np.int32),
                 elebuf2eleall=np.zeros(na, np.int32), tag2eleall=np.zeros(mt + 1, np.int32))
        self._check(self.lib.conp_fix_get_maps(self.h, *[_iptr(m[k]) for k in (
            "ele2tag", "ele2eleall", "eleall2tag", "eleall2ele", "elecheck_eleall", "elebuf2eleall", "tag2eleall")]))
        return m

    def matrix(self):
        ne = self.info().elenum_all
        a = np.zeros((ne, ne))
        self._check(self.lib.conp_fix_get_matrix(self.h, _dptr(a)))
        return a

    def set_matrix(self, a, runstage):
        a = np.ascontiguousarray(a, dtype=np.float64)
        self._check(self.lib.conp_fix_set_matrix(self.h, _dptr(a), runstage))

    def vectors(self):
        ne = self.info().elenum_all
        b, q, sq = np.zeros(ne), np.zeros(ne), np.zeros(ne)
        self._check(self.lib.conp_fix_get_vectors(self.h, _dptr(b), _dptr(q), _dptr(sq)))
        return b, q, sq

    def sfac(self):
        K = self.info().kcount
        sr, si = np.zeros(K), np.zeros(K)
        self._check(self.lib.conp_fix_get_sfac(self.h, _dptr(sr), _dptr(si)))
        return sr, si

    def ele_trig(self):
        i = self.info()
        c = np.zeros((i.elenum_all, i.kcount_flat)); s = np.zeros((i.elenum_all, i.kcount_flat))
        self._check(self.lib.conp_fix_get_ele_trig(self.h, _dptr(c), _dptr(s)))
        return c, s

    def inv_project(self, a, nullneutral=True, zneutr=False, eleallz=None, zhalf=0.0):
        a = np.ascontiguousarray(a, dtype=np.float64).copy()
        n = a.shape[0]
        z = np.zeros(n) if eleallz is None else np.ascontiguousarray(eleallz, dtype=np.float64)
        tot = C.c_double()
        self._check(self.lib.conp_inv_project(self.h, n, _dptr(a), int(nullneutral), int(zneutr), _dptr(z), zhalf, C.byref(tot)))
        return a, tot.value

    def write_matrix_file(self, path, which):
        self._check(self.lib.conp_fix_write_matrix_file(self.h, str(path).encode(), which))

    def read_matrix_file(self, at, path):
        self._check(self.lib.conp_fix_read_matrix_file(self.h, C.byref(self.atoms_view(at)), str(path).encode()))

    def invert(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64).copy()
        self._check(self.lib.conp_invert(self.h, a.shape[0], _dptr(a)))
        return a

    # -- PPPM coupling beyond b, compute potential/atom ---------------------------------------------
    def pppm_keep_density(self, on=True):
        self._check(self.lib.conp_pppm_keep_density(self.h, 1 if on else 0))

    def pppm_compute(self, at):
        """collective: the mesh potential of the total density (u_brick) formed and cached"""
        self._check(self.lib.conp_pppm_compute(self.h, C.byref(self.atoms_view(at))))

    def pppm_make_rho(self, at, nfft):
        d, e, l = np.zeros(nfft), np.zeros(nfft), np.zeros(nfft)
        self._check(self.lib.conp_pppm_make_rho(self.h, C.byref(self.atoms_view(at)), _dptr(d), _dptr(e), _dptr(l)))
        return d, e, l

    def pppm_group_potential(self, at, sel):
        sel = np.ascontiguousarray(sel, np.int32)
        out = np.zeros(at.nlocal)
        self._check(self.lib.conp_pppm_compute_group_potential(self.h, C.byref(self.atoms_view(at)), _iptr(sel), _dptr(out)))
        return out

    def pppm_particle_potential(self, at, i):
        u = C.c_double()
        self._check(self.lib.conp_pppm_compute_particle_potential(self.h, C.byref(self.atoms_view(at)), int(i), C.byref(u)))
        return u.value

    def compute_potential_atom(self, at, pairlist, sel, etasel=None, eta=0.0, pair=True, kspace=True, qsum=True):
        sel = np.ascontiguousarray(sel, np.int32)
        es = None if etasel is None else np.ascontiguousarray(etasel, np.int32)
        pot = np.zeros(at.nlocal + at.nghost)
        pa = conp_potential_args(pairflag=int(pair), kspaceflag=int(kspace), qsumflag=int(qsum), eta=float(eta))
        neigh = pairlist.neigh if pairlist.neigh.size else np.zeros(1, np.int32)
        lv = conp_neighlist(inum=pairlist.inum, ilist=_iptr(pairlist.ilist), numneigh=_iptr(pairlist.numneigh),
                            first=_iptr(pairlist.first), neigh=_iptr(neigh), nneigh=int(pairlist.neigh.size))
        self._check(self.lib.conp_compute_potential_atom(self.h, C.byref(self.atoms_view(at)), C.byref(lv), _iptr(sel),
                                                         _iptr(es) if es is not None else C.POINTER(C.c_int)(), C.byref(pa), _dptr(pot)))
        return pot

    # -- several ranks ---------------------------------------------------------------------------
    def set_comm_torch(self, group=None):
        """conp_fix_set_comm with callbacks on torch.distributed (any backend that moves CPU tensors, e.g. gloo): the atoms and
        lists handed to the hooks from now on are THIS RANK's sub-domain -- the role MPI plays for the LAMMPS glue"""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def cb_sum(_ctx, buf, n):
            a = np.ctypeslib.as_array(buf, shape=(n,))
            t = torch.from_numpy(a)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return 0

        def cb_maxi(_ctx, buf, n):
            a = np.ctypeslib.as_array(buf, shape=(n,))
            t = torch.from_numpy(a)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            return 0

        def cb_gather_int(_ctx, v, out):
            t = torch.tensor([v], dtype=torch.int32)
            outs = [torch.zeros(1, dtype=torch.int32) for _ in range(world)]
            dist.all_gather(outs, t, group=group)
            for r in range(world):
                out[r] = int(outs[r][0])
            return 0

        def cb_gatherv(_ctx, send, nbytes, recv, counts, displs):
            mine = torch.from_numpy(np.frombuffer((C.c_char * nbytes).from_address(send), dtype=np.uint8).copy()) if nbytes else torch.zeros(0, dtype=torch.uint8)
            outs = [torch.zeros(int(counts[r]), dtype=torch.uint8) for r in range(world)]
            dist.all_gather(outs, mine, group=group) if len({int(counts[r]) for r in range(world)}) == 1 else _uneven_all_gather(outs, mine, group, world)
            for r in range(world):
                n = int(counts[r])
                if n:
                    C.memmove(recv + int(displs[r]), outs[r].numpy().ctypes.data, n)
            return 0

        self._comm_cbs = (_CB_SUM(cb_sum), _CB_MAXI(cb_maxi), _CB_GATHER_INT(cb_gather_int), _CB_GATHERV(cb_gatherv))
        self._comm = conp_comm(ctx=None, rank=rank, nranks=world, allreduce_sum=self._comm_cbs[0], allreduce_max_int=self._comm_cbs[1],
                               allgather_int=self._comm_cbs[2], allgatherv=self._comm_cbs[3])
        self._check(self.lib.conp_fix_set_comm(self.h, C.byref(self._comm)))

    def comm_init_rccl(self, group=None):
        """One RCCL communicator inside the library (one rank per GPU): rank 0 makes the id, torch.distributed carries its 128 bytes.
        Returns True when EVERY rank has its communicator, False when every rank is without one (all ranks take the same branch):
          1. each rank probes librccl locally (conp_rccl_available) and the ranks agree (MIN) before anything collective is
             entered -- a rank that cannot load RCCL never leaves its partners inside ncclCommInitRank;
          2. rank 0's id (or the news that it could not make one) is broadcast;
          3. after ncclCommInitRank the ranks agree again; on any failure every rank destroys its communicator."""
        import torch
        import torch.distributed as dist

        def all_ok(ok):
            dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
            flag = torch.tensor([1 if ok else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return int(flag.item()) == 1

        if not all_ok(self.lib.conp_rccl_available() == 0):
            return False
        buf = (C.c_char * 128)()
        obj = [None]
        if dist.get_rank(group) == 0:
            if self.lib.conp_rccl_unique_id(buf) == 0:
                obj = [bytes(buf.raw)]
        dist.broadcast_object_list(obj, src=0, group=group)
        if obj[0] is None:
            return False
        idb = (C.c_char * 128).from_buffer_copy(obj[0])
        rc = self.lib.conp_fix_comm_init_rccl(self.h, idb)
        if all_ok(rc == 0):
            return True
        self.lib.conp_fix_comm_destroy_rccl(self.h)
        return False

    def pin_host_arrays(self, at):
        """page-lock at.x / at.q in place (conp_fix_pin_host_arrays): the host-buffer hooks then upload by asynchronous DMA"""
        self._pinned = (at.x, at.q)                 # keep them alive and in place
        self._check(self.lib.conp_fix_pin_host_arrays(self.h, _dptr(at.x), _dptr(at.q), int(at.nlocal + at.nghost)))

    def unpin_host_arrays(self):
        self._check(self.lib.conp_fix_unpin_host_arrays(self.h))
        self._pinned = None

    # -- device-resident path --------------------------------------------------------------------
    def set_stream(self, stream_ptr: int):
        self._check(self.lib.conp_fix_set_stream(self.h, C.c_void_p(stream_ptr)))

    def bind_device_buffers(self, d_b: Optional[int], d_q: Optional[int]):
        self._check(self.lib.conp_fix_bind_device_buffers(self.h, C.c_void_p(d_b or 0), C.c_void_p(d_q or 0)))

    def row_range(self):
        a, b = C.c_int(), C.c_int()
        self._check(self.lib.conp_fix_row_range(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def b_cal_device(self, d_x: int, d_q: int):
        self._check(self.lib.conp_fix_b_cal_device(self.h, C.c_void_p(d_x), C.c_void_p(d_q)))

    def solve_device(self, potdiff):
        self._check(self.lib.conp_fix_solve_device(self.h, potdiff))

    def scatter_device(self, d_q_atoms: int, potdiff):
        self._check(self.lib.conp_fix_scatter_device(self.h, C.c_void_p(d_q_atoms), potdiff))

    def pre_force_device(self, d_x: int, d_q: int, potdiff):
        self._check(self.lib.conp_fix_pre_force_device(self.h, C.c_void_p(d_x), C.c_void_p(d_q), potdiff))

    def profile(self, enable):
        """0 off, 1 events around every kernel, 2 around the dominant kernel only (conp_hip.h)"""
        self._check(self.lib.conp_fix_profile(self.h, int(enable)))

    def profile_read(self):
        n = C.c_int()
        names = (C.c_char_p * 16)()
        ms = (C.c_double * 16)()
        cnt = (C.c_int * 16)()
        self._check(self.lib.conp_fix_profile_read(self.h, C.byref(n), names, ms, cnt))
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(n.value)}

    def write_timing(self):
        """the three timing lines of fix_conp.cpp:564-566 go to the log buffer"""
        self._check(self.lib.conp_fix_write_timing(self.h))

    def log_drain(self):
        """text the reference would have printed to its log file (fix_conp.cpp:119) since the last drain"""
        return self.lib.conp_fix_log_drain(self.h).decode()

    def mesg_drain(self):
        """the `conp output: <e,e>` / `<d,d>` lines of fix_conp.cpp:1006-1009, 458-461"""
        return self.lib.conp_fix_mesg_drain(self.h).decode()

    def close(self):
        if self.h:
            self.lib.conp_fix_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _uneven_all_gather(outs, mine, group, world):
    """all_gather with different lengths per rank: pad to the longest (the collectives want equal sizes), trim after"""
    import torch
    import torch.distributed as dist
    nmax = max(o.numel() for o in outs)
    pad = torch.zeros(nmax, dtype=torch.uint8)
    pad[: mine.numel()] = mine
    bufs = [torch.zeros(nmax, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    for r in range(world):
        outs[r].copy_(bufs[r][: outs[r].numel()])


# ---- host-only helpers (no GPU needed) -----------------------------------------------------------------------------
def host_ktables(s: "_systems.System"):
    lib = load_library()
    info = np.zeros(16, np.int32)
    args = (s.g_ewald, s.accuracy, s.slab_volfactor, int(s.slabflag), float(s.prd[0]), float(s.prd[1]), float(s.prd[2]), s.qsqsum,
            int(s.natoms), _systems.QQRD2E, 1.0)
    null_i, null_d = C.POINTER(C.c_int)(), C.POINTER(C.c_double)()
    rc = lib.conp_host_ktables(*args, _iptr(info), null_i, null_i, null_i, null_d, null_i, null_i, null_i, null_i, null_i)
    if rc:
        raise ConpError(rc, lib.conp_last_error().decode())
    K, E = int(info[0]), max(int(info[2]), 1)
    out = dict(kxvecs=np.zeros(K, np.int32), kyvecs=np.zeros(K, np.int32), kzvecs=np.zeros(K, np.int32), ug=np.zeros(K),
               kxy_list=np.zeros(E, np.int32), kz_list=np.zeros(E, np.int32), plan_p=np.zeros(K, np.int32),
               plan_m=np.zeros(K, np.int32), plan_sign=np.zeros(K, np.int32))
    rc = lib.conp_host_ktables(*args, _iptr(info), _iptr(out["kxvecs"]), _iptr(out["kyvecs"]), _iptr(out["kzvecs"]), _dptr(out["ug"]),
                               _iptr(out["kxy_list"]), _iptr(out["kz_list"]), _iptr(out["plan_p"]), _iptr(out["plan_m"]),
                               _iptr(out["plan_sign"]))
    if rc:
        raise ConpError(rc, lib.conp_last_error().decode())
    out["kxy_list"] = out["kxy_list"][:int(info[2])]; out["kz_list"] = out["kz_list"][:int(info[2])]
    out.update(kcount=K, kcount_flat=int(info[1]), kcount_expand=int(info[2]), kxmax=int(info[3]), kymax=int(info[4]),
               kzmax=int(info[5]), kmax=int(info[6]), kmax3d=int(info[7]), kcount_dims=info[8:15].copy(), n_planar=int(info[15]))
    return out


def host_index(tag0, echeck0, tag1=None, echeck1=None):
    lib = load_library()
    tag0 = np.ascontiguousarray(tag0, np.int32); echeck0 = np.ascontiguousarray(echeck0, np.int32)
    if tag1 is None:
        tag1, echeck1, n1 = tag0, echeck0, 0
    else:
        tag1 = np.ascontiguousarray(tag1, np.int32); echeck1 = np.ascontiguousarray(echeck1, np.int32); n1 = len(tag1)
    ne = int((echeck0 != 0).sum()); mt = int(tag0.max())
    sizes = np.zeros(4, np.int32)
    m = dict(ele2tag=np.zeros(ne, np.int32), ele2eleall=np.zeros(ne, np.int32), eleall2tag=np.zeros(ne, np.int32),
             eleall2ele=np.zeros(ne + 1, np.int32), elebuf2eleall=np.zeros(ne, np.int32), tag2eleall=np.zeros(mt + 1, np.int32))
    rc = lib.conp_host_index(len(tag0), _iptr(tag0), _iptr(echeck0), n1, _iptr(tag1), _iptr(echeck1), _iptr(sizes),
                             *[_iptr(m[k]) for k in ("ele2tag", "ele2eleall", "eleall2tag", "eleall2ele", "elebuf2eleall", "tag2eleall")])
    if rc:
        raise ConpError(rc, lib.conp_last_error().decode())
    m["sizes"] = sizes
    return m


def host_pair_rows(which, lst, at, newton=False):
    lib = load_library()
    x = np.ascontiguousarray(at.x, dtype=np.float64)
    neigh = lst.neigh if lst.neigh.size else np.zeros(1, np.int32)
    av = conp_atoms(nlocal=at.nlocal, nghost=at.nghost, x=_dptr(x), q=_dptr(at.q), type=_iptr(at.type), tag=_iptr(at.tag),
                    echeck=_iptr(at.echeck))
    lv = conp_neighlist(inum=lst.inum, ilist=_iptr(lst.ilist), numneigh=_iptr(lst.numneigh), first=_iptr(lst.first),
                        neigh=_iptr(neigh), nneigh=int(lst.neigh.size))
    null_i = C.POINTER(C.c_int)()
    n = lib.conp_host_pair_rows(which, C.byref(lv), C.byref(av), int(newton), null_i, null_i, null_i, null_i)
    if n < 0:
        raise ConpError(-2, lib.conp_last_error().decode())
    ne = int((at.echeck[:at.nlocal] != 0).sum())
    out = dict(row_ptr=np.zeros(ne + 1, np.int32), ele_atom=np.zeros(max(n, 1), np.int32), oth_atom=np.zeros(max(n, 1), np.int32),
               col=np.zeros(max(n, 1), np.int32))
    lib.conp_host_pair_rows(which, C.byref(lv), C.byref(av), int(newton), _iptr(out["row_ptr"]), _iptr(out["ele_atom"]),
                            _iptr(out["oth_atom"]), _iptr(out["col"]))
    for k in ("ele_atom", "oth_atom", "col"):
        out[k] = out[k][:n]
    out["npairs"] = int(n)
    return out
