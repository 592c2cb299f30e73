"""no GPU needed: the C-ABI library loads, exports every symbol include/conp_hip.h declares, parses the fix command
like the reference constructor, and refuses to compute without a device (no CPU fallback)."""
import os
import re

import pytest

from conp_amd import capi, systems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "conp_hip.h")).read()
    declared = set(re.findall(r"^(?:int|double|void \*|void|int64_t|const char \*)\s*(conp_[a-z_0-9]+)\s*\(", hdr, re.M))
    lib = capi.load_library()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(capi.SYMBOLS)
    assert lib.conp_abi_version() == 1


def test_fix_command_parser_matches_reference_syntax():
    a = capi.parse_fix_command("e eleleft conp 5 eleright 1.979 v_v log etypes 2 5 3 ffield zneutr".split(), 5)
    assert (a.everynum, a.eta, a.potdiff_is_variable, a.ff_flag, a.zneutr) == (5, 1.979, 1, 1, 1)
    assert a.potdiff_var == b"v" and a.group2 == b"eleright" and a.logfile == b"log"
    assert a.smartlist == 1 and a.eletypenum == 2 and list(a.eletypes)[:2] == [5, 3]
    assert (a.minimizer, a.maxiter, a.tolerance) == (1, 100, 1e-6)       # fix_conp.cpp:88-90 defaults
    a = capi.parse_fix_command("e l conp 1 r 1.979 2.0 log noslab zneutr matout qinit himem nonneutral".split(), 5)
    assert (a.ff_flag, a.zneutr, a.matout, a.qinit, a.lowmem, a.nullneutral) == (2, 1, 1, 1, 0, 0)
    a = capi.parse_fix_command("e l conp 1 r 1.979 2.0 log cg maxiter 50 tol 1e-8".split(), 5)
    assert (a.minimizer, a.maxiter, a.tolerance) == (0, 50, 1e-8)
    a = capi.parse_fix_command("e l conp 1 r 1.979 2.0 log org amatrix".split(), 5)
    assert a.a_matrix_f == 1 and a.a_matrix_file == b"amatrix"


@pytest.mark.parametrize("cmd,msg", [
    ("e l conp 1 r 1.979 2.0", "too few input parameters"),                                  # fix_conp.cpp:86
    ("e l conp 1 r 1.979 2.0 log ffield noslab", "ffield and noslab cannot both be chosen"),  # :127
    ("e l conp 1 r 1.979 2.0 log noslab ffield", "ffield and noslab cannot both be chosen"),  # :131
    ("e l conp 1 r 1.979 2.0 log org a inv b", "A matrix file specified more than once"),     # :135
    ("e l conp 1 r 1.979 2.0 log org", "No A matrix filename given"),                         # :139
    ("e l conp 1 r 1.979 2.0 log etypes 1 9", "Invalid atom type in etypes"),                 # :156
    ("e l conp 1 r 1.979 2.0 log etypes", "Insufficient input entries for etypes"),           # :148
    ("e l conp 1 r 1.979 2.0 log frobnicate", "unknown option: frobnicate"),                  # :171-174
])
def test_fix_command_errors(cmd, msg):
    with pytest.raises(capi.ConpError) as e:
        capi.parse_fix_command(cmd.split(), 5)
    assert msg in str(e.value)


def test_no_cpu_fallback():
    from helpers import has_gpu
    if has_gpu():
        pytest.skip("GPU present")
    with pytest.raises(capi.ConpError) as e:
        capi.FixConp(systems.deck("dilute"))
    assert e.value.code == -3


def test_glue_driver_fails_loudly_without_a_gpu(tmp_path):
    """the C++ glue on a box without a GPU: FixConpHip::init -> conp_fix_create -> error->all, no CPU path behind it"""
    import os
    import subprocess
    from helpers import has_gpu
    if has_gpu():
        pytest.skip("GPU present")
    from conp_amd import neighbor
    from conp_amd.capi import fix_command_for
    from test_gpu_glue import DRIVER, write_case
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER), "driver"], stdout=subprocess.DEVNULL)
    s = systems.small_random(ne_side=4, n_elyte=32, lz=60.0)
    at, alist, blist = neighbor.build_lists(s)
    case = str(tmp_path / "case.txt")
    write_case(case, s, at, [alist] if alist is blist else [alist, blist], fix_command_for(s), [(0, 1.0, 0, None)])
    p = subprocess.run([DRIVER, case], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert p.returncode == 2
    assert "ERROR: " in p.stdout and "no CPU fallback" in p.stdout


def test_product_library_reads_only_operational_knobs_from_the_environment():
    """round 4's review counted 35 CONP_* comparison switches in the shipped library -- every experiment had left one behind.  Now the
    product reads operational knobs only (guard zones, graph replay, the inverse panel's limits, host threads, timing print-outs);
    what the parity tests compare with goes through conp_debug_set_paths (a bit mask, no strings); decided experiments live in the
    diagnostic build (`make -C csrc diag`).  List the CONP_[A-Z0-9_]+ strings of the built library: at most 10, all of them known."""
    import re
    from conp_amd import capi
    data = open(capi.library_path(), "rb").read()
    names = sorted({m.decode() for m in re.findall(rb"CONP_[A-Z0-9_]+", data)})
    allowed = {"CONP_GUARD", "CONP_GRAPH", "CONP_PANEL_SINGLE", "CONP_PANEL_MAXG", "CONP_PANEL_SPIN", "CONP_HOST_THREADS",
               "CONP_TIME_HOST", "CONP_TIME_REN"}
    assert len(names) <= 10, names
    assert set(names) <= allowed, sorted(set(names) - allowed)
