import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-user-conp2_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the shared libraries are build artefacts (git-ignored): a tree that was never built gets them here, the way
    # __graft_entry__.build() makes them (hipcc cross-compiles gfx950 without a GPU; about a minute)
    lib = os.path.join(ROOT, "lammps-user-conp2_amd", "conp_amd", "libconp_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "lammps-user-conp2_amd", "csrc"), "-j4"],
                              stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    return oracle_py.load()


@pytest.fixture(scope="session", autouse=True)
def _torch_owns_the_gpu_first():
    """A few -m gpu tests keep tensors on the device with torch in the same process as libconp_hip.so.  torch's bundled
    HIP runtime only finds the GPU if it initialises BEFORE the library's first HIP call (bench.py and smoke() have that
    order anyway), so do it once up front.  On the CPU box this is a no-op."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield


@pytest.fixture(autouse=True)
def _default_code_paths():
    """the test hooks of the ABI (conp_debug_set_paths) are process-wide: whatever a test selected is switched off behind it"""
    yield
    capi = sys.modules.get("conp_amd.capi")
    lib = getattr(capi, "_LIB", None) if capi else None
    if lib is not None and hasattr(lib, "conp_debug_set_paths"):
        lib.conp_debug_set_paths(0)
        lib.conp_debug_set_sk_workgroups(0)


def pytest_sessionfinish(session, exitstatus):
    """CONP_GUARD=1 python -m pytest tests -m gpu: the whole suite with guard zones around every device buffer of the library
    (conp_fix.cpp GuardZones) -- every zone, of live buffers and of every buffer released during the session, must be intact"""
    if not os.environ.get("CONP_GUARD"):
        return
    from conp_amd import capi
    lib = capi.load_library()
    lib.conp_debug_check_guards.restype = int
    bad = lib.conp_debug_check_guards()
    msg = lib.conp_last_error().decode() if bad > 0 else ""
    print(f"\nCONP_GUARD: {bad} damaged guard zone(s) {msg}")
    if bad != 0:
        session.exitstatus = 1
