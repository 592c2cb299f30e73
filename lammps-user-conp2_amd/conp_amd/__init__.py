"""conp_amd -- Python-side harness over libconp_hip.so, the MI355X-native constant-potential charge solver.

The product is the C-ABI library (include/conp_hip.h, built from ../csrc) plus the LAMMPS glue in ../lammps_glue.
This package only (a) loads the library through ctypes for tests / bench / smoke, (b) builds LAMMPS-like inputs
(systems.py, neighbor.py).  No compute happens in Python and there is no CPU fallback: loading fails loudly when the
library is missing, and every compute call fails when no gfx950 device is present.
"""
from .capi import FixConp, ConpError, load_library, library_path  # noqa: F401
