#!/usr/bin/env python3
"""PCIe-inclusive update time through the C++ glue (FixConpHip::pre_force with host arrays), without the ctypes harness:
writes the bench workload as a glue_driver case and lets the driver time N pre_force calls.
usage: python tools/glue_time.py [workload] [N]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-user-conp2_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def main():
    import bench
    from conp_amd import neighbor
    from conp_amd.capi import fix_command_for
    from test_gpu_glue import DRIVER, write_case
    wl = sys.argv[1] if len(sys.argv) > 1 else "headline"
    n = sys.argv[2] if len(sys.argv) > 2 else "200"
    s = bench.make_workload(wl)
    at, alist, blist = neighbor.build_lists(s)
    with tempfile.TemporaryDirectory() as d:
        case = os.path.join(d, "case.txt")
        write_case(case, s, at, [alist] if alist is blist else [alist, blist], fix_command_for(s), [(0, s.potdiff, 0, None)])
        p = subprocess.run([DRIVER, case], cwd=d, capture_output=True, text=True, env=dict(os.environ, GLUE_DRIVER_TIME=n))
        for line in p.stdout.splitlines():
            if line.startswith(("time_pre_force_ms", "time_post_force_ms", "ERROR", "scalar")):
                print(wl, line)
        for line in p.stderr.splitlines():          # CONP_TIME_HOST=1: the library's own breakdown of the host-buffer update
            if line.startswith("conp host-buffer"):
                print(wl, line)


if __name__ == "__main__":
    main()
