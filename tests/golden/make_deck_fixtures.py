#!/usr/bin/env python3
"""Turn the geometry the reference's test decks hold into compact .npz fixtures.

Run in the authoring container only (needs /root/reference):
    python tests/golden/make_deck_fixtures.py
Inputs : /root/reference/tests/{dilute,il_onelayer,cond2}/data   (LAMMPS data files, 'atom_style full')
Outputs: tests/golden/deck_dilute.npz, deck_il.npz, deck_cond2.npz  (data only: box, id, mol, type, q, x)
         il_onelayer, il_twolayer, cond and zmirror share one data file; cond2 has its own.
The known-answer numbers of tests/dilute/persist.log are written to tests/golden/dilute_persist.json.
"""
import json
import os
import re
import numpy as np

REF = "/root/reference/tests"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_data(path):
    with open(path) as fh:
        lines = fh.read().splitlines()
    box = {}
    natoms = ntypes = None
    i = 0
    while i < len(lines):
        ln = lines[i].split("#")[0].strip()
        if ln.endswith("atoms"):
            natoms = int(ln.split()[0])
        elif ln.endswith("atom types"):
            ntypes = int(ln.split()[0])
        elif ln.endswith("xlo xhi"):
            box["x"] = tuple(map(float, ln.split()[:2]))
        elif ln.endswith("ylo yhi"):
            box["y"] = tuple(map(float, ln.split()[:2]))
        elif ln.endswith("zlo zhi"):
            box["z"] = tuple(map(float, ln.split()[:2]))
        elif ln.startswith("Atoms"):
            i += 1
            while not lines[i].strip():
                i += 1
            rows = []
            for k in range(natoms):
                f = lines[i + k].split("#")[0].split()
                rows.append((int(f[0]), int(f[1]), int(f[2]), float(f[3]), float(f[4]), float(f[5]), float(f[6])))
            break
        i += 1
    rows.sort(key=lambda r: r[0])
    a = np.array(rows)
    return dict(
        boxlo=np.array([box["x"][0], box["y"][0], box["z"][0]]),
        boxhi=np.array([box["x"][1], box["y"][1], box["z"][1]]),
        tag=a[:, 0].astype(np.int32), mol=a[:, 1].astype(np.int32), type=a[:, 2].astype(np.int32),
        q=a[:, 3].astype(np.float64), x=a[:, 4:7].astype(np.float64), ntypes=np.int32(ntypes),
    )


def main():
    d = parse_data(os.path.join(REF, "dilute", "data"))
    np.savez_compressed(os.path.join(HERE, "deck_dilute.npz"), **d)
    print("dilute:", len(d["tag"]), "atoms, ntypes", d["ntypes"])
    d = parse_data(os.path.join(REF, "il_onelayer", "data"))
    np.savez_compressed(os.path.join(HERE, "deck_il.npz"), **d)
    print("il:", len(d["tag"]), "atoms, ntypes", d["ntypes"])
    d = parse_data(os.path.join(REF, "cond2", "data"))
    np.savez_compressed(os.path.join(HERE, "deck_cond2.npz"), **d)
    print("cond2:", len(d["tag"]), "atoms, ntypes", d["ntypes"])
    # known-answer material: tests/dilute/persist.log
    log = open(os.path.join(REF, "dilute", "persist.log")).read()
    g = float(re.search(r"G vector \(1/distance\) = ([0-9.eE+-]+)", log).group(1))
    grid = re.search(r"grid = (\d+) (\d+) (\d+)", log).groups()
    rows = []
    intable = False
    for ln in log.splitlines():
        if ln.startswith("Step Temp c_tempsl c_qleft c_qright c_qall"):
            intable = True
            continue
        if intable:
            f = ln.split()
            if len(f) != 6 or not f[0].isdigit():
                break
            rows.append([int(f[0])] + [float(v) for v in f[1:]])
    out = dict(
        source="tests/dilute/persist.log (lines 112-113 G vector/grid, 142-168 thermo table)",
        fix_command="fix e all conp/v4 1 1.979 81 82 -0.5 0.5 inv iter etypes 1 3 ffield  (v1.1: fix e eleleft conp 1 eleright 1.979 1.0 log etypes 1 3 ffield)",
        boundary="p p p", pair_cutoff=4.0, kspace_relative_accuracy=1.0e-6, g_ewald=g, pppm_grid=[int(v) for v in grid],
        thermo_columns=["step", "temp", "c_tempsl", "c_qleft", "c_qright", "c_qall"], thermo=rows,
    )
    with open(os.path.join(HERE, "dilute_persist.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("persist: g_ewald", g, "step0", rows[0])


if __name__ == "__main__":
    main()
