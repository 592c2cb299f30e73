"""Host-side stand-ins for what LAMMPS hands to `fix conp`: atoms, box, force-field constants.

Two sources of systems:
  * the geometry of the reference's own test decks (tests/golden/deck_*.npz, made by
    tests/golden/make_deck_fixtures.py from /root/reference/tests/*/data), set up the way the
    deck input scripts do (groups by molecule id, boundary, slab factor);
  * seeded synthetic graphene-electrode / ionic-liquid boxes.

Nothing here is on the measured path; it only produces inputs.
"""
from __future__ import annotations

import dataclasses
import os
from typing import Optional

import numpy as np

# `units real` constants supplied by LAMMPS (force.cpp / update.cpp)
QQR2E = 332.06371
QQRD2E = 332.06371       # dielectric = 1
QE2F = 23.060549
EVSCALE = QE2F / QQR2E   # fix_conp.cpp:412


@dataclasses.dataclass
class System:
    """One MPI rank's view of the simulation, LAMMPS conventions (owned atoms only; ghosts are
    added by conp_amd.neighbor.build_lists)."""
    name: str
    boxlo: np.ndarray            # [3]
    boxhi: np.ndarray            # [3]
    periodic: tuple              # (px, py, pz) booleans; 'boundary p p f' -> (True, True, False)
    x: np.ndarray                # [n,3] float64
    q: np.ndarray                # [n]
    type: np.ndarray             # [n] int32, 1-based
    tag: np.ndarray              # [n] int32, 1-based
    echeck: np.ndarray           # [n] int32: +1 group1 ("left"), -1 group2, 0 electrolyte (fix_conp.cpp:599-605)
    ntypes: int
    cutoff: float                # pair cutoff (cut_coul and every cutsq entry: lj/cut/coul/long with one global cutoff)
    skin: float
    g_ewald: float
    accuracy_relative: float
    slab_volfactor: float        # 1.0 unless 'kspace_modify slab 3.0'
    slabflag: int
    eta: float
    ff_flag: int                 # 0 NORMAL, 1 FFIELD, 2 NOSLAB
    zneutr: bool = False
    eletypes: Optional[tuple] = None   # 'etypes' promise, or None for the generic list
    potdiff: float = 1.0
    newton: bool = False

    @property
    def prd(self):
        return self.boxhi - self.boxlo

    @property
    def natoms(self):
        return len(self.tag)

    @property
    def accuracy(self):
        """absolute force accuracy handed to the provider (persist.log:115-116; kspace.cpp two_charge_force)"""
        return self.accuracy_relative * QQR2E

    @property
    def qsqsum(self):
        return float(np.sum(self.q * self.q))

    def cutsq_table(self):
        t = np.full((self.ntypes + 1, self.ntypes + 1), self.cutoff * self.cutoff, dtype=np.float64)
        return t

    def copy(self):
        return dataclasses.replace(self, x=self.x.copy(), q=self.q.copy(), type=self.type.copy(),
                                   tag=self.tag.copy(), echeck=self.echeck.copy(),
                                   boxlo=self.boxlo.copy(), boxhi=self.boxhi.copy())


def _golden_dir():
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.normpath(os.path.join(here, "..", "..", "tests", "golden"))


def _wrap(x, boxlo, boxhi, periodic):
    x = x.copy()
    prd = boxhi - boxlo
    for c in range(3):
        if periodic[c]:
            x[:, c] = boxlo[c] + np.mod(x[:, c] - boxlo[c], prd[c])
    return x


def _sort_like_lammps(sys_: System, seed: Optional[int]) -> System:
    """LAMMPS spatially sorts owned atoms, so local order != tag order; emulate with a seeded shuffle."""
    if seed is None:
        return sys_
    perm = np.random.default_rng(seed).permutation(sys_.natoms)
    return dataclasses.replace(sys_, x=sys_.x[perm], q=sys_.q[perm], type=sys_.type[perm], tag=sys_.tag[perm],
                               echeck=sys_.echeck[perm])


def deck(name: str, mode: str = "ffield", etypes: bool = True, shuffle_seed: Optional[int] = None,
         g_ewald: Optional[float] = None) -> System:
    """The reference's test decks.

    name: 'dilute' (tests/dilute), 'il_onelayer', 'il_twolayer' (tests/il_*), 'cond' (tests/cond: the il_onelayer data
          file and groups), 'cond2' (tests/cond2: its own data file, 2 x 1248 electrode atoms with rough inner layers,
          cutoff 15).
    mode: 'slab' (boundary p p f + kspace_modify slab 3.0), 'ffield' (p p p), 'noslab_zneutr'
          (doubled antisymmetric cell, tests/dilute/input:50-63 trial 4 / il trial 6), 'zmirror' (doubled mirrored cell of
          tests/zmirror/input, `noslab zneutr`).
    g_ewald normally comes from LAMMPS PPPM; persist.log:112 gives 0.77236341 for dilute/ffield.  For the
    IL decks the LAMMPS estimate is not stored anywhere in the reference, so SURVEY.md's 0.20693 is used.
    """
    gd = _golden_dir()
    if name == "dilute":
        d = np.load(os.path.join(gd, "deck_dilute.npz"))
        left, right, etype, cutoff, acc, skin = (81,), (82,), (3,), 4.0, 1.0e-6, 2.0
        g_default = 0.77236341
    elif name == "cond2":
        d = np.load(os.path.join(gd, "deck_cond2.npz"))
        left, right, etype, cutoff, acc, skin = (1443,), (1444,), (5,), 15.0, 1.0e-7, 2.0   # tests/cond2/input:19,29-31,33
        g_default = 0.21          # LAMMPS' own estimate is not stored in the reference; any explicit `kspace_modify gewald`
    elif name in ("il_onelayer", "il_twolayer", "cond"):
        d = np.load(os.path.join(gd, "deck_il.npz"))
        etype, cutoff, acc, skin = (5,), 16.0, 1.0e-7, 2.0
        if name in ("il_onelayer", "cond"):
            left, right = (641,), (642,)
        else:  # tests/il_twolayer/input:41-42 relabels 643->641, 644->642
            left, right = (641, 643), (642, 644)
        g_default = 0.20693
    else:
        raise ValueError(name)
    boxlo, boxhi = d["boxlo"].copy(), d["boxhi"].copy()
    x, q, typ, tag, mol = d["x"].copy(), d["q"].copy(), d["type"].copy(), d["tag"].copy(), d["mol"].copy()
    echeck = np.zeros(len(tag), dtype=np.int32)
    echeck[np.isin(mol, left)] = 1
    echeck[np.isin(mol, right)] = -1
    zneutr = False
    if mode == "slab":
        periodic, slabf, slabflag, ff = (True, True, False), 3.0, 1, 0
    elif mode == "ffield":
        periodic, slabf, slabflag, ff = (True, True, True), 1.0, 0, 1
    elif mode == "noslab_zneutr":
        # replicate 1 1 2; recentre; second copy keeps coordinates but swaps electrode roles (trial "anti")
        lz = boxhi[2] - boxlo[2]
        x2 = x.copy(); x2[:, 2] += lz
        x = np.concatenate([x, x2]); q = np.concatenate([q, q]); typ = np.concatenate([typ, typ])
        tag = np.concatenate([tag, tag + len(tag)]).astype(np.int32)
        echeck = np.concatenate([echeck, -echeck]).astype(np.int32)
        boxhi[2] = boxlo[2] + 2 * lz
        shift = -(boxlo[2] + lz)  # change_box z final -lz' /2 .. lz'/2 remap
        x[:, 2] += shift; boxlo[2] += shift; boxhi[2] += shift
        periodic, slabf, slabflag, ff, zneutr = (True, True, True), 1.0, 0, 2, True
    elif mode == "zmirror":
        # tests/zmirror/input:33-40: replicate 1 1 2, recentre on z = 0, then reflect the upper copy (z -> lz/2 - z); both
        # copies keep their electrode roles (group eleleft = molecule molleft and molleft + molmax)
        lz = boxhi[2] - boxlo[2]
        x2 = x.copy(); x2[:, 2] += lz
        x = np.concatenate([x, x2]); q = np.concatenate([q, q]); typ = np.concatenate([typ, typ])
        tag = np.concatenate([tag, tag + len(tag)]).astype(np.int32)
        echeck = np.concatenate([echeck, echeck]).astype(np.int32)
        boxhi[2] = boxlo[2] + 2 * lz
        shift = -(boxlo[2] + lz)
        x[:, 2] += shift; boxlo[2] += shift; boxhi[2] += shift
        pos = x[:, 2] >= 0.0                       # region pos block EDGE EDGE EDGE EDGE 0 EDGE
        x[pos, 2] = lz - x[pos, 2]                 # variable newz atom lz/2-z with the doubled lz
        periodic, slabf, slabflag, ff, zneutr = (True, True, True), 1.0, 0, 2, True
    else:
        raise ValueError(mode)
    x = _wrap(x, boxlo, boxhi, periodic)
    s = System(name=f"{name}:{mode}", boxlo=boxlo, boxhi=boxhi, periodic=periodic, x=x, q=q,
               type=typ.astype(np.int32), tag=tag.astype(np.int32), echeck=echeck, ntypes=int(d["ntypes"]),
               cutoff=cutoff, skin=skin, g_ewald=g_default if g_ewald is None else g_ewald,
               accuracy_relative=acc, slab_volfactor=slabf, slabflag=slabflag, eta=1.979, ff_flag=ff,
               zneutr=zneutr, eletypes=etype if etypes else None, potdiff=1.0 if name == "dilute" else 2.0)
    return _sort_like_lammps(s, shuffle_seed)


def graphene_sheet(nx: int, ny: int, z: float):
    """rectangular 4-atom graphene cell 2.4769 x 4.3 A (13 x 8 cells = the decks' 32.2 x 34.4 A layer)"""
    a, b = 32.2 / 13.0, 34.4 / 8.0
    basis = np.array([[0.0, 0.0], [0.5, 1.0 / 6.0], [0.5, 0.5], [0.0, 2.0 / 3.0]])
    cells = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij"), -1).reshape(-1, 1, 2)
    xy = ((cells + basis[None]) * np.array([a, b])).reshape(-1, 2)
    return np.concatenate([xy, np.full((len(xy), 1), z)], axis=1), nx * a, ny * b


def synthetic(n_cells_x: int = 32, n_cells_y: int = 16, lz: float = 600.0, n_elyte: int = 32768,
              layers: int = 1, cutoff: float = 16.0, accuracy_relative: float = 1e-7, g_ewald: float = 0.21218,
              mode: str = "ffield", seed: int = 12345, min_dist: float = 2.0, potdiff: float = 2.0,
              shuffle_seed: Optional[int] = None, name: Optional[str] = None) -> System:
    """Synthetic graphene-electrode / ionic-liquid box.

    Defaults = headline: 2 x 2048 electrode atoms (32 x 16 cells, 79.26 x 68.8 A), 32768 electrolyte sites,
    Lz = 600 A.  Electrolyte = 3-site cation (+0.4374, +0.1578, +0.1848) + 1-site anion (-0.78), i.e. n_elyte/4
    ion pairs, placed uniformly at random between the electrodes with a min-distance rejection against already
    placed sites of other ions (cheap grid hash); cation sites sit ~1.5-2.7 A apart like the deck's SHAKE triangle.
    """
    rng = np.random.default_rng(seed)
    zpos = lz / 2.0 - 13.125
    sheets = []
    echeck = []
    for layer in range(layers):
        sl, lx, ly = graphene_sheet(n_cells_x, n_cells_y, -zpos - 3.35 * layer)
        sr, _, _ = graphene_sheet(n_cells_x, n_cells_y, zpos + 3.35 * layer)
        sheets += [sl, sr]
        echeck += [np.full(len(sl), 1, np.int32), np.full(len(sr), -1, np.int32)]
    xe = np.concatenate(sheets)
    ee = np.concatenate(echeck)
    npairs = n_elyte // 4
    zlo_l, zhi_l = -zpos + 3.5, zpos - 3.5
    # rigid cation triangle (site offsets, A) taken to resemble the deck's bond lengths 2.7076 / 3.8213
    tri = np.array([[0.0, 0.0, 0.0], [2.7076, 0.0, 0.0], [-1.65, 3.45, 0.0]])
    cell = max(min_dist, 1e-6)
    ncx, ncy = int(np.ceil(lx / cell)), int(np.ceil(ly / cell))
    ncz = int(np.ceil((zhi_l - zlo_l) / cell)) + 2
    occ = {}
    pts = []

    def key(p):
        return (int(p[0] // cell) % ncx, int(p[1] // cell) % ncy, int((p[2] - zlo_l) // cell))

    def ok(p):
        kx, ky, kz = key(p)
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    for o in occ.get(((kx + dx) % ncx, (ky + dy) % ncy, kz + dz), ()):
                        d = p - pts[o]
                        d[0] -= lx * np.round(d[0] / lx)
                        d[1] -= ly * np.round(d[1] / ly)
                        if d @ d < min_dist * min_dist:
                            return False
        return True

    def place(sites):
        base = len(pts)
        for s in sites:
            pts.append(s)
        for k, s in enumerate(sites):
            occ.setdefault(key(s), []).append(base + k)

    xl, ql, tl = [], [], []
    for ip in range(npairs):
        for kind in (0, 1):
            for _try in range(200):
                c = np.array([rng.uniform(0, lx), rng.uniform(0, ly), rng.uniform(zlo_l + 2.0, zhi_l - 2.0)])
                if kind == 0:
                    # random rotation of the triangle
                    qv = rng.normal(size=4); qv /= np.linalg.norm(qv)
                    w, a, b, cc = qv
                    R = np.array([[1 - 2 * (b * b + cc * cc), 2 * (a * b - cc * w), 2 * (a * cc + b * w)],
                                  [2 * (a * b + cc * w), 1 - 2 * (a * a + cc * cc), 2 * (b * cc - a * w)],
                                  [2 * (a * cc - b * w), 2 * (b * cc + a * w), 1 - 2 * (a * a + b * b)]])
                    sites = [c + R @ t for t in tri]
                else:
                    sites = [c]
                if all(zlo_l < s[2] < zhi_l for s in sites) and all(ok(s.copy()) for s in sites):
                    break
            place(sites)
            if kind == 0:
                xl += sites; ql += [0.4374, 0.1578, 0.1848]; tl += [1, 2, 3]
            else:
                xl += sites; ql += [-0.78]; tl += [4]
    xl = np.array(xl)
    x = np.concatenate([xl, xe])
    q = np.concatenate([np.array(ql), np.zeros(len(xe))])
    typ = np.concatenate([np.array(tl, np.int32), np.full(len(xe), 5, np.int32)])
    ech = np.concatenate([np.zeros(len(xl), np.int32), ee])
    tag = np.arange(1, len(x) + 1, dtype=np.int32)
    boxlo = np.array([0.0, 0.0, -lz / 2.0])
    boxhi = np.array([lx, ly, lz / 2.0])
    if mode == "ffield":
        periodic, slabf, slabflag, ff = (True, True, True), 1.0, 0, 1
    elif mode == "slab":
        periodic, slabf, slabflag, ff = (True, True, False), 3.0, 1, 0
    else:
        raise ValueError(mode)
    x = _wrap(x, boxlo, boxhi, periodic)
    s = System(name=name or f"synthetic_{len(xe)}_{len(xl)}:{mode}", boxlo=boxlo, boxhi=boxhi, periodic=periodic,
               x=x, q=q, type=typ, tag=tag, echeck=ech, ntypes=5, cutoff=cutoff, skin=2.0, g_ewald=g_ewald,
               accuracy_relative=accuracy_relative, slab_volfactor=slabf, slabflag=slabflag, eta=1.979, ff_flag=ff,
               eletypes=(5,), potdiff=potdiff)
    return _sort_like_lammps(s, shuffle_seed)


def synthetic_fast(n_cells_x: int = 32, n_cells_y: int = 16, lz: float = 600.0, n_elyte: int = 32768, layers: int = 1,
                   cutoff: float = 16.0, accuracy_relative: float = 1e-7, g_ewald: float = 0.21218, mode: str = "ffield",
                   seed: int = 12345, potdiff: float = 2.0, name: Optional[str] = None) -> System:
    """Vectorised variant of `synthetic` for the large benchmark boxes: ions on a jittered lattice (so the
    2 A minimum distance holds by construction) instead of sequential rejection sampling."""
    rng = np.random.default_rng(seed)
    zpos = lz / 2.0 - 13.125
    sheets, echeck = [], []
    for layer in range(layers):
        sl, lx, ly = graphene_sheet(n_cells_x, n_cells_y, -zpos - 3.35 * layer)
        sr, _, _ = graphene_sheet(n_cells_x, n_cells_y, zpos + 3.35 * layer)
        sheets += [sl, sr]
        echeck += [np.full(len(sl), 1, np.int32), np.full(len(sr), -1, np.int32)]
    xe = np.concatenate(sheets); ee = np.concatenate(echeck)
    nions = n_elyte // 2                      # n_elyte/4 cations (3 sites) + n_elyte/4 anions (1 site)
    zlo_l, zhi_l = -zpos + 5.0, zpos - 5.0
    vol = lx * ly * (zhi_l - zlo_l)
    h = (vol / nions) ** (1.0 / 3.0)
    gx, gy, gz = max(1, int(lx / h)), max(1, int(ly / h)), max(1, int((zhi_l - zlo_l) / h))
    while gx * gy * gz < nions:
        gz += 1
    idx = rng.permutation(gx * gy * gz)[:nions]
    ix, iy, iz = idx // (gy * gz), (idx // gz) % gy, idx % gz
    hx, hy, hz = lx / gx, ly / gy, (zhi_l - zlo_l) / gz
    centres = np.stack([(ix + 0.5) * hx, (iy + 0.5) * hy, zlo_l + (iz + 0.5) * hz], 1)
    centres += rng.uniform(-0.15, 0.15, size=centres.shape) * np.array([hx, hy, hz])
    ncat = nions // 2
    tri = np.array([[0.0, 0.0, 0.0], [1.3, 0.0, 0.0], [-0.6, 1.1, 0.0]])  # compact triangle keeps sites inside the cell
    qv = rng.normal(size=(ncat, 4)); qv /= np.linalg.norm(qv, axis=1, keepdims=True)
    w, a, b, c = qv.T
    R = np.stack([np.stack([1 - 2 * (b * b + c * c), 2 * (a * b - c * w), 2 * (a * c + b * w)], -1),
                  np.stack([2 * (a * b + c * w), 1 - 2 * (a * a + c * c), 2 * (b * c - a * w)], -1),
                  np.stack([2 * (a * c - b * w), 2 * (b * c + a * w), 1 - 2 * (a * a + b * b)], -1)], 1)
    cat = centres[:ncat, None, :] + np.einsum("nij,kj->nki", R, tri)
    xl = np.concatenate([cat.reshape(-1, 3), centres[ncat:]])
    ql = np.concatenate([np.tile([0.4374, 0.1578, 0.1848], ncat), np.full(nions - ncat, -0.78)])
    tl = np.concatenate([np.tile([1, 2, 3], ncat), np.full(nions - ncat, 4)]).astype(np.int32)
    x = np.concatenate([xl, xe]); q = np.concatenate([ql, np.zeros(len(xe))])
    typ = np.concatenate([tl, np.full(len(xe), 5, np.int32)])
    ech = np.concatenate([np.zeros(len(xl), np.int32), ee])
    tag = np.arange(1, len(x) + 1, dtype=np.int32)
    boxlo = np.array([0.0, 0.0, -lz / 2.0]); boxhi = np.array([lx, ly, lz / 2.0])
    if mode == "ffield":
        periodic, slabf, slabflag, ff = (True, True, True), 1.0, 0, 1
    elif mode == "slab":
        periodic, slabf, slabflag, ff = (True, True, False), 3.0, 1, 0
    else:
        raise ValueError(mode)
    x = _wrap(x, boxlo, boxhi, periodic)
    return System(name=name or f"synthetic_{len(xe)}_{len(xl)}:{mode}", boxlo=boxlo, boxhi=boxhi, periodic=periodic,
                  x=x, q=q, type=typ, tag=tag, echeck=ech, ntypes=5, cutoff=cutoff, skin=2.0, g_ewald=g_ewald,
                  accuracy_relative=accuracy_relative, slab_volfactor=slabf, slabflag=slabflag, eta=1.979, ff_flag=ff,
                  eletypes=(5,), potdiff=potdiff)


def small_random(ne_side: int = 4, n_elyte: int = 64, seed: int = 7, mode: str = "ffield", lz: float = 40.0,
                 cutoff: float = 8.0, g_ewald: float = 0.35, accuracy_relative: float = 1e-5) -> System:
    """tiny box for fast unit tests"""
    s = synthetic(n_cells_x=ne_side, n_cells_y=max(1, ne_side // 2), lz=lz, n_elyte=n_elyte, cutoff=cutoff,
                  accuracy_relative=accuracy_relative, g_ewald=g_ewald, mode=mode, seed=seed, min_dist=1.5,
                  name=f"small_{ne_side}_{n_elyte}:{mode}")
    return s
