"""LAMMPS-semantics ghost atoms and half neighbour lists for one MPI rank (test/bench input builder).

What the reference's loops rely on and this module reproduces:
  * atom arrays = nlocal owned atoms followed by nghost periodic-image copies that carry the owner's
    tag / type / q / group membership and shifted coordinates;
  * half list, newton off: an owned-owned pair is stored once (in the list of the lower local index), an
    owned-ghost pair is stored in the owned atom's list -- so a periodic pair shows up from both ends;
  * newton on: owned-ghost pairs are stored once (LAMMPS' coordinate tie-break rule);
  * `etypes` skip lists (fix_conp.cpp:304-361): alist keeps (ele-type, ele-type) pairs of ele-type owners,
    blist keeps pairs with exactly one ele-type member, for every owner;
  * neighbour entries may carry special-bond bits in the top two bits (NEIGHMASK = 0x3FFFFFFF).
Lists are returned in CSR form: for ii < inum, i = ilist[ii]; neighbours = neigh[first[i] : first[i]+numneigh[i]].
"""
from __future__ import annotations

import dataclasses
import itertools

import numpy as np
from scipy.spatial import cKDTree

NEIGHMASK = 0x3FFFFFFF


@dataclasses.dataclass
class Atoms:
    nlocal: int
    nghost: int
    x: np.ndarray       # [nall,3]
    q: np.ndarray       # [nall]
    type: np.ndarray    # [nall] int32
    tag: np.ndarray     # [nall] int32
    echeck: np.ndarray  # [nall] int32
    owner: np.ndarray   # [nall] int32 local index of the owning atom (ghost bookkeeping for tests)

    @property
    def nall(self):
        return self.nlocal + self.nghost


@dataclasses.dataclass
class NeighList:
    inum: int
    ilist: np.ndarray     # [inum] int32
    numneigh: np.ndarray  # [nall] int32 (0 for atoms without a list)
    first: np.ndarray     # [nall] int32 offsets into neigh
    neigh: np.ndarray     # [npairs] int32 (may carry special bits)

    @property
    def npairs(self):
        return int(self.neigh.shape[0])


def make_ghosts(sys_) -> Atoms:
    cutneigh = sys_.cutoff + sys_.skin
    prd = sys_.prd
    rng_shifts = []
    for c in range(3):
        if sys_.periodic[c]:
            m = int(np.ceil(cutneigh / prd[c]))
            rng_shifts.append(range(-m, m + 1))
        else:
            rng_shifts.append(range(0, 1))
    xs, owners = [sys_.x], [np.arange(sys_.natoms, dtype=np.int32)]
    lo, hi = sys_.boxlo - cutneigh, sys_.boxhi + cutneigh
    for s in itertools.product(*rng_shifts):
        if s == (0, 0, 0):
            continue
        xi = sys_.x + np.array(s) * prd
        keep = np.all((xi >= lo) & (xi < hi), axis=1)
        if keep.any():
            xs.append(xi[keep])
            owners.append(np.nonzero(keep)[0].astype(np.int32))
    x = np.ascontiguousarray(np.concatenate(xs))
    owner = np.concatenate(owners)
    return Atoms(nlocal=sys_.natoms, nghost=len(owner) - sys_.natoms, x=x, q=sys_.q[owner].copy(),
                 type=sys_.type[owner].copy(), tag=sys_.tag[owner].copy(), echeck=sys_.echeck[owner].copy(),
                 owner=owner)


def _half_pairs(at: Atoms, sel_a: np.ndarray, sel_b: np.ndarray | None, r: float, newton: bool) -> np.ndarray:
    """all pairs (i,j) within r with at least one owned member, oriented the LAMMPS half-list way.
    sel_a / sel_b: index arrays into the atom arrays; sel_b None -> pairs inside sel_a."""
    if sel_b is None:
        t = cKDTree(at.x[sel_a])
        p = t.query_pairs(r, output_type="ndarray")
        i, j = sel_a[p[:, 0]], sel_a[p[:, 1]]
    else:
        ta, tb = cKDTree(at.x[sel_a]), cKDTree(at.x[sel_b])
        sp = ta.sparse_distance_matrix(tb, r, output_type="ndarray")
        i, j = sel_a[sp["i"]], sel_b[sp["j"]]
    # strict '<' like the neighbour build (rsq <= cutneighsq in LAMMPS; boundary cases are measure-zero)
    nl = at.nlocal
    both_ghost = (i >= nl) & (j >= nl)
    i, j = i[~both_ghost], j[~both_ghost]
    lo, hi = np.minimum(i, j), np.maximum(i, j)   # owned index is always the smaller one for owned-ghost pairs
    if newton:
        # newton on: an owned-ghost pair is kept by exactly one of its two owners (LAMMPS half/bin/newton uses a
        # coordinate tie-break; any rule that keeps exactly one copy is equivalent for the sums computed here)
        ghost = hi >= nl
        xi, xj = at.x[lo], at.x[hi]
        keep = ~ghost | (xj[:, 2] > xi[:, 2]) | ((xj[:, 2] == xi[:, 2]) & ((xj[:, 1] > xi[:, 1]) | (
            (xj[:, 1] == xi[:, 1]) & (xj[:, 0] > xi[:, 0]))))
        lo, hi = lo[keep], hi[keep]
    return np.stack([lo, hi], 1)


def _csr(at: Atoms, pairs: np.ndarray, ilist: np.ndarray, rng: np.random.Generator | None,
         special_frac: float = 0.0) -> NeighList:
    nall = at.nall
    order = np.lexsort((pairs[:, 1], pairs[:, 0]))
    if rng is not None:   # bin-traversal order is arbitrary in LAMMPS: shuffle inside each row
        order = np.lexsort((rng.random(len(pairs)), pairs[:, 0]))
    pairs = pairs[order]
    numneigh = np.bincount(pairs[:, 0], minlength=nall).astype(np.int32)
    first = np.zeros(nall, dtype=np.int32)
    first[1:] = np.cumsum(numneigh)[:-1]
    neigh = pairs[:, 1].astype(np.int32)
    if special_frac > 0 and rng is not None and len(neigh):
        bits = rng.integers(1, 4, size=len(neigh)).astype(np.int64) << 30
        mark = rng.random(len(neigh)) < special_frac
        neigh = np.where(mark, (neigh.astype(np.int64) | bits).astype(np.uint32).view(np.int32), neigh)
        neigh = np.ascontiguousarray(neigh, dtype=np.int32)
    return NeighList(inum=len(ilist), ilist=ilist.astype(np.int32), numneigh=numneigh, first=first, neigh=neigh)


def build_lists(sys_, order_seed: int | None = 1, special_frac: float = 0.0):
    """returns (atoms, alist, blist).  Without `etypes` both are the same generic half list."""
    at = make_ghosts(sys_)
    r = sys_.cutoff + sys_.skin
    rng = np.random.default_rng(order_seed) if order_seed is not None else None
    allidx = np.arange(at.nall, dtype=np.int64)
    if sys_.eletypes is None:
        pairs = _half_pairs(at, allidx, None, r, sys_.newton)
        lst = _csr(at, pairs, np.arange(at.nlocal), rng, special_frac)
        return at, lst, lst
    is_e = np.isin(at.type, np.array(sys_.eletypes))
    e_idx, l_idx = allidx[is_e], allidx[~is_e]
    pa = _half_pairs(at, e_idx, None, r, sys_.newton)
    alist = _csr(at, pa, np.nonzero(is_e[:at.nlocal])[0], rng, special_frac)
    pb = _half_pairs(at, e_idx, l_idx, r, sys_.newton)
    blist = _csr(at, pb, np.arange(at.nlocal), rng, special_frac)
    return at, alist, blist


def build_lists_decomposed(sys_, nranks: int, axis: int = 0, order_seed: int | None = 1):
    """LAMMPS' spatial decomposition into `nranks` slabs along `axis`: per rank (atoms, alist, blist).  A rank owns the atoms of
    its slab; its ghosts are every image -- periodic copies AND the unshifted atoms of other ranks -- within cutoff + skin of the
    slab.  Half lists.  newton off: an owned-ghost pair is in the list of each owner.  newton on
    (sys_.newton): it is in the list of exactly ONE of its two owners -- the coordinate tie-break of _half_pairs is the same
    comparison on both ranks (a ghost is its owner shifted by a lattice vector), so the ranks agree on who keeps the pair; what
    that rank adds to the other rank's electrode atom is the reference's newtonbuf + MPI_Allreduce route (fix_conp.cpp:1345-1361)."""
    cutneigh = sys_.cutoff + sys_.skin
    prd = sys_.prd
    shifts = []
    for c in range(3):
        if sys_.periodic[c]:
            m = int(np.ceil(cutneigh / prd[c]))
            shifts.append(range(-m, m + 1))
        else:
            shifts.append(range(0, 1))
    edges = sys_.boxlo[axis] + prd[axis] * np.arange(nranks + 1) / nranks
    which = np.clip(np.searchsorted(edges, sys_.x[:, axis], side="right") - 1, 0, nranks - 1)
    out = []
    for r in range(nranks):
        own = np.nonzero(which == r)[0]
        lo, hi = sys_.boxlo - cutneigh, sys_.boxhi + cutneigh
        lo = lo.copy(); hi = hi.copy()
        lo[axis], hi[axis] = edges[r] - cutneigh, edges[r + 1] + cutneigh
        xs, src = [sys_.x[own]], [own]
        for s in itertools.product(*shifts):
            xi = sys_.x + np.array(s) * prd
            keep = np.all((xi >= lo) & (xi < hi), axis=1)
            if s == (0, 0, 0):
                keep &= which != r
            if keep.any():
                xs.append(xi[keep]); src.append(np.nonzero(keep)[0])
        src = np.concatenate(src)
        nlocal = len(own)
        loc_of = {int(g): i for i, g in enumerate(own)}
        owner = np.array([loc_of.get(int(g), -1) for g in src], dtype=np.int32)
        at = Atoms(nlocal=nlocal, nghost=len(src) - nlocal, x=np.ascontiguousarray(np.concatenate(xs)), q=sys_.q[src].copy(),
                   type=sys_.type[src].copy(), tag=sys_.tag[src].copy(), echeck=sys_.echeck[src].copy(), owner=owner)
        rng = np.random.default_rng(order_seed + r) if order_seed is not None else None
        allidx = np.arange(at.nall, dtype=np.int64)
        if sys_.eletypes is None:
            pairs = _half_pairs(at, allidx, None, cutneigh, sys_.newton)
            lst = _csr(at, pairs, np.arange(at.nlocal), rng)
            out.append((at, lst, lst))
            continue
        is_e = np.isin(at.type, np.array(sys_.eletypes))
        e_idx, l_idx = allidx[is_e], allidx[~is_e]
        alist = _csr(at, _half_pairs(at, e_idx, None, cutneigh, sys_.newton), np.nonzero(is_e[:at.nlocal])[0], rng)
        blist = _csr(at, _half_pairs(at, e_idx, l_idx, cutneigh, sys_.newton), np.arange(at.nlocal), rng)
        out.append((at, alist, blist))
    return out
