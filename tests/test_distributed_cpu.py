"""world_size-2 / -3 gloo test (CPU): the product's decomposition -- every rank contracts ITS share of the electrolyte atoms for all
k-vectors and projects on all electrode rows (conp_amd/distributed.py atom_shard = conp_fix.cpp build_items), rows of the solve
sharded -- and its two collectives reproduce the single-rank charge update.  The per-rank arithmetic comes from the CPU oracle (the
HIP path cannot run here); the choreography is the one bench.py uses on GPUs (conp_amd/distributed.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "lammps-user-conp2_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    from conp_amd import neighbor, systems
    from conp_amd.distributed import atom_shard, row_range, sharded_update
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = oracle_py.load()
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0, mode="slab")
    at, alist, blist = neighbor.build_lists(s)
    q = at.q.copy()
    at_o = neighbor.Atoms(nlocal=at.nlocal, nghost=at.nghost, x=at.x, q=q, type=at.type, tag=at.tag, echeck=at.echeck,
                          owner=at.owner)
    fo = oracle_py.Fix(lib, s)
    fo.set_atoms(at_o); fo.set_lists(alist, blist); fo.post_neighbor()
    assert fo.linalg_setup() == 0
    S = fo.matrix()
    _, _, setq = fo.vectors()
    ks = fo.ks
    ne = fo.sizes()["elenum_all"]
    csk, snk = fo.trig()
    # this rank's share of the compact electrolyte list (km_ewald.cpp:686: electrode_check == 0 && q != 0, local order): the
    # structure factors of ITS atoms only -- the other atoms' charges read as zero, which is exactly the list filter
    elyte = np.nonzero((at.echeck[:at.nlocal] == 0) & (q[:at.nlocal] != 0))[0]
    j0, j1 = atom_shard(len(elyte), rank, world)
    q_mine = np.zeros_like(q)
    q_mine[elyte[j0:j1]] = q[elyte[j0:j1]]
    sr, si = ks.sincos_b(at.x, q_mine, at.echeck, at.nlocal)
    r0, r1 = row_range(ne, rank, world)
    m = fo.maps()

    class Backend:
        def b_local(self):
            bk = ks.bbb(csk, snk, sr, si)               # all k-vectors, all rows, this rank's atoms
            if rank == 0 and s.slabflag:
                xele = np.zeros((ne, 3)); loc = {int(t): i for i, t in enumerate(at.tag[:at.nlocal])}
                for ia, t in enumerate(m["eleall2tag"]):
                    xele[ia] = at.x[loc[int(t)]]
                lib.orc_slabcorr(ks.h, at.nlocal, np.ascontiguousarray(at.x), q, at.echeck, ne, xele, bk)
            breal = np.zeros(ne); breal[m["ele2eleall"]] = fo.blist_only()
            bk[r0:r1] += breal[r0:r1]                                                # this rank's rows of the real-space term
            return torch.from_numpy(bk.copy())

        def solve_rows(self, b):
            return torch.from_numpy(S[r0:r1] @ b.numpy())

        def finish(self, q_all):
            self.q = q_all.numpy() + s.potdiff * setq

    be = Backend()
    b, q_all = sharded_update(be, ne, rank, world)
    # single-rank reference
    fo.pre_force(s.potdiff)
    b_ref, q_ref, _ = fo.vectors()
    out[rank] = (float(np.abs(b.numpy() - b_ref).max() / np.abs(b_ref).max()),
                 float(np.abs(q_all.numpy() - q_ref).max() / np.abs(q_ref).max()), len(elyte), j0, j1)
    fo.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_update_matches_single_rank(world):
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + os.getpid() % 2000 + world
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        eb, eq, n_elyte, j0, j1 = out[rank]
        assert eb < 1e-12 and eq < 1e-11, (rank, eb, eq)
    # the ranks' atom ranges tile the compact list exactly once, in rank order
    assert out[0][3] == 0 and out[world - 1][4] == out[0][2]
    for rank in range(1, world):
        assert out[rank][3] == out[rank - 1][4]
