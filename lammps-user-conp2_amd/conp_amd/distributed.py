"""One charge update across N processes (one GPU each): the exchange choreography of SURVEY.md 8e.

    b_local  = the k-space b (all rows) of this rank's share of the electrolyte ATOMS (atom_shard) + its rows of the real-space b
               (+ slab on rank 0)
    b        = all_reduce(b_local, SUM)                       Ne doubles
    q[rows]  = S[rows, :] @ b                                 this rank's electrode rows
    q        = all_gather(q[rows])                            Ne doubles
    scatter  : q_i = q[e] + dV * setq[e] for every owned / ghost electrode atom

`backend` objects expose b_local() -> tensor[Ne], solve_rows(b) -> tensor[rows], finish(q_all); bench.py plugs the HIP
library in (device tensors, RCCL), the CPU test plugs in an oracle-backed model (gloo)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def row_range(ne: int, rank: int, world: int):
    """contiguous electrode-row shard in blocks of ceil(ne / world) rows; must match conp_fix.cpp (post_neighbor: rows_per)"""
    per = (ne + world - 1) // world
    r0 = min(ne, rank * per)
    return r0, min(ne, r0 + per)


def atom_shard(n_elyte: int, rank: int, world: int):
    """The k-shard of the product (conp_fix.cpp km_conp_setup + build_items, round 3 on): rank r multiplies the r-th N-th of the ATOMS
    of every (row tile, column tile) of the structure-factor contraction -- the compact list of charged electrolyte atoms
    (`electrode_check == 0 && q != 0`, km_ewald.cpp:686, in local order), cut in chunks of 16 atoms on the list padded to a multiple
    of 32; the boundaries are round(r / N * chunks), the expression both neighbours evaluate, so the ranges meet exactly.  Every rank
    projects its partial structure factors on ALL electrode rows; the all-reduce of b adds the pieces (the sum over atoms is linear).
    Returns the half-open range [j0, j1) of positions in the compact list (j1 clipped to the list's length)."""
    nl_pad = max(32, (n_elyte + 31) // 32 * 32)
    nchunks = nl_pad // 16
    # C's lround: half away from zero (Python's round() rounds half to even)
    lround = lambda v: int(v + 0.5) if v >= 0 else -int(-v + 0.5)
    c0 = lround(rank / world * nchunks)
    c1 = lround((rank + 1) / world * nchunks)
    return min(n_elyte, 16 * c0), min(n_elyte, 16 * c1)


def sharded_update(backend, ne: int, rank: int, world: int, group=None):
    b = backend.b_local()
    if world > 1:
        dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)
    q_rows = backend.solve_rows(b)
    r0, r1 = row_range(ne, rank, world)
    assert q_rows.numel() == r1 - r0
    if world > 1:
        sizes = [row_range(ne, r, world)[1] - row_range(ne, r, world)[0] for r in range(world)]
        if len(set(sizes)) == 1:
            q_all = torch.empty(ne, dtype=q_rows.dtype, device=q_rows.device)
            dist.all_gather_into_tensor(q_all, q_rows.contiguous(), group=group)
        else:   # uneven shards: pad to the largest (collectives want equal sizes), trim after the gather
            nmax = max(sizes)
            pad = torch.zeros(nmax, dtype=q_rows.dtype, device=q_rows.device)
            pad[: q_rows.numel()] = q_rows
            buf = torch.empty(world * nmax, dtype=q_rows.dtype, device=q_rows.device)
            dist.all_gather_into_tensor(buf, pad, group=group)
            q_all = torch.cat([buf[r * nmax: r * nmax + sizes[r]] for r in range(world)])
    else:
        q_all = q_rows
    backend.finish(q_all)
    return b, q_all
