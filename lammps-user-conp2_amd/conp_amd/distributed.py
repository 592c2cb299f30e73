"""One charge update across N processes (one GPU each): the exchange choreography of SURVEY.md 8e.

    b_local  = this rank's k-shard of the k-space b (all rows) + its rows of the real-space b (+ slab on rank 0)
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


def my_row_tiles(n_row_tiles: int, rank: int, world: int, costs=None):
    """Row tiles (rings of planar k-vectors) go heaviest first to the least loaded rank (lowest rank on ties) -- the rule of
    conp_fix.cpp km_conp_setup, where a tile's cost is its number of active kz blocks.  With equal costs (the default here:
    the CPU model of the exchange only needs SOME disjoint cover, b is a sum over tiles) this is round-robin."""
    costs = [1] * n_row_tiles if costs is None else list(costs)
    order = sorted(range(n_row_tiles), key=lambda t: (-costs[t], t))
    load = [0] * world
    owner = [0] * n_row_tiles
    for t in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[t] = r
        load[r] += costs[t]
    return [t for t in range(n_row_tiles) if owner[t] == rank]


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
