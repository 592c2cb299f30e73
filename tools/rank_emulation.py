#!/usr/bin/env python3
"""Per-rank compute time of the sharded update WITHOUT the collectives: one process builds the handle of rank r of N
(conp_env.rank / nranks) on the single GPU and times b_cal_device + solve_device + scatter_device.  Used to see how the
atom-shard (structure factors) / row-shard (solve) scales before an N-GPU node is available; the two 32-KB RCCL collectives come on top.
Runs against the PRODUCT library: the three device entry points below contain no collective (the host would make them between
the calls), so no emulation switch is involved.
usage: python tools/rank_emulation.py [--workload NAME] [--json PATH] [N ...]
       --json: append {workload: {N: {rank: {ms_per_update, kernels_ms}}, ...}} to PATH (profiles/r03_rank_emulation.json)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-user-conp2_amd"))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from conp_amd import FixConp, neighbor
    argv = sys.argv[1:]
    wl = "headline"
    jpath = None
    while argv and argv[0].startswith("--"):
        if argv[0] == "--workload":
            wl, argv = argv[1], argv[2:]
        elif argv[0] == "--json":
            jpath, argv = argv[1], argv[2:]
        else:
            raise SystemExit(__doc__)
    record = {}
    s = bench.make_workload(wl)
    at, alist, blist = neighbor.build_lists(s)
    d_x = torch.from_numpy(np.ascontiguousarray(at.x)).cuda()
    d_q = torch.from_numpy(at.q.copy()).cuda()
    for n in [int(a) for a in argv] or [1, 2, 4, 8]:
        worst = None
        for rank in sorted({0, n - 1, n // 2}):
            fx = FixConp(s, device=0, rank=rank, nranks=n)
            fx.init_lists(alist, blist)
            fx.setup_post_neighbor(at)
            fx.linalg_setup(at)
            ne = fx.info().elenum_all
            d_b = torch.zeros(ne, dtype=torch.float64, device="cuda")
            d_sol = torch.zeros(ne, dtype=torch.float64, device="cuda")
            fx.bind_device_buffers(d_b.data_ptr(), d_sol.data_ptr())

            def step():
                fx.b_cal_device(d_x.data_ptr(), d_q.data_ptr())
                fx.solve_device(s.potdiff)
                fx.scatter_device(d_q.data_ptr(), s.potdiff)
            reps = 200 if wl != "big" else 10
            for _ in range(max(2, reps // 10)):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            fx.profile(True)
            for _ in range(max(3, reps // 4)):
                step()
            torch.cuda.synchronize()
            prof = {k: round(v[0], 4) for k, v in fx.profile_read().items()}
            fx.profile(False)
            print(f"N={n} rank={rank}: {ms:.4f} ms/update (compute only)  {prof}", flush=True)
            record.setdefault(str(n), {})[str(rank)] = dict(ms_per_update=ms, kernels_ms=prof)
            worst = ms if worst is None else max(worst, ms)
            fx.close()
        print(f"N={n}: slowest sampled rank {worst:.4f} ms", flush=True)
        record[str(n)]["slowest_sampled_rank_ms"] = worst
    if jpath:
        import json
        doc = {}
        if os.path.exists(jpath):
            doc = json.load(open(jpath))
        doc["_about"] = ("per-rank COMPUTE time of the atom-shard (structure factors) / row-shard (solve) update, one process per emulated rank on ONE MI355X "
                         "(tools/rank_emulation.py); the two Ne-double collectives (all-reduce b, all-gather q) are not in these "
                         "numbers; NOT a hardware scaling curve")
        doc[s.name] = record
        json.dump(doc, open(jpath, "w"), indent=1)


if __name__ == "__main__":
    main()
