#!/usr/bin/env python3
"""bench.py -- charge-solve updates/s of the MI355X constant-potential solver (BASELINE.json metric).

One "step" = one charge update of `fix conp` = b-vector assembly (Ewald structure factors of the electrolyte on the
FP64 matrix cores, projection on the electrode atoms, real-space electrode-electrolyte pairs) + q = S b (projected
inverse GEMV) + charge write, with atoms, neighbour rows and S resident in HBM.  Workload at every N: the synthetic
graphene-electrode / ionic-liquid box the metric is quoted on (4096 electrode / 32768 electrolyte atoms).
N > 1: k-vectors (planar row tiles) and electrode rows are sharded over the ranks; one all-reduce of b (Ne doubles) and
one all-gather of q (Ne doubles) per update, made INSIDE the library on its own RCCL communicator (conp_fix_comm_init_rccl);
the once-per-run A build is sharded by tiles and summed over RCCL, every rank keeps only its rows of the projected inverse
-> "strong" scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline|headline_slab|big|il_onelayer|il_twolayer|dilute|cond2] [--no-configs] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lammps-user-conp2_amd"))

import numpy as np  # noqa: E402

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector = matrix peak (AMD spec); 77.8 measured with tools/microbench/mfma_f64_bench.hip
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md


def make_workload(name):
    from conp_amd import systems
    if name == "headline":
        return systems.synthetic_fast(n_cells_x=32, n_cells_y=16, lz=600.0, n_elyte=32768, cutoff=16.0,
                                      accuracy_relative=1e-7, g_ewald=0.21218, mode="ffield", seed=12345,
                                      name="synthetic graphene/IL 4096 electrode + 32768 electrolyte, ffield")
    if name == "headline_slab":      # the same box in the reference's DEFAULT geometry: boundary p p f + kspace_modify slab 3.0
        return systems.synthetic_fast(n_cells_x=32, n_cells_y=16, lz=600.0, n_elyte=32768, cutoff=16.0,
                                      accuracy_relative=1e-7, g_ewald=0.21218, mode="slab", seed=12345,
                                      name="synthetic graphene/IL 4096 electrode + 32768 electrolyte, slab 3.0")
    if name == "headline_rough":     # the same box with ROUGH electrodes: every electrode atom its own z (jitter as tests/test_gpu_parity.py `rough`),
        # no z classes -> the general path of the update: partial tiles + sk_reduce + b_project, what the reference's tests/cond2 deck needs
        s = systems.synthetic_fast(n_cells_x=32, n_cells_y=16, lz=600.0, n_elyte=32768, cutoff=16.0,
                                   accuracy_relative=1e-7, g_ewald=0.21218, mode="ffield", seed=12345,
                                   name="synthetic graphene/IL 4096 rough electrode + 32768 electrolyte, ffield")
        rng = np.random.default_rng(4)
        ele = s.echeck != 0
        s.x[ele, 2] += rng.uniform(-0.05, 0.05, size=int(ele.sum()))
        return s
    if name == "big":
        return systems.synthetic_fast(n_cells_x=64, n_cells_y=32, lz=1200.0, n_elyte=262144, cutoff=12.0,
                                      accuracy_relative=1e-6, g_ewald=0.2554, mode="ffield", seed=12345,
                                      name="synthetic graphene/IL 16384 electrode + 262144 electrolyte, ffield")
    if name in ("il_onelayer", "il_twolayer", "dilute", "cond2"):      # the reference's own decks (tests/*)
        return systems.deck(name, "ffield")
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(s, at, alist, blist, S_matrix, threads):
    """the oracle (scalar port of km_ewald.cpp sincos_b + bbb_from_sincos_b, fix_conp.cpp blist_coul_cal + the ddot GEMV)
    timed on this host on ONE full charge update of the same workload"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    lib = oracle_py.load(fast=True)
    threads = lib.orc_set_threads(threads)
    ks = oracle_py.KSpace.from_system(lib, s)
    fo = oracle_py.Fix(lib, s)
    q = at.q.copy()
    from conp_amd import neighbor
    at_o = neighbor.Atoms(nlocal=at.nlocal, nghost=at.nghost, x=at.x, q=q, type=at.type, tag=at.tag, echeck=at.echeck,
                          owner=at.owner)
    fo.set_atoms(at_o); fo.set_lists(alist, blist); fo.post_neighbor()
    m = fo.maps()
    xele = np.zeros((len(m["eleall2tag"]), 3))
    loc = {int(t): i for i, t in enumerate(at.tag[:at.nlocal])}
    for ia, t in enumerate(m["eleall2tag"]):
        xele[ia] = at.x[loc[int(t)]]
    csk, snk = ks.ele_trig(xele)       # setup, not timed (reference: a_read, once per run)
    t0 = time.perf_counter()
    sr, si = ks.sincos_b(at.x, q, at.echeck, at.nlocal)
    t1 = time.perf_counter()
    b = ks.bbb(csk, snk, sr, si)
    t2 = time.perf_counter()
    breal = fo.blist_only()
    t3 = time.perf_counter()
    y = np.zeros(len(b))
    lib.orc_gemv_rows(len(b), np.ascontiguousarray(S_matrix), b, y)
    t4 = time.perf_counter()
    total = t4 - t0
    fo.close()
    return dict(value=1.0 / total, unit="updates/s", cores=threads, kind="port",
                sample=f"1 full charge update of the same workload ({total:.2f} s: S(k) {t1 - t0:.2f}, k-space b {t2 - t1:.2f}, "
                       f"real-space b {t3 - t2:.3f}, GEMV {t4 - t3:.3f}); oracle/conp_oracle.c -O3 -mavx2 -mfma")


def sk_gemm_flops(info, world=1):
    """algorithmic work of the structure-factor contraction per launch (DESIGN.md "roofline"): one complex MAC per (half-space
    k, atom) with the +-kz pair sharing its products = 4 flop per k per atom (SURVEY 8d counts the reference loop's 8)"""
    return 4.0 * info.n_elyte_charged * info.kcount / world


def zn_gemm_flops(info):
    """the z-window form (conp_zn.hip): a GEMM of zn_rows x zn_cols over the electrolyte atoms -- 2 flop per (row, column, atom)"""
    return 2.0 * info.zn_rows * info.zn_cols * info.n_elyte_charged


def composite_roofline(info, ms_per_step, world=1, pppm=False):
    """T_roof = sum over the kernels of max(algorithmic flops / FP64 peak, algorithmic bytes / HBM peak)"""
    ne = info.elenum_all
    t_roof = {
        "structure_factors_mfma": 0.0 if pppm else sk_gemm_flops(info, world) / (FP64_PEAK_TFLOPS * 1e12),
        "b_projection_hbm": 0.0 if pppm else 8.0 * 2.0 * info.kcount_flat * ne / world / (HBM_PEAK_GBS * 1e9),
        "b_real_space_hbm": info.n_blist_pairs * (8 + 32) / world / (HBM_PEAK_GBS * 1e9),
        "gemv_hbm": 8.0 * ne * ne / world / (HBM_PEAK_GBS * 1e9),
    }
    if pppm:
        t_roof["pppm_mesh_hbm"] = 16.0 * pppm[0] * pppm[1] * pppm[2] * 6 / (HBM_PEAK_GBS * 1e9)      # 16 N x ~6 passes
    t_roof_ms = 1e3 * sum(t_roof.values())
    zn = getattr(info, "zn_cols", 0) > 0 and not pppm
    if zn:
        # the z-window form does NOT execute the survey's 4 Nl K flop: its own count bounds it (a fraction above 1 against the survey's
        # count would only say that the algorithm changed)
        t_roof["structure_factors_mfma"] = zn_gemm_flops(info) / world / (FP64_PEAK_TFLOPS * 1e12)
        t_roof_ms = 1e3 * sum(t_roof.values())
    return dict(t_roof_ms=t_roof_ms, frac=t_roof_ms / ms_per_step, parts_ms={k: 1e3 * v for k, v in t_roof.items()},
                formulation="z-window (type-2 NUFFT along z, conp_zn.hip)" if zn else "full (planar, kz) contraction",
                survey_structure_factor_roof_ms=None if pppm else 1e3 * sk_gemm_flops(info, world) / (FP64_PEAK_TFLOPS * 1e12),
                note="sum of per-kernel max(algorithmic flops / 78.6 TF, algorithmic bytes / 8 TB/s); collectives excluded"
                     + ("; PPPM: mesh of 16 N bytes x 6 passes (SURVEY 8d)" if pppm else ""))


def measure_config(label, workload, solver="inv", pppm=None, steps=200, warmup=20, dev_index=0):
    """one BASELINE config on one GPU, device-resident like the headline: setup, a per-kernel pass (HIP events around every
    kernel), warm-up, K timed updates.  Returns the entry of the JSON line's `configs` block."""
    import torch
    from conp_amd import FixConp, neighbor
    t0 = time.perf_counter()
    s = make_workload(workload)
    at, alist, blist = neighbor.build_lists(s)
    extra = (["pppm"] if pppm else []) + (["cg"] if solver == "cg" else [])
    fx = FixConp(s, device=dev_index, extra_args=extra, pppm_mesh=tuple(pppm) if pppm else None)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    fx.linalg_setup(at)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    info = fx.info()
    d_x = torch.from_numpy(np.ascontiguousarray(at.x)).cuda()
    d_q = torch.from_numpy(at.q.copy()).cuda()

    def run(n):
        for _ in range(n):
            fx.pre_force_device(d_x.data_ptr(), d_q.data_ptr(), s.potdiff)

    fx.profile(1)
    run(max(50, steps // 2))
    torch.cuda.synchronize()
    prof = fx.profile_read()
    fx.profile(0)
    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    info = fx.info()                      # (cg_iterations of the last solve)
    kernels = {k: round(v[0], 5) for k, v in prof.items() if k not in ("a_kspace", "a_real", "inverse")}
    dom = max(kernels, key=kernels.get) if kernels else None
    dominant = None
    if dom is not None:
        t_ms = kernels[dom]
        ne = info.elenum_all
        if dom == "sk_gemm":
            ach = sk_gemm_flops(info) / (t_ms * 1e-3) / 1e12
            dominant = dict(kernel="sk_gemm_kernel", ms=t_ms, bound="mfma", achieved=ach, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=ach / FP64_PEAK_TFLOPS)
        elif dom == "zn_gemm":
            ach = zn_gemm_flops(info) / (t_ms * 1e-3) / 1e12
            dominant = dict(kernel="zn_gemm_kernel", ms=t_ms, bound="mfma", achieved=ach, peak=FP64_PEAK_TFLOPS,
                            unit="TFLOP/s", frac=ach / FP64_PEAK_TFLOPS, executed_flops=zn_gemm_flops(info),
                            survey_count_tflops=2 * sk_gemm_flops(info) / (t_ms * 1e-3) / 1e12,
                            note="the z-window form: 2 x rows x window columns x atoms flop (what the kernel executes); the survey's "
                                 "8 Nl K over the same time is given for reference only -- another algorithm for the same class table")
        elif dom in ("gemv_charge", "gemv", "cg"):
            # inverse: the matrix once; CG: the matrix once per iteration (fix_conp.cpp:864-930)
            passes = max(1, int(info.cg_iterations)) if dom == "cg" else 1
            gbs = 8.0 * ne * ne * passes / 1e9 / (t_ms * 1e-3)
            dominant = dict(kernel=dom, ms=t_ms, bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                            matrix_passes=passes)
        elif dom == "pppm_b" and pppm:
            # SURVEY 8d, PPPM b: the complex mesh (16 N bytes) moved ~6 times (spread target, two forward passes, Green's function
            # in place, two backward passes, gather source)
            gbs = 16.0 * pppm[0] * pppm[1] * pppm[2] * 6 / 1e9 / (t_ms * 1e-3)
            dominant = dict(kernel="pppm_b (pppm_fft_xy / pppm_fft / pppm_gather launches)", ms=t_ms, bound="hbm", achieved=gbs,
                            peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS, mesh_passes=6,
                            note="mesh of 16 N bytes x 6 passes (SURVEY 8d) over the HIP-event time of the mesh launches; a mesh "
                                 "of a few MB sits in L2 / Infinity Cache: the launches are latency-bound, the fraction says how far")
        else:
            # b projection + row assembly (lowmem row): max(7 Ne K flop at the FP64 peak, 16 Ne kflat + 24 K + 8 Ne bytes)
            k_, kf = info.kcount, info.kcount_flat
            t_roof = max(7.0 * ne * k_ / (FP64_PEAK_TFLOPS * 1e12), (16.0 * ne * kf + 24.0 * k_ + 8.0 * ne) / (HBM_PEAK_GBS * 1e9))
            dominant = dict(kernel=dom, ms=t_ms, bound="mfma", achieved=7.0 * ne * k_ / (t_ms * 1e-3) / 1e12, peak=FP64_PEAK_TFLOPS,
                            unit="TFLOP/s", frac=t_roof / (t_ms * 1e-3),
                            note="b projection (SURVEY 8d lowmem row: 7 Ne K flop, 16 Ne kflat + 24 K + 8 Ne bytes): roofline time "
                                 "over the HIP-event time of the launches that form b from the structure factors")
    out = dict(workload=s.name, Ne=int(info.elenum_all), Nl=int(info.n_elyte_charged), K=int(info.kcount),
               mode="ffield" if s.ff_flag == 1 else "slab", solver=solver,
               kspace=("pppm %dx%dx%d order 5" % tuple(pppm)) if pppm else "ewald",
               ms_per_update=ms, updates_per_s=1e3 / ms, steps=steps, warmup=warmup, kernels_ms=kernels, dominant_kernel=dominant,
               composite_roofline=composite_roofline(info, ms, pppm=tuple(pppm) if pppm else None), setup_s=t_setup)
    fx.close()
    return label, out


# every BASELINE.json config that fits one GPU, measured AFTER the headline's timed region (so `value` is untouched)
AUX_CONFIGS = [
    ("configs[1] il_onelayer inv", dict(workload="il_onelayer", solver="inv")),
    ("configs[2] il_twolayer cg+etypes", dict(workload="il_twolayer", solver="cg")),
    ("configs[3] il_onelayer pppm 40x45x180", dict(workload="il_onelayer", solver="inv", pppm=(40, 45, 180))),
    ("headline slab 3.0 (reference default geometry)", dict(workload="headline_slab", solver="inv", steps=100, warmup=10)),
    # the GENERAL path of the update (rough electrodes: no z classes -> partial tiles, sk_reduce, b_project; once per run the (planar, kz)
    # SYRK): the headline box with jittered electrode z, and the reference's own rough deck tests/cond2 (2 x 1248 electrode atoms)
    ("headline, rough electrodes (general projection path)", dict(workload="headline_rough", solver="inv", steps=100, warmup=10)),
    ("tests/cond2 deck (rough electrodes)", dict(workload="cond2", solver="inv")),
    # BASELINE configs[4]'s box (16384 electrode / 262144 electrolyte atoms) on ONE GPU: its 8-GPU half is the driver's scaling run
    ("configs[4] 16384/262144 on one GPU", dict(workload="big", solver="inv", steps=20, warmup=3)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="headline")
    ap.add_argument("--solver", choices=["inv", "cg"], default="inv",
                    help="inv: GEMV with the projected inverse (the headline); cg: the reference's conjugate-gradient solver "
                         "(BASELINE configs[2] runs il_twolayer with it)")
    ap.add_argument("--classic", action="store_true",
                    help="the full (planar, kz) contraction (sk_gemm) instead of the z-window form -- round 4's kernels, for comparison")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=1)
    ap.add_argument("--no-profile", action="store_true", help="do not time individual kernels with HIP events")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` block (the other BASELINE configs, measured after the headline's timed region)")
    ap.add_argument("--pppm", type=int, nargs=3, metavar=("NX", "NY", "NZ"), default=None,
                    help="k-space b through the PPPM mesh (`pppm` keyword, BASELINE configs[3]) with this mesh, order 5")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from conp_amd import FixConp, neighbor
    from conp_amd.distributed import sharded_update

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.classic:
        from conp_amd import capi
        capi.load_library().conp_debug_set_paths(capi.PATH_SK_CLASSIC)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # rehearsal on a one-GPU box: CONP_BENCH_REHEARSE=1 puts every rank on cuda:0 and runs the collectives over gloo
    rehearse = os.environ.get("CONP_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))

    t_setup0 = time.perf_counter()
    s = make_workload(args.workload)
    at, alist, blist = neighbor.build_lists(s)
    fx = FixConp(s, device=dev_index, rank=rank, nranks=world, extra_args=(["pppm"] if args.pppm else []) + (["cg"] if args.solver == "cg" else []),
                 pppm_mesh=tuple(args.pppm) if args.pppm else None,
                 # one rank: every ghost is a periodic image of an owned atom -- the host-buffer hooks upload the owned x, q only and
                 # rebuild the ghosts on the device (conp_env.ghost_images, what FixConpHip sets); no effect on the device-resident
                 # hooks `value` is measured on
                 ghost_images=(world == 1))
    # multi-rank: the library makes its own RCCL communicator (rank 0's unique id travels through torch.distributed) and runs
    # the collectives on its own stream, in order with its kernels.  The rehearsal on a one-GPU box cannot (RCCL refuses two
    # ranks on one device): it keeps the Python choreography over gloo.
    lib_collectives = world > 1 and not rehearse
    collectives = "none (one rank)" if world == 1 else "gloo rehearsal on one GPU (torch.distributed choreography)"
    if lib_collectives:
        # Every rank gets the library's communicator or none does: FixConp.comm_init_rccl agrees on RCCL's availability BEFORE the
        # collective ncclCommInitRank and again after it (all ranks take the same branch, no rank exits alone).  Without it the
        # run falls back to the Python choreography over torch.distributed -- and SAYS so in the JSON line ("collectives").
        lib_collectives = fx.comm_init_rccl()
        collectives = "rccl-in-library" if lib_collectives else "torch-fallback"
        if not lib_collectives:
            print(f"[bench] rank {rank}: no RCCL communicator inside the library on some rank; ALL ranks fall back to "
                  "torch.distributed collectives", file=sys.stderr)
    if world > 1 and not lib_collectives:
        fx.set_stream(torch.cuda.current_stream().cuda_stream)
    fx.init_lists(alist, blist)
    fx.setup_post_neighbor(at)
    t_a0 = time.perf_counter()
    fx.linalg_setup(at)                     # A build (MFMA SYRK + real space) + LU inverse + projection + setq: once per run
    torch.cuda.synchronize()
    t_a1 = time.perf_counter()
    info = fx.info()
    ne = info.elenum_all

    d_x = torch.from_numpy(np.ascontiguousarray(at.x)).cuda()
    d_q = torch.from_numpy(at.q.copy()).cuda()
    d_b = torch.zeros(ne, dtype=torch.float64, device="cuda")
    d_sol = torch.zeros(ne, dtype=torch.float64, device="cuda")
    if not lib_collectives:
        fx.bind_device_buffers(d_b.data_ptr(), d_sol.data_ptr())
    row0, row1 = fx.row_range()
    potdiff = s.potdiff

    class HipBackend:
        """plugs the HIP library into conp_amd.distributed.sharded_update (device tensors, RCCL collectives)"""

        def b_local(self):
            fx.b_cal_device(d_x.data_ptr(), d_q.data_ptr())       # this rank's atoms' share of the k-space b for all rows + own real-space rows -> d_b
            return d_b.cpu() if rehearse else d_b

        def solve_rows(self, b):
            if rehearse:
                d_b.copy_(b)
            fx.solve_device(potdiff)                              # own rows of S b -> d_sol[row0:row1]
            return d_sol[row0:row1].cpu() if rehearse else d_sol[row0:row1]

        def finish(self, q_all):
            if q_all.data_ptr() != d_sol.data_ptr():
                d_sol.copy_(q_all)
            fx.scatter_device(d_q.data_ptr(), potdiff)

    backend = HipBackend()

    def step():
        if world == 1 or lib_collectives:
            fx.pre_force_device(d_x.data_ptr(), d_q.data_ptr(), potdiff)      # N > 1: all-reduce(b) / all-gather(q) inside, on RCCL
        else:
            sharded_update(backend, ne, rank, world)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- auxiliary passes FIRST (not `value`): they also let the chip's clock governor settle -- a rocprofv3 trace of this command
    #      shows sk_gemm going from 274 to 242 us over the first ~65 launches after the setup phase (profiles/r02_*, r03_*), so a
    #      short timed loop started cold would measure the ramp, not the update.
    # (a) the host-buffer hook (what FixConpHip::pre_force calls): H2D of x, q and D2H of the charges every update.
    #     PCIe-inclusive, reported beside `value`, never as `value`.
    #     Twice: out of pageable host arrays (staged, blocking copy), and with the arrays page-locked in place
    #     (conp_fix_pin_host_arrays -- what the LAMMPS glue does between re-neighbourings): the second is the reported figure.
    host_ms = host_ms_pageable = None
    if world == 1:
        nh = max(10, args.steps // 4)

        def host_pass():
            for k in range(3):
                fx.pre_force(at, k, potdiff)
            t0 = time.perf_counter()
            for k in range(nh):
                fx.pre_force(at, k, potdiff)
            return (time.perf_counter() - t0) / nh * 1e3
        host_ms_pageable = host_pass()
        try:
            fx.pin_host_arrays(at)
            host_ms = host_pass()
            fx.unpin_host_arrays()
        except Exception as e:      # noqa: BLE001 -- a runtime that refuses to page-lock: the pageable figure stands
            print(f"bench: page-locking the host arrays failed ({e}); reporting the pageable rate", file=sys.stderr)
            host_ms = host_ms_pageable
    # (b) updates with a HIP-event pair around EVERY kernel on the library's stream: the per-kernel breakdown (`kernels_ms`;
    #     the event records cost host time, so this pass is not `value`)
    prof = {}
    dt_prof = float("nan")
    n_prof = max(args.steps, 100)
    if not args.no_profile:
        fx.profile(1)
        t0 = time.perf_counter()
        for _ in range(n_prof):
            step()
        fence()
        dt_prof = time.perf_counter() - t0
        prof = fx.profile_read()
        fx.profile(0)

    for _ in range(args.warmup):
        step()
    fence()
    # ---- the timed region: exactly K updates.  A HIP-event pair around every 4th launch of the dominant kernel rides along
    #      (conp_fix_profile mode 2): the roofline's launch duration is measured live inside these very K updates.
    #      (A pair around EVERY launch costs 5 us per update -- 0.3057 -> 0.3110 ms -- because it drains the queue twice.)
    if not args.no_profile:
        fx.profile(2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    timed_prof = {}
    if not args.no_profile:
        timed_prof = fx.profile_read()
        fx.profile(0)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = args.steps / dt

    if rank == 0:
        K, nl = info.kcount, info.n_elyte_charged
        # dominant kernel: the structure-factor contraction.  Algorithmic work per launch (DESIGN.md "roofline"):
        # one complex MAC per (half-space k, atom) with the +-kz pair sharing its products = 4 real FMA per (kxy,kz)
        # pair = 4 flop per k per atom.  (SURVEY 8d counts the reference loop's 16 flop per pair = 8 per k.)
        flops = sk_gemm_flops(info, world)
        roofline = None
        if "zn_gemm" in timed_prof:
            # the z-window form (conp_zn.hip): the dominant launch is the 32 / 48-column contraction (the window matrix is written by the
            # phase kernel).  Its
            # algorithmic work is ITS formulation's: 2 x rows x columns x atoms (the survey's 8 Nl K describes the reference loop, which
            # this path does not execute: a fraction against that count would exceed 1 and say nothing about the kernel)
            t_ms = timed_prof["zn_gemm"][0]
            zf = zn_gemm_flops(info) / world
            ach = zf / (t_ms * 1e-3) / 1e12
            traffic, traffic_src = None, None
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_headline_summary.json")))
            pj = cands[-1] if cands else ""
            if args.workload == "headline" and world == 1 and pj and os.path.exists(pj):
                pmc = json.load(open(pj)).get("pmc", {})
                if "zn_gemm_kernel" in pmc and "FETCH_SIZE" in pmc["zn_gemm_kernel"] and "WRITE_SIZE" in pmc["zn_gemm_kernel"]:
                    # rocprofv3 reports KiB; FETCH_SIZE x2: gfx950 counts 16-B-per-lane streams at half (MI355X_MICROARCH.md, HBM)
                    traffic = (2.0 * pmc["zn_gemm_kernel"]["FETCH_SIZE"] + pmc["zn_gemm_kernel"]["WRITE_SIZE"]) * 1024.0
                    traffic_src = f"profiles/{os.path.basename(pj)} (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
            roofline = dict(bound="mfma", kernel="zn_gemm_kernel (v_mfma_f64_16x16x4_f64)", achieved=ach,
                            peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / FP64_PEAK_TFLOPS, traffic=traffic, traffic_source=traffic_src,
                            avg_launch_ms=t_ms, launches_averaged=timed_prof["zn_gemm"][1],
                            measured="HIP events on the library's stream around every 4th zn_gemm launch of the timed region",
                            algorithmic_flops_per_launch=zf, formulation="z-window: 2 x %d rows x %d columns x %d atoms" % (info.zn_rows, info.zn_cols, nl),
                            survey_count_tflops=2 * flops / (t_ms * 1e-3) / 1e12, full_contraction_tflops=flops / (t_ms * 1e-3) / 1e12,
                            note="achieved = the flops this formulation executes over its time; full_contraction_tflops = 4 Nl K (round 4's "
                                 "sk_gemm count) over the same time, survey_count_tflops = SURVEY 8d's 8 Nl K: both exceed what any kernel "
                                 "of the full contraction could reach -- the class table is evaluated through a 32-column window instead of "
                                 "252 kz columns")
        elif "sk_gemm" in timed_prof:
            t_ms = timed_prof["sk_gemm"][0]
            ach = flops / (t_ms * 1e-3) / 1e12
            traffic, traffic_src = None, None
            # the newest committed profile of this command (profiles/rNN_bench_headline_summary.json)
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_headline_summary.json")))
            pj = cands[-1] if cands else ""
            if args.workload == "headline" and world == 1 and pj and os.path.exists(pj):
                pm = json.load(open(pj)).get("pmc", {}).get("sk_gemm_kernel", {})
                if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
                    # rocprofv3 reports KiB; FETCH_SIZE x2: gfx950 counts 16-B-per-lane streams at half (MI355X_MICROARCH.md, HBM)
                    traffic = (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0
                    traffic_src = f"profiles/{os.path.basename(pj)} (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
            roofline = dict(bound="mfma", kernel="sk_gemm_kernel (v_mfma_f64_16x16x4_f64)", achieved=ach, peak=FP64_PEAK_TFLOPS,
                            unit="TFLOP/s", frac=ach / FP64_PEAK_TFLOPS, traffic=traffic, traffic_source=traffic_src,
                            avg_launch_ms=t_ms, launches_averaged=timed_prof["sk_gemm"][1],
                            measured="HIP events on the library's stream around every 4th sk_gemm launch of the timed region",
                            algorithmic_flops_per_launch=flops, survey_count_tflops=2 * ach,
                            note="achieved uses 4 flop per (k, atom); SURVEY 8d's reference-loop count (8 per k) would double it")
        # the HBM-bound member of the update: the GEMV with the projected inverse streams 8 Ne^2 / N bytes once (HIP events of the
        # per-kernel pass; rocprofv3's kernel-only duration is ~2 us shorter, profiles/r04_bench_headline_summary.txt)
        hbm_member = None
        if "gemv_charge" in prof and prof["gemv_charge"][0] > 0:
            t_s = prof["gemv_charge"][0] * 1e-3
            gb = 8.0 * ne * ne / world / 1e9                      # SURVEY 8d's algorithmic count: the matrix once
            sym = world == 1 and ne >= 2048 and args.solver == "inv"
            # from 2048 electrode atoms up the fused solve takes the projected inverse as a symmetric matrix: packed lower-triangle
            # tiles of 128 x 128, half the bytes.  `frac` is what the launches actually move (round 4's
            # review: the survey's 8 Ne^2 over the time of kernels that read half of it is not an achieved bandwidth); the
            # survey's count is kept beside it as survey_*
            nb = (ne + 127) // 128
            gb_exec = (nb * (nb + 1) // 2) * 128 * 128 * 8.0 / 1e9 if sym else gb
            hbm_member = dict(kernel="sym_gemv_kernel + sym_finish_kernel" if sym else "gemv_finish_kernel", bound="hbm",
                              achieved=gb_exec / t_s, peak=HBM_PEAK_GBS, unit="GB/s", frac=gb_exec / t_s / HBM_PEAK_GBS,
                              bytes_per_launch=gb_exec * 1e9,
                              survey_bytes_per_launch=8.0 * ne * ne / world, survey_gbs=gb / t_s,
                              survey_frac=gb / t_s / HBM_PEAK_GBS,
                              note="achieved / frac: the bytes this formulation reads (symmetric storage: 4 Ne^2 + the diagonal "
                                   "tiles) over the HIP-event time of the launch(es); survey_*: SURVEY 8d's algorithmic 8 Ne^2 "
                                   "over the same time")
        composite = composite_roofline(info, ms_per_step, world, pppm=tuple(args.pppm) if args.pppm else None)
        out = dict(metric="charge-solve updates/sec + ns/day, 4096-atom electrode / 32k electrolyte", value=value,
                   unit="updates/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step,
                   higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload=s.name, Ne=int(ne), Nl=int(nl), K=int(K), kflat=int(info.kcount_flat),
                               box=[float(v) for v in s.prd], cutoff=s.cutoff, g_ewald=s.g_ewald,
                               accuracy_relative=s.accuracy_relative, mode="ffield" if s.ff_flag == 1 else "slab",
                               solver=args.solver, kspace=("pppm %dx%dx%d order 5" % tuple(args.pppm)) if args.pppm else "ewald",
                               blist_pairs=int(info.n_blist_pairs),
                               structure_factor_form=("z-window (%d columns, grid %d)" % (info.zn_cols, info.zn_grid)) if info.zn_cols > 0
                               else "full (planar, kz) contraction",
                               parallelism=f"S(k): electrolyte atoms sharded (all k, all rows) + all-reduce(b); solve: electrode rows sharded + all-gather(q); x{world}" + (", RCCL inside libconp_hip" if lib_collectives else "")),
                   collectives=collectives,
                   ns_per_day_solver_limited=value * 2.0 * 86400 * 1e-6,   # Nevery=1, dt = 2 fs (tests/il_onelayer/input:87)
                   ms_per_step_host_buffers_pcie=host_ms, ms_per_step_host_buffers_pcie_pageable=host_ms_pageable,
                   host_buffers_ghost_images=bool(world == 1),
                   setup_s=dict(total=t_a1 - t_setup0, a_build_inverse_setq=t_a1 - t_a0),
                   kernels_ms={k: round(v[0], 5) for k, v in prof.items()},
                   ms_per_step_profiled_pass=dt_prof / n_prof * 1e3,
                   order_of_passes=("setup; host-buffer hook pass; %d updates with events around every kernel (kernels_ms); "
                                   "%d warm-up + %d timed updates (value; events around every 4th launch of the dominant kernel only)"
                                   % (n_prof if not args.no_profile else 0, args.warmup, args.steps)),
                   roofline=roofline, roofline_hbm_member=hbm_member, composite_roofline=composite)
        if not args.no_cpu_baseline and world == 1:
            S = fx.matrix()
            base = cpu_baseline(s, at, alist, blist, S, args.cpu_threads)
            out["cpu_baseline"] = base
            out["speedup_vs_cpu_baseline"] = value / base["value"]
            # the same port with OpenMP over k rows / electrode rows on this box's CPU share (16 threads per GPU): the analogue
            # of running the reference on 16 MPI ranks ("all cores" figure)
            if args.cpu_threads == 1:
                mt = cpu_baseline(s, at, alist, blist, S, 16)
                out["cpu_baseline_16_threads"] = mt
                out["speedup_vs_cpu_baseline_16_threads"] = value / mt["value"]
        fx.close()
        fx = None
        # ---- the other BASELINE configs on this GPU (after `value` was measured; each in a handle of its own)
        if world == 1 and args.workload == "headline" and not args.no_configs:
            cfgs = {}
            for label, kw in AUX_CONFIGS:
                try:
                    k, v = measure_config(label, dev_index=dev_index, **kw)
                    cfgs[k] = v
                except Exception as e:      # noqa: BLE001 -- one config failing must not lose the headline line
                    cfgs[label] = dict(error=str(e))
            out["configs"] = cfgs
        print(json.dumps(out))
    if fx is not None:
        fx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
