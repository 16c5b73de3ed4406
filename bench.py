#!/usr/bin/env python
"""bench.py -- two-site DMRG sweep time + ground-state energy/site (BASELINE.json metric).

One "step" = one full two-site DMRG sweep (MPSKit DMRG2 order, 2L-3 bond updates) of the
one-band Hubbard chain L=64, U/t=4, half filling, SU(2)xU(1)xfZ2, at bond dimension chi
(TensorKit `dim` units) after the state has been grown 16 -> ... -> chi in untimed sweeps.
Default chi = 1024: the configuration BASELINE.json's `metric` is quoted on ("L=64 chi=1024"); it fits one
GPU, so N=1 runs it, and N>1 shards the effective-Hamiltonian apply over ranks (owner-computes over output
tiles + RCCL all-reduce) on the SAME problem => strong scaling.  `--chi 512` gives BASELINE configs[1].

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel k_grouped_gemm_z (H_eff
apply launches inside the timed sweeps, HIP events on the launch stream); `cpu_baseline` times
the oracle (numpy restatement, oracle/) on a bounded sample of the same state.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F64_MFMA_TFLOPS = 78.6      # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz (v_mfma_f64_16x16x4_f64, 64 clk)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--L", type=int, default=64)
    ap.add_argument("--U", type=float, default=4.0)
    ap.add_argument("--chi", type=int, default=1024)
    ap.add_argument("--model", default="one_band", choices=["one_band", "one_band_nnn", "polyacetylene"],
                    help="one_band = BASELINE configs[1] (default, the bench line); the others are extra measurements")
    ap.add_argument("--grow", type=str, default="16x8,32x4,64x4,128x2,256x2,512x2",
                    help="untimed growth schedule chi x sweeps (state preparation, loose Lanczos)")
    ap.add_argument("--grow-tol", type=float, default=1e-6)
    ap.add_argument("--profile", action="store_true", help="sync-bracketed per-stage host timers (perturbs timing)")
    ap.add_argument("--lanczos-tol", type=float, default=1e-10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-bonds", type=int, default=12,
                    help="centre-bond updates the numpy oracle is timed on (about 0.85 s each at chi=1024 on one BLAS thread: a ~10 s sample)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--rank-cut", type=float, default=None,
                    help="engine.rank_cut (default 0 = off, full 1e-8 parity of every kept Schmidt value): fraction of the "
                         "truncation cut below which singular directions are dropped before the Jacobi sweeps; 0.05 "
                         "trades relative accuracy of the smallest kept values (<= 1.3e-3) for ~16 %% sweep time")
    ap.add_argument("--force-shard", action="store_true",
                    help="exercise the sharded-apply code path (zero_y + all-reduce hook) even at world size 1")
    return ap.parse_args()


def cpu_baseline(eng, L, t, u, lanczos_tol, nbonds, log):
    """time the oracle on `nbonds` centre bond updates of the SAME state (envs downloaded from
    the device), extrapolated to a sweep by the recorded per-bond work of the GPU run."""
    from oracle import dmrg_su2, mpo as ompo
    mpo = ompo.hubbard_mpo(L, t, u)
    o = object.__new__(dmrg_su2.DMRG2)
    psi = dmrg_su2.MPS(L, (L, 0))
    psi.bonds = [dict(b.dims) for b in eng.bonds]
    psi.tensors = [None] * L
    o.psi, o.mpo, o.L = psi, mpo, L
    o.chi_full, o.cutoff, o.weighting = eng.chi_full, eng.cutoff, eng.weighting
    o.krylovdim, o.lanczos_tol, o.maxrestart = eng.krylovdim, lanczos_tol, eng.maxrestart
    o.Lenvs, o.Renvs = [None] * (L + 1), [None] * (L + 1)
    o.stats, o.energy = [], None
    # the engine finished a sweep at bond 0 ('left' placement): sites >= 1 are right-canonical, the
    # centre sits on site 0; move it (on the device) to the sample bond by a partial rightward pass
    i0 = L // 2 - 1
    for i in range(0, i0):
        eng.update_bond(i, +1, "right")
    for s in (i0, i0 + 1):
        psi.tensors[s] = eng.download_site(s)
    o.Lenvs[i0] = eng.download_env("L", i0)
    Rt = eng.download_env("R", i0 + 2)
    o.Renvs[i0 + 2] = {(bra, w, ket): m.T.copy() for (ket, w, bra), m in Rt.items()}
    t0 = time.perf_counter()
    E = None
    for k in range(nbonds):
        E, _ = o.update_bond(i0, +1, "left")       # centre stays put: repeatable sample
    dt = (time.perf_counter() - t0) / nbonds
    st = o.stats[-1]
    log(f"cpu oracle bond {i0 + 1}: {dt:.2f}s  E={E:.10f} nmv={st['nmv']}")
    return dt, st, i0


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    from hubbardtn_amd import engine, models, mps
    from hubbardtn_amd.device import HipOps

    def log(msg):
        if args.verbose and rank == 0:
            print(msg, file=sys.stderr, flush=True)

    shard = None
    if world > 1 or args.force_shard:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world)

        def allreduce(y):
            dist.all_reduce(torch.view_as_real(y))
        shard = (rank, world, allreduce)
    ops = HipOps(local)
    L, t, u = args.L, [1.0], [args.U]
    if args.model == "one_band":
        sim = models.OB_Sim(t, u, 0.0, 1, 1, 2.0, 8)
        mpo = models.hamiltonian(sim, L)
    elif args.model == "one_band_nnn":          # examples/One_band.jl:25 hopping t = [1.0, 0.1] (BASELINE configs[4])
        t = [1.0, 0.1]
        sim = models.OB_Sim(t, u, 0.0, 1, 1, 2.0, 8)
        mpo = models.hamiltonian(sim, L)
    else:                                       # examples/polyacetylene.jl:29-33, L/2 cells of 2 bands
        tm = np.array([[0.000, 3.803, -0.548, 0.000], [3.803, 0.000, 2.977, -0.501]])
        Um = np.array([[10.317, 6.264, 0.000, 0.000], [6.264, 10.317, 6.162, 0.000]])
        Jm = np.array([[0.000, 0.123, 0.000, 0.000], [0.123, 0.000, 0.113, 0.000]])
        sim = models.MB_Sim(tm, Um, Jm, 1, 1, 2.5, 20)
        mpo = models.hamiltonian(sim, L // 2)
        args.no_cpu_baseline = True
    bonds, tens = mps.random_mps(L, (L, 0), 4, seed=1234)
    eng = engine.DMRG2(ops, mpo, bonds, tens, chi_full=16, lanczos_tol=args.lanczos_tol, shard=shard)
    t_start = time.perf_counter()
    eng.lanczos_tol = args.grow_tol
    for item in [x for x in args.grow.split(",") if x]:
        chi, nsw = (int(v) for v in item.split("x"))
        if chi >= args.chi:
            continue
        eng.chi_full = chi
        for _ in range(nsw):
            t0 = time.perf_counter()
            E = eng.sweep()
            log(f"grow chi={eng.chi_full} E/L={E / L:.10f} {time.perf_counter() - t0:.2f}s "
                f"mv={sum(s.n_matvec for s in eng.stats[-(2 * L - 3):])}")
    eng.chi_full = args.chi
    eng.lanczos_tol = args.lanczos_tol
    if args.rank_cut is not None:
        eng.rank_cut = args.rank_cut
    eng.profile = args.profile
    for _ in range(args.warmup):
        t0 = time.perf_counter()
        E = eng.sweep()
        log(f"warmup chi={eng.chi_full} E/L={E / L:.10f} {time.perf_counter() - t0:.2f}s max chi={max(eng.bond_dims())}")

    # ---- timed region ----
    eng.stats.clear()
    ops.event_log = []
    if shard is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        E = eng.sweep()
    torch.cuda.synchronize()
    if shard is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if shard is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ev = ops.event_log
    ops.event_log = None
    sweep_s = dt / args.steps

    # ---- roofline of the dominant kernel (H_eff apply launches of k_grouped_gemm_z) ----
    # HIP events recorded by htn_lanczos_z around every matvec launch on the launch stream
    stats = eng.stats
    k_ms = sum(x[1] for x in ev if x[0] == "matvec_ms")
    k_n = sum(x[2] for x in ev if x[0] == "matvec_ms")
    k_fl = sum(s.n_matvec * s.apply_flops for s in stats)
    achieved = (k_fl / world) / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    tot_mv = sum(s.n_matvec for s in stats)
    out = {
        "metric": ("DMRG sweep time (s) + GS energy/site, 1-band Hubbard L=%d chi=%d" % (L, args.chi)) if args.model != "polyacetylene"
        else "DMRG sweep time (s) + GS energy/site, polyacetylene 2-band model %d sites chi=%d" % (L, args.chi),
        "value": sweep_s, "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": sweep_s * 1e3, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
        "dtype": "c128", "data": "synthetic",
        "config": {"workload": f"one-band Hubbard chain L={L} U/t={args.U:g} half filling, fZ2xSU(2)xU(1), "
                               f"two-site DMRG sweep (2L-3={2 * L - 3} bond updates) at chi={args.chi} "
                               "(TensorKit dim units)",
                   "model": args.model, "L": L, "chi": args.chi, "krylovdim": eng.krylovdim, "lanczos_tol": args.lanczos_tol,
                   "parallelism": "sector-parallel apply x%d" % world},
        "energy_per_site": E / L,
        "max_bond_dim": max(eng.bond_dims()), "max_multiplets": max(b.multiplets for b in eng.bonds),
        "matvecs_per_sweep": tot_mv / args.steps,
        "max_trunc_weight": max(s.trunc_weight for s in stats),
        "rank_cut": eng.rank_cut,
        "host_plan_s_per_sweep": sum(s.t_plan for s in stats) / args.steps,
        # host wall per stage; lanczos and svd end in a stream sync by construction, so their wall = GPU time + host
        # work of the stage; plan/theta and env are enqueue-only unless --profile adds syncs
        "stage_s_per_sweep": {"plan+theta": sum(s.t_plan for s in stats) / args.steps,
                              "lanczos": sum(s.t_lanczos for s in stats) / args.steps,
                              "svd+truncate": sum(s.t_svd for s in stats) / args.steps,
                              "env": sum(s.t_env for s in stats) / args.steps, "synced": bool(args.profile)},
        # SVD stage (LAPACK-equivalent flops / time, no roofline claim): 4 (4 m n^2 + 8 n^3) per block
        "svd": {"lapack_equiv_gflop_per_sweep": sum(s.svd_flops for s in stats) / args.steps / 1e9,
                "achieved_tflops": sum(s.svd_flops for s in stats) / max(sum(s.t_svd for s in stats), 1e-9) / 1e12,
                "max_jacobi_sweeps": max(s.jacobi_sweeps for s in stats)},
        "roofline": {"bound": "mfma", "kernel": "k_grouped_gemm_z (H_eff apply)", "achieved": achieved,
                     "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F64_MFMA_TFLOPS,
                     "traffic": None, "launches": k_n, "avg_launch_us": (k_ms * 1e3 / k_n) if k_n else None,
                     "flop_per_launch": (k_fl / world / k_n) if k_n else None},
    }
    # HBM traffic per launch comes from offline rocprofv3 PMC passes (counters cannot be read live); see
    # profiles/pmc_traffic.json for how it was collected and corrected
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(f"L{L}_chi{args.chi}")
        if pmc and world == 1 and args.model == "one_band":
            out["roofline"]["traffic"] = pmc["traffic_bytes_per_launch"]
            out["roofline"]["algorithmic_bytes_per_launch"] = sum(s.n_matvec * s.apply_bytes for s in stats) / max(tot_mv, 1)
    except Exception:
        pass
    log(json.dumps(out))
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "one_band":
        try:
            work = sum(s.n_matvec * s.apply_flops + s.svd_flops for s in stats) / args.steps
            # the oracle's dense algebra is numpy -> OpenBLAS / LAPACK, which would silently use every host core: pin
            # it to ONE thread (the reference runs 1 BLAS thread per task, src:28-39) so that `cores` is what ran
            ncores = 1
            try:
                from threadpoolctl import threadpool_limits
                limiter = threadpool_limits(limits=1)
            except Exception:
                limiter = None
                ncores = os.cpu_count() or 1
            try:
                cdt, cst, i0 = cpu_baseline(eng, L, t, u, args.lanczos_tol, args.cpu_bonds, log)
            finally:
                if limiter is not None:
                    limiter.restore_original_limits()
            # work of the sampled bond in the same model, taken from the engine's own stats of that bond
            sb = [s for s in stats if s.bond == i0 + 1]
            wb = np.mean([s.n_matvec * s.apply_flops + s.svd_flops for s in sb])
            out["cpu_baseline"] = {"value": cdt * work / wb, "unit": "s", "cores": ncores,
                                   "kind": "port",
                                   "sample": f"{args.cpu_bonds} update(s) of centre bond {i0 + 1} of the same chi={args.chi} state "
                                             f"by the numpy oracle ({cdt:.2f} s each), scaled to a sweep by the recorded "
                                             "per-bond flops (n_matvec*F_apply + F_svd)"}
        except Exception as exc:     # the baseline is reporting only; never lose the GPU line
            out["cpu_baseline"] = {"value": None, "unit": "s", "cores": 1, "kind": "port", "sample": f"failed: {exc!r}"}
    out["total_runtime_s"] = time.perf_counter() - t_start
    if rank == 0:
        print(json.dumps(out), flush=True)
    if shard is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    main()
