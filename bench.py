#!/usr/bin/env python
"""bench.py -- two-site DMRG sweep time + ground-state energy/site (BASELINE.json metric).

One "step" = one full two-site DMRG sweep (MPSKit DMRG2 order, 2L-3 bond updates; ONE `htn_dmrg2_sweep` call into
libhubbardtn_hip.so) of the one-band Hubbard chain L=64, U/t=4, half filling, SU(2)xU(1)xfZ2, at bond dimension chi
(TensorKit `dim` units) after the state has been grown 16 -> ... -> chi in untimed sweeps.
Default chi = 1024: the configuration BASELINE.json's `metric` is quoted on ("L=64 chi=1024"); it fits one GPU, so N=1
runs it, and N>1 shards the effective-Hamiltonian apply over ranks (output tiles dealt over ranks inside the library +
one RCCL all-reduce per matvec) on the SAME problem => strong scaling.  `--chi 512` gives BASELINE configs[1].

`--gpus N` with N > 1 and no torchrun environment: this process starts N ranks itself (python -m torch.distributed.run,
fresh child processes, before anything here touches torch or the GPU) and relays rank 0's JSON line.
`--backend cpu` runs the same driver on the CPU baseline library over gloo (launch-path rehearsal without GPUs).

Prints ONE JSON line (rank 0).  `roofline` is for k_grouped_gemm_z (the H_eff apply launches inside the timed sweeps,
HIP events recorded by the library on its launch stream); `cpu_baseline` times the CPU baseline library
(oracle/cpu_backend: same C++ planner / sweep driver, OpenMP task-parallel host kernels, LAPACK zgesvd with one BLAS
thread per task) on whole-sweep work of the same state, all host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

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
                    help="one_band = the bench line; the others are extra measurements (BASELINE configs[3], [4])")
    ap.add_argument("--grow", type=str, default="16x8,32x4,64x4,128x2,256x2,512x2",
                    help="untimed growth schedule chi x sweeps (state preparation, loose Lanczos)")
    ap.add_argument("--grow-tol", type=float, default=1e-6)
    ap.add_argument("--profile", action="store_true", help="sync-bracketed per-stage host timers (perturbs timing)")
    ap.add_argument("--lanczos-tol", type=float, default=1e-10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=45.0,
                    help="seconds of CPU-baseline bond updates (sweep order from the same state); a sweep that does not fit is "
                         "extrapolated by the recorded per-bond work and the sample says so")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--rank-cut", type=float, default=None,
                    help="engine.rank_cut (default 0 = off, full 1e-8 parity of every kept Schmidt value)")
    ap.add_argument("--force-shard", action="store_true",
                    help="exercise the sharded-apply code path (zero y + RCCL all-reduce) even at world size 1")
    ap.add_argument("--backend", default="hip", choices=["hip", "cpu"],
                    help="cpu = rehearsal of the multi-rank launch path on the CPU baseline library over gloo (not a bench line)")
    ap.add_argument("--master-port", type=int, default=29533)
    return ap.parse_args()


def self_launch(args):
    """--gpus N outside torchrun: start N fresh ranks (children never inherit an initialised GPU: this parent has not
    imported torch) and relay their output; the JSON line comes from rank 0"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env)
    sys.exit(p.returncode)


def host_cores_info():
    """host cores this process may really use, and how that number was arrived at: affinity mask, capped by the cgroup CPU
    quota; $HTN_CPU_THREADS overrides.  A one-GPU box of this pool exposes every core of the host in the mask and sets no
    quota, but grants a 16-core share: a team of 256 OpenMP threads on it is throttled to a crawl (measured: 9 edge-bond
    updates in 50 s), so an uncapped mask wider than 32 is READ AS that 16-core share -- a rule, not a measurement, and the
    bench line says so."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(period)
    except Exception:
        pass
    info = {"affinity_cpus": aff, "cgroup_cpu_max": quota}
    if os.environ.get("HTN_CPU_THREADS"):
        info.update(cores=max(1, int(os.environ["HTN_CPU_THREADS"])), rule="HTN_CPU_THREADS")
        return info
    n = aff if quota is None else min(aff, max(1, int(quota)))
    if n > 32:
        info.update(cores=16, rule="affinity mask wider than 32 and no cgroup quota: taken as the pool's 16-core share of a one-GPU box (assumed, not measured)")
    else:
        info.update(cores=n, rule="min(affinity mask, cgroup cpu.max quota)")
    return info


def host_cores():
    return host_cores_info()["cores"]


def cpu_baseline(eng_gpu, mpo_sites, opts, budget, log):
    """the CPU baseline library on the SAME state: bond updates in sweep order until the sweep is done or the time budget
    is spent.  Returns dict(value = seconds per sweep, ...)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_ops import CpuOps                    # context provider of oracle/cpu_backend (checker / baseline only)
    from hubbardtn_amd import engine, storage
    from oracle.cpu_backend import build as cpu_build
    cinfo = host_cores_info()
    ncores = cinfo["cores"]
    lap = cpu_build.lapack_path()
    ops = CpuOps(lapack=True)
    ops.set_threads(ncores)                       # (torch's OpenMP runtime is already in the process: the env var is too late)
    sites, bonds = storage.state_dicts(eng_gpu)
    ceng = engine.DMRG2(ops, mpo_sites, bonds, [s["blocks"] for s in sites], chi_full=opts["chi"], lanczos_tol=opts["lanczos_tol"])
    L = ceng.L
    order = [(i, +1, "right" if i < L - 2 else "left") for i in range(L - 1)] + [(i, -1, "left") for i in range(L - 3, -1, -1)]
    t0 = time.perf_counter()
    done = 0
    for (i, d, pl_) in order:
        ceng.update_bond(i, d, pl_)
        done += 1
        if time.perf_counter() - t0 > budget:
            break
    dt = time.perf_counter() - t0
    st = ceng.stats
    work = lambda s: s.n_matvec * s.apply_flops + s.svd_flops
    # work of the whole sweep from the GPU run's stats of the same bonds (same state, same settings)
    gst = eng_gpu.stats[-len(order):]
    total = sum(work(s) for s in gst)
    part = sum(work(s) for s in gst[:done])
    value = dt if done == len(order) else dt * total / max(part, 1)
    log(f"cpu baseline: {done}/{len(order)} bond updates in {dt:.1f}s on {ncores} cores -> {value:.1f} s/sweep, E={st[-1].energy:.10f}")
    return {"value": value, "unit": "s", "cores": ncores, "kind": "port",
            "cores_rule": cinfo["rule"], "affinity_cpus": cinfo["affinity_cpus"], "cgroup_cpu_max": cinfo["cgroup_cpu_max"],
            "blas": (os.path.basename(lap) + " (LAPACKE zgesvd per block, 1 BLAS thread per task)") if lap else "built-in one-sided Jacobi",
            "threads": ncores, "dtype": "c128",
            "energy_last_bond": st[-1].energy,
            "sample": (f"{done} of {len(order)} bond updates of one sweep (sweep order, from the SAME chi={opts['chi']} state) by the CPU "
                       f"baseline library (oracle/cpu_backend: the product's C++ planner + sweep driver, OpenMP task-parallel "
                       f"tiles) in {dt:.1f} s" + ("" if done == len(order) else ", scaled to the sweep by recorded work n_matvec*F_apply + F_svd")),
            "restatement": "CPU restatement of the path behind the same C ABI, not MPSKit (no Julia in the image)"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    import numpy as np
    import torch
    from hubbardtn_amd import engine, models, mps

    def log(msg):
        if args.verbose and rank == 0:
            print(msg, file=sys.stderr, flush=True)

    dist = None
    if world > 1 or args.force_shard:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", str(args.master_port))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.backend == "hip":
        from hubbardtn_amd.device import HipOps
        if dist is not None:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world)
        ops = HipOps(local)
        ops.set_timing(True)
        if dist is not None:
            ops.set_comm(rank, world)                 # RCCL inside the library; the id travels through torch.distributed
        sync = torch.cuda.synchronize
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from cpu_ops import CpuOps
        ops = CpuOps()
        if dist is not None:
            dist.init_process_group("gloo", rank=rank, world_size=world)

            def allreduce(y):
                dist.all_reduce(torch.from_numpy(y.view(np.float64)))
            ops.set_exchange(rank, world, allreduce)
        sync = lambda: None
    L, t, u = args.L, [1.0], [args.U]
    if args.model == "one_band":
        mpo = models.hamiltonian(models.OB_Sim(t, u, 0.0, 1, 1, 2.0, 8), L)
    elif args.model == "one_band_nnn":          # examples/One_band.jl:25 hopping t = [1.0, 0.1] (BASELINE configs[4])
        t = [1.0, 0.1]
        mpo = models.hamiltonian(models.OB_Sim(t, u, 0.0, 1, 1, 2.0, 8), L)
    else:                                       # examples/polyacetylene.jl:29-33, L/2 cells of 2 bands
        tm = np.array([[0.000, 3.803, -0.548, 0.000], [3.803, 0.000, 2.977, -0.501]])
        Um = np.array([[10.317, 6.264, 0.000, 0.000], [6.264, 10.317, 6.162, 0.000]])
        Jm = np.array([[0.000, 0.123, 0.000, 0.000], [0.123, 0.000, 0.113, 0.000]])
        mpo = models.hamiltonian(models.MB_Sim(tm, Um, Jm, 1, 1, 2.5, 20), L // 2)
    bonds, tens = mps.random_mps(L, (L, 0), 4, seed=1234)
    eng = engine.DMRG2(ops, mpo, bonds, tens, chi_full=16, lanczos_tol=args.lanczos_tol)
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

    # ---- timed region: K calls of htn_dmrg2_sweep, barrier + device sync on both sides, max over ranks ----
    eng.stats.clear()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        E = eng.sweep()
    sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "hip" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    sweep_s = dt / args.steps

    # ---- roofline of k_grouped_gemm_z over the H_eff applies of the timed sweeps (HIP events of the library) ----
    stats = list(eng.stats)
    # the library brackets every 8th matvec launch with HIP events on its launch stream and scales each solve's sample to
    # the solve's launch count (an event marker costs ~5 us of pipeline bubble: timing every launch slowed the sweep by 3 %)
    tstats = [s for s in stats if s.matvec_ms > 0.0]
    k_ms = sum(s.matvec_ms for s in tstats)
    k_n = sum(s.n_matvec for s in tstats)
    k_fl = sum(s.n_matvec * s.apply_flops for s in tstats)
    achieved = (k_fl / world) / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    bdims = eng.bonds
    out = {
        "metric": ("DMRG sweep time (s) + GS energy/site, 1-band Hubbard L=%d chi=%d" % (L, args.chi)) if args.model != "polyacetylene"
        else "DMRG sweep time (s) + GS energy/site, polyacetylene 2-band model %d sites chi=%d" % (L, args.chi),
        "value": sweep_s, "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": sweep_s * 1e3, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
        "dtype": "c128", "data": "synthetic (seeded random initial MPS, grown and converged in the untimed phase)",
        "config": {"workload": {"one_band": f"one-band Hubbard chain L={L} U/t={args.U:g}",
                                "one_band_nnn": f"one-band Hubbard chain L={L} t=[1.0, 0.1] U/t={args.U:g}",
                                "polyacetylene": f"polyacetylene two-band model (examples/Polyacetylene.jl parameters), {L} chain sites"
                                }.get(args.model, args.model)
                               + f", half filling, fZ2xSU(2)xU(1), two-site DMRG sweep (2L-3={2 * L - 3} bond updates) "
                                 f"at chi={args.chi} (TensorKit dim units)",
                   "model": args.model, "L": L, "chi": args.chi, "krylovdim": eng.krylovdim, "lanczos_tol": args.lanczos_tol,
                   "parallelism": "sector-parallel apply x%d" % world, "backend": args.backend},
        "energy_per_site": E / L,
        "max_bond_dim": max(b.dim_full for b in bdims), "max_multiplets": max(b.multiplets for b in bdims),
        "matvecs_per_sweep": k_n / args.steps,
        "max_trunc_weight": max(s.trunc_weight for s in stats),
        "rank_cut": eng.rank_cut,
        "host_plan_s_per_sweep": sum(s.t_plan for s in stats) / args.steps,
        # host wall per stage inside the library; lanczos and svd end in a stream sync by construction, so their wall =
        # GPU time + host work of the stage; plan/theta and env are enqueue-only unless --profile adds syncs
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
                     "traffic": None, "launches": k_n, "timed_sample": "every 8th launch (HIP events), scaled per solve", "avg_launch_us": (k_ms * 1e3 / k_n) if k_n else None,
                     "flop_per_launch": (k_fl / world / k_n) if k_n else None,
                     "share_of_sweep_time": (k_ms * 1e-3 / args.steps) / sweep_s if sweep_s > 0 else None},
    }
    # Sweep-level roofline as SURVEY 8(d) defines it: sum over the bond updates of the stage bounds, divided by the sweep
    # time.  Per bond: H_eff applies n_matvec x max(F / peak_f64, B / peak_hbm); Lanczos vector algebra 16 |theta| (2 j + 4)
    # bytes for iteration j of a restart cycle; SVD LAPACK-equivalent flops / peak_f64; environment update max(F / 2 /
    # peak_f64, B / 2 / peak_hbm).  All counts come from the run's own per-bond statistics.
    PEAK_F, PEAK_B = PEAK_F64_MFMA_TFLOPS * 1e12, 8.0e12
    kd = eng.krylovdim
    b_apply = b_lan = b_svd = b_env = 0.0
    for s_ in stats:
        one = max(s_.apply_flops / world / PEAK_F, s_.apply_bytes / PEAK_B)
        b_apply += s_.n_matvec * one
        b_lan += sum(16.0 * s_.theta_size * (2 * (j % kd) + 4) for j in range(s_.n_matvec)) / PEAK_B
        b_svd += s_.svd_flops / PEAK_F
        b_env += max(0.5 * s_.apply_flops / PEAK_F, 0.5 * s_.apply_bytes / PEAK_B)
    bound = (b_apply + b_lan + b_svd + b_env) / args.steps
    out["roofline"]["sweep"] = {"frac": bound / sweep_s if sweep_s > 0 else None, "bound_s": bound,
                                "stage_bound_s": {"apply": b_apply / args.steps, "lanczos_vectors": b_lan / args.steps,
                                                  "svd": b_svd / args.steps, "env": b_env / args.steps},
                                "definition": "SURVEY 8(d): sum of per-stage roofline bounds (78.6 TFLOP/s f64, 8.0 TB/s HBM) / sweep time"}
    # HBM traffic per launch comes from OFFLINE rocprofv3 PMC passes (counters cannot be read live); the number is a committed
    # constant, not a measurement of this run: see profiles/pmc_traffic.json for how it was collected and corrected
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(f"L{L}_chi{args.chi}")
        if pmc and world == 1 and args.model == "one_band":
            out["roofline"]["traffic"] = pmc["traffic_bytes_per_launch"]
            out["roofline"]["traffic_source"] = "offline PMC (FETCH_SIZE x2 + WRITE_SIZE, separate passes), profiles/pmc_traffic.json: " \
                                                + str(pmc.get("source", "see file"))
            out["roofline"]["algorithmic_bytes_per_launch"] = sum(s.n_matvec * s.apply_bytes for s in stats) / max(k_n, 1)
    except Exception:
        pass
    log(json.dumps(out))
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "one_band" and args.backend == "hip":
        try:
            out["cpu_baseline"] = cpu_baseline(eng, mpo, {"chi": args.chi, "lanczos_tol": args.lanczos_tol}, args.cpu_budget, log)
        except Exception as exc:     # the baseline is reporting only; never lose the GPU line
            out["cpu_baseline"] = {"value": None, "unit": "s", "cores": 1, "kind": "port", "sample": f"failed: {exc!r}"}
    out["total_runtime_s"] = time.perf_counter() - t_start
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
