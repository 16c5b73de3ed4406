"""The product's C++ planner + sweep driver (hubbardtn_amd/csrc/htn_plan.cpp, htn_engine.cpp) executed WITHOUT a GPU on
the CPU baseline backend (oracle/cpu_backend, same C ABI): byte-identity with the Python test statement of the planner,
parity with the numpy oracle and exact diagonalisation, observables, the on-disk formats and the C-ABI-only drive of a
whole sweep.  (The `-m gpu` suite runs the same engine on the HIP kernels.)"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import ref_planner as pl
from cpu_ops import CpuOps
from hubbardtn_amd import abi, api, engine, models, mps, storage
from oracle import dmrg_su2, ed, mpo as ompo

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_r01.json")))
E_ED_L8_U4 = -4.235806999130


@pytest.fixture(scope="module")
def cpu_ops():
    return CpuOps()


def _poly():
    t = np.array([[0.000, 3.803, -0.548, 0.000], [3.803, 0.000, 2.977, -0.501]])
    U = np.array([[10.317, 6.264, 0.000, 0.000], [6.264, 10.317, 6.162, 0.000]])
    J = np.array([[0.000, 0.123, 0.000, 0.000], [0.123, 0.000, 0.113, 0.000]])
    return models.MB_Sim(t, U, J, 1, 1, 2.5, 20)


@pytest.mark.parametrize("model", ["nn", "nnn_nn", "exchange", "poly"])
def test_cxx_apply_plans_are_byte_identical_to_the_python_statement(cpu_ops, model):
    """every bond of a short chain after one sweep (ends with empty environments included): tiles, K-slab segments,
    recoupling coefficients (alpha), Z-stage size and flop counts of the H_eff apply, C++ vs tests/ref_planner.py"""
    L = 8
    H = {"nn": lambda: models.hamiltonian(models.OB_Sim([1.0], [4.0]), L),
         "nnn_nn": lambda: models.hamiltonian(models.OB_Sim([1.0, 0.3], [4.0, 0.5]), L),
         "exchange": lambda: models.hamiltonian(models.OB_Sim([1.0, 0.2], [4.0, 0.5], 0.0, [0.3, 0.1], 1, 1), L),
         "poly": lambda: models.hamiltonian(_poly(), L // 2)}[model]()
    bonds, tens = mps.random_mps(L, (L, 0), 5, seed=2)
    eng = engine.DMRG2(cpu_ops, H, bonds, tens, chi_full=60)
    eng.sweep()
    eb = eng.bonds
    # after a sweep the left environments 0..L-2 (rightward pass) and right environments 2..L exist
    for i in range(L - 1):
        tl = pl.ThetaLayout.build(eb[i], eb[i + 2])
        Ll = pl.EnvLayout.build("L", eb[i], H[i].left)
        Rl = pl.EnvLayout.build("R", eb[i + 2], H[i + 1].right)
        if eng.lib.htn_mps_env_size(eng.handle, 0, i) != Ll.size or eng.lib.htn_mps_env_size(eng.handle, 1, i + 2) != Rl.size:
            continue        # environment of an older bond table (the sweep moved on): plan not comparable
        tz, ty, zsize, nterms = pl.plan_apply(tl, Ll, Rl, H[i], H[i + 1])
        for stage, ref in ((0, tz), (1, ty)):
            tiles, segs, zs, fl = eng.plan_apply_dump(i, stage)
            if ref is None:
                assert len(tiles) == 0 and len(segs) == 0
                continue
            assert zs == zsize and fl == ref.flops, (i, stage)
            assert tiles.tobytes() == ref.tiles[:ref.ntiles].tobytes(), (i, stage)
            rs = ref.segs[:ref.nsegs]
            assert len(segs) == len(rs)
            for f in abi.SEG_DT.names:
                if f.startswith("alpha"):
                    # the Python statement evaluates the Racah sums with exact integer factorials, C++ with doubles:
                    # the recoupling coefficients agree to rounding (2 ulp), every other field bit for bit
                    assert np.abs(segs[f] - rs[f]).max() <= 1e-15 * max(np.abs(rs[f]).max(), 1.0), (i, stage, f)
                else:
                    assert np.array_equal(segs[f], rs[f]), (i, stage, f)


@pytest.mark.parametrize("name", ["L8_U4_chi64", "L12_t2_chi48"])
def test_cxx_sweep_driver_matches_oracle_golden(cpu_ops, name):
    """C++ layouts, 9j coefficients, task lists, truncation, SVD staging and sweep loop vs the oracle's golden
    energies / truncated spectra (1e-9; north_star asks 1e-8)"""
    rec = GOLD["oracle_runs"][name]
    L = rec["L"]
    bonds, tens = mps.random_mps(L, (L, 0), rec["cap"], rec["seed"])
    eng = engine.DMRG2(cpu_ops, models.hamiltonian(models.OB_Sim(rec["t"], rec["u"]), L), bonds, tens, chi_full=rec["chi"])
    for k in range(rec["sweeps"]):
        E = eng.sweep()
        assert abs(E - rec["energies"][k]) <= 1e-9 * abs(E)
    spectra = eng.spectra
    for b, s in rec["spectra_last_sweep"].items():
        for c, v in s.items():
            key = tuple(int(x) for x in c.split(","))
            assert np.abs(spectra[int(b)][key] - np.asarray(v)).max() < 1e-9
    assert eng.bond_dims() == rec["bond_dims"]
    assert eng.cache_hits > 0 and len(eng.stats) == rec["sweeps"] * (2 * L - 3)


def _generic_oracle_vs_cxx(ops, mpo_sites, nsites, target, chi, nsweeps, cap, seed=11, cutoff=0.0):
    bonds, tens = mps.random_mps(nsites, target, cap, seed)
    psi = dmrg_su2.MPS(nsites, target)
    psi.bonds, psi.tensors = [dict(b) for b in bonds], [dict(x) for x in tens]
    omp = [{"left": W.left, "right": W.right, "entries": W.entries} for W in mpo_sites]
    ref = dmrg_su2.DMRG2(psi, omp, chi_full=chi, cutoff=cutoff)
    eng = engine.DMRG2(ops, mpo_sites, bonds, tens, chi_full=chi, cutoff=cutoff)
    for _ in range(nsweeps):
        Er, spec = ref.sweep()
        Eg = eng.sweep()
        assert abs(Eg - Er) <= 1e-8 * max(abs(Er), 1.0)
    got = eng.spectra
    for b in spec:
        for c, v in spec[b].items():
            assert np.abs(np.asarray(v) - got[b][c]).max() <= 1e-8 * max(v)
    return eng


def test_cxx_engine_other_models_and_fillings_match_oracle(cpu_ops):
    """two-band MB_Sim, exchange terms (spin-1 and pair levels), the polyacetylene parameter set, fillings 1/2 and
    3/2, an odd-particle target, and the reference's own truncation scheme truncbelow (src:1007-1010)"""
    t = np.array([[0.0, 1.2, -0.3, 0.0], [1.2, 0.2, 0.9, -0.2]])
    u = np.array([[6.0, 2.0, 0.5, 0.0], [2.0, 5.0, 0.7, 0.1]])
    _generic_oracle_vs_cxx(cpu_ops, models.hamiltonian(models.MB_Sim(t, u, np.zeros((2, 4))), 4), 8, (8, 0), 40, 2, 5)
    _generic_oracle_vs_cxx(cpu_ops, models.hamiltonian(models.OB_Sim([1.0, 0.2], [4.0, 0.5], 0.0, [0.3, 0.1], 1, 1), 8), 8, (8, 0), 40, 2, 5)
    _generic_oracle_vs_cxx(cpu_ops, models.hamiltonian(_poly(), 4), 8, (8, 0), 40, 2, 5)
    H = models.hamiltonian(models.OB_Sim([1.0], [5.0]), 8)
    _generic_oracle_vs_cxx(cpu_ops, H, 8, (4, 0), 40, 2, 5)
    _generic_oracle_vs_cxx(cpu_ops, H, 8, (12, 0), 40, 2, 5)
    _generic_oracle_vs_cxx(cpu_ops, H, 8, (7, 1), 40, 2, 5)
    _generic_oracle_vs_cxx(cpu_ops, H, 8, (8, 0), None, 2, 5, cutoff=1e-2)
    # three-equal-index terms (U13, src:452-458): restricted ladder operators on hop-type MPO levels
    _generic_oracle_vs_cxx(cpu_ops, models.hamiltonian(models.OB_Sim([1.0, 0.2], [4.0], 0.0, 1, 1, U13=[0.3, 0.1]), 8), 8, (8, 0), 40, 2, 5)


def test_cxx_engine_untruncated_equals_exact_diagonalisation_and_heff_is_hermitian(cpu_ops):
    L = 8
    bonds, tens = mps.random_mps(L, (L, 0), 8, 1234)
    eng = engine.DMRG2(cpu_ops, models.hamiltonian(models.OB_Sim([1.0], [4.0]), L), bonds, tens, chi_full=None)
    for _ in range(2):
        E = eng.sweep()
    assert abs(E - E_ED_L8_U4) < 1e-10
    assert eng.bond_dims()[4] == 256
    # H_eff of the centre bond: Hermitian in the plain (tilde) metric, and theta is its eigenvector
    for i in range(3):
        eng.update_bond(i, +1, "right")
    n = len(eng.theta(3))
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    z = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    Hx, Hz = eng.apply_heff(3, x), eng.apply_heff(3, z)
    assert abs(np.vdot(z, Hx) - np.vdot(Hz, x)) < 1e-11 * abs(np.vdot(z, Hx))
    th = eng.theta(3)
    th /= np.linalg.norm(th)
    assert np.linalg.norm(eng.apply_heff(3, th) - E_ED_L8_U4 * th) < 1e-8


def test_expectation_value_is_the_energy_of_the_truncated_state(cpu_ops):
    """api.expectation_value (examples/One_band.jl:42-43): per-site energies of the state AS STORED.  Untruncated: their
    sum is the ED energy.  Truncated: the sum is <psi|H|psi> (variational: above ED and above the pre-truncation Ritz
    value of the last eigensolve, by about the discarded weight times the bandwidth), not the Ritz value."""
    sim = api.OB_Sim([1.0], [4.0], 0.0, 1, 1, 2.0, 8, L=8)
    H = api.hamiltonian(sim)
    psi = api.initialize_mps(H, 1, 8, ops=cpu_ops)
    psi, envs, delta = api.find_groundstate(psi, H, api.DMRG2(trscheme=None, tol=1e-11, maxiter=6, eigsolve_tol=1e-13))
    e = api.expectation_value(psi, H)
    assert e.shape == (8,) and abs(e.sum() - E_ED_L8_U4) < 1e-9
    n_docc = api.double_occupancy(psi)
    # on-site part of every e_i is U <n_up n_dn>_i; the rest is the hopping on bond (i-1, i): negative
    assert np.all(e[1:] - 4.0 * n_docc[1:] < 0.0)
    assert abs(e[0] - 4.0 * n_docc[0]) < 1e-9
    # truncated
    psi2 = api.initialize_mps(H, 1, 8, ops=cpu_ops)
    psi2, _, _ = api.find_groundstate(psi2, H, api.DMRG2(trscheme=api.truncdim(24), tol=1e-10, maxiter=8, eigsolve_tol=1e-12))
    # pre-truncation Ritz values of the last sweep: the centre bonds' are biased low by the weight they then discard
    ritz = min(s.energy for s in psi2.engine.stats[-13:])
    e2 = api.expectation_value(psi2, H)
    tot, per_bond = psi2.engine.bond_energies()
    assert abs(e2.sum() - tot) < 1e-10 and np.ptp(per_bond) < 1e-10      # the same number at every bond
    assert tot > E_ED_L8_U4 + 1e-7 and tot > ritz + 1e-9                    # variational, and not the Ritz value
    assert tot - ritz < 1e-2
    assert len(psi2.engine.stats) == 8 * 13 or len(psi2.engine.stats) % 13 == 0    # bookkeeping passes are not recorded


def test_density_state_and_truncstate_on_cpu(cpu_ops):
    sim = api.OB_Sim([1.0], [5.0], 0.0, 1, 2, 2.0, 8, L=8)      # quarter filling: N = 4
    H = api.hamiltonian(sim)
    psi = api.initialize_mps(H, 1, 8, Q=2, ops=cpu_ops)
    psi, _, _ = api.find_groundstate(psi, H, api.DMRG2(trscheme=None, tol=1e-11, maxiter=6, eigsolve_tol=1e-13))
    n = api.density_state(psi)
    assert abs(n.sum() - 4.0) < 1e-9 and np.abs(n - n[::-1]).max() < 1e-7      # particle number, mirror symmetry
    e = ed.SectorED(8, 2, 2, [1.0], [5.0])
    E0, _ = e.ground_state()
    assert abs(psi.engine.energy - E0) < 1e-9
    nstats = len(psi.engine.stats)
    api.density_state(psi)
    assert len(psi.engine.stats) == nstats                    # observables leave the sweep statistics alone
    E_cut = psi.engine.svd_cut(12)
    assert max(psi.engine.bond_dims()) <= 12 and E_cut > E0 + 1e-8


def test_on_disk_formats_round_trip(cpu_ops, tmp_path, monkeypatch):
    """produce_groundstate caches under the reference's directory / prefix scheme (src:1134-1166) and reloads without
    recomputing; save_state / load_state (src:1669-1691) round-trip the per-site dictionaries; init_state warm start"""
    monkeypatch.setenv("HTN_PROJECT_DIR", str(tmp_path))
    monkeypatch.setattr(api, "_OPS", cpu_ops)
    sim = api.OB_Sim([1.0], [4.0], 0.0, 1, 1, 2.0, 6, L=8)
    sub, stem = storage.cache_name(sim)
    assert sub == "OB" and stem.startswith("groundstate_nospin_t[1.0]_u[4.0]_J[0.0]_U13[0.0]_JMs0.0_0.0_P=1_Q=1_bond_dim=6")
    d1 = api.produce_groundstate(sim, tol=1e-9, maxiter=4)
    E1 = api.expectation_value(d1["groundstate"], d1["ham"]).sum()
    assert os.path.isdir(os.path.join(str(tmp_path), "data", "sims", "OB", stem))
    calls = []
    monkeypatch.setattr(api, "compute_groundstate", lambda *a, **k: calls.append(1))
    d2 = api.produce_groundstate(sim, tol=1e-9, maxiter=4)
    assert not calls                                          # loaded, not recomputed
    E2 = api.expectation_value(d2["groundstate"], d2["ham"]).sum()
    assert abs(E1 - E2) < 1e-12 and api.dim_state(d1["groundstate"]) == api.dim_state(d2["groundstate"])
    monkeypatch.undo()
    # per-site state files
    p = storage.save_state(d1["groundstate"], str(tmp_path), "psi")
    assert sorted(os.listdir(p))[:2] == ["bonds.json", "state1.npz"]
    bonds, sites = storage.load_state(p)
    assert len(sites) == 8 and sites[1]["kind"] == "R"
    eng = engine.DMRG2(cpu_ops, d1["ham"], bonds, [s["blocks"] for s in sites], chi_full=None)
    assert abs(eng.bond_energies()[0] - E1) < 1e-11           # the reloaded state IS the state
    # warm start (init_state, src:1003-1004): the stored state was cut at truncbelow(1e-2); an untruncated sweep from it
    # can only lower the energy, and lands on the exact ground state
    E3 = eng.sweep()
    assert E3 < E1 + 1e-9 and abs(E3 - E_ED_L8_U4) < 1e-6


def test_whole_sweep_through_the_c_abi_only():
    """a caller that has nothing but the C entry points of include/hubbardtn_hip.h (ctypes here, ccall
    in INTEGRATION.md) drives an L=8 sweep -- ctx, MPO tables, TensorKit-shaped MPS data, htn_dmrg2_sweep, spectra --
    and meets the oracle's golden energies.  No hubbardtn_amd.engine, no planner on this side."""
    from oracle.cpu_backend import build as cpu_build
    lib = C.CDLL(cpu_build.build_library(verbose=False))
    abi.declare_engine(lib)
    rec = GOLD["oracle_runs"]["L8_U4_chi64"]
    L = rec["L"]
    ctx = C.c_void_p()
    assert lib.htn_ctx_create(abi.BACKEND_CPU, 0, None, C.byref(ctx)) == 0
    assert lib.htn_ctx_create(abi.BACKEND_HIP, 0, None, C.byref(C.c_void_p())) != 0      # no cross-backend fallback
    assert b"CPU baseline" in lib.htn_last_error()
    # --- MPO tables of H = U sum docc - t sum (c+c + h.c.), written out by hand (SURVEY App. A.3 Jordan form) ---
    SQ2 = np.sqrt(2.0)
    ops = np.zeros(6, dtype=abi.SITE_OP_DT)                   # id, F, docc, cdagF, c, Fc, cdag  (reduced elements, src:281-293)

    def red(entries):
        m = np.zeros((4, 4))
        for (o, i), v in entries.items():
            m[o, i] = v
        return m.reshape(-1)
    table = [("id", 0, 0, {(0, 0): 1, (1, 1): 1, (2, 2): 1}), ("F", 0, 0, {(0, 0): 1, (1, 1): -1, (2, 2): 1}),
             ("docc", 0, 0, {(2, 2): 1}), ("cdagF", 1, 1, {(1, 0): 1, (2, 1): SQ2}), ("c", 1, -1, {(0, 1): SQ2, (1, 2): 1}),
             ("Fc", 1, -1, {(0, 1): SQ2, (1, 2): -1}), ("cdag", 1, 1, {(1, 0): 1, (2, 1): -SQ2})]
    ops = np.zeros(len(table), dtype=abi.SITE_OP_DT)
    oid = {}
    for k, (name, kk, dN, ent) in enumerate(table):
        ops[k]["k"], ops[k]["dN"], ops[k]["red"] = kk, dN, red(ent)
        oid[name] = k
    U, t = rec["u"][0], rec["t"][0]
    bulk = [(0, 0), (1, 1), (-1, 1), (0, 0)]                  # start, hop+ (c+ emitted), hop- (c emitted), final
    level_ptr, levels, entry_ptr, entries = [0], [], [0], []
    for b in range(L + 1):
        lv = [(0, 0)] if b in (0, L) else bulk
        levels += lv
        level_ptr.append(len(levels))
    for i in range(L):
        first, last = i == 0, i == L - 1
        fin_r = 0 if last else 3
        if not last:
            entries.append((0, 0, oid["id"], 1.0))            # start -> start
        if not first:
            entries.append((3, fin_r, oid["id"], 1.0))        # final -> final
        entries.append((0, fin_r, oid["docc"], U))
        if not last:
            entries += [(0, 1, oid["cdagF"], 1.0), (0, 2, oid["Fc"], 1.0)]
        if not first:
            entries += [(1, fin_r, oid["c"], -t * SQ2), (2, fin_r, oid["cdag"], t * SQ2)]
        entry_ptr.append(len(entries))
    ent = np.zeros(len(entries), dtype=abi.MPO_ENTRY_DT)
    for q, (wl, wr, op, cf) in enumerate(entries):
        ent[q] = (wl, wr, op, 0, cf, 0.0)
    sym = abi.Symmetry()
    sym.kind, sym.n_site = abi.SYM_SU2_U1, 3
    for s, (N, j) in enumerate(((0, 0), (1, 1), (2, 0))):
        sym.site_N[s], sym.site_j[s] = N, j
    lp, lv = np.array(level_ptr, dtype=np.int32), np.array(levels, dtype=np.int32)
    ep = np.array(entry_ptr, dtype=np.int32)
    mpo = C.c_void_p()
    assert lib.htn_mpo_create(ctx, C.byref(sym), L, ops.ctypes.data, len(table), lp.ctypes.data, lv.ctypes.data, ep.ctypes.data,
                              ent.ctypes.data, C.byref(mpo)) == 0, lib.htn_last_error()
    # --- MPS: flat vector per site + sub-block table (TensorKit-shaped) ---
    bonds, tens = mps.random_mps(L, (L, 0), rec["cap"], rec["seed"])
    bond_ptr, secs, sub_ptr, subs, data_ptr, chunks = [0], [], [0], [], [0], []
    for b in bonds:
        secs += [(N, j, n) for (N, j), n in sorted(b.items())]
        bond_ptr.append(len(secs))
    for i in range(L):
        off = 0
        for (l, s, r), blk in tens[i].items():
            subs.append((l[0], l[1], s, r[0], r[1], blk.shape[0], off))
            chunks.append(np.asfortranarray(blk).reshape(-1, order="F"))
            off += blk.size
        sub_ptr.append(len(subs))
        data_ptr.append(data_ptr[-1] + off)
    sec = np.array(secs, dtype=np.int32)
    sb = np.zeros(len(subs), dtype=abi.SUBBLOCK_DT)
    for q, rec_ in enumerate(subs):
        sb[q] = rec_
    data = np.concatenate(chunks)
    bp, sp, dp = np.array(bond_ptr, dtype=np.int32), np.array(sub_ptr, dtype=np.int32), np.array(data_ptr, dtype=np.int64)
    psi = C.c_void_p()
    assert lib.htn_mps_create(ctx, mpo, L, bp.ctypes.data, sec.ctypes.data, sp.ctypes.data, sb.ctypes.data, dp.ctypes.data,
                              data.ctypes.data, None, None, C.byref(psi)) == 0, lib.htn_last_error()
    o = abi.SweepOpts()
    o.chi_full, o.krylovdim, o.maxrestart, o.lanczos_tol = rec["chi"], 30, 3, 1e-12
    stats = np.zeros(2 * L - 3, dtype=abi.BOND_STATS_DT)
    for k in range(rec["sweeps"]):
        E = C.c_double()
        assert lib.htn_dmrg2_sweep(psi, C.byref(o), stats.ctypes.data, C.byref(E)) == 0, lib.htn_last_error()
        assert abs(E.value - rec["energies"][k]) <= 1e-9 * abs(E.value)
    assert stats["bond"].tolist() == list(range(1, L)) + list(range(L - 2, 0, -1))
    n = lib.htn_mps_spectrum(psi, 4, None, None)
    secs_o, vals = np.zeros(n, dtype=abi.SECTOR_DT), np.zeros(n)
    lib.htn_mps_spectrum(psi, 4, secs_o.ctypes.data, vals.ctypes.data)
    pos = 0
    gold = rec["spectra_last_sweep"]["4"]
    for r in secs_o:
        if pos >= n:
            break
        v = np.asarray(gold[f"{int(r['N'])},{int(r['j'])}"])
        assert np.abs(vals[pos:pos + int(r["count"])] - v).max() < 1e-9
        pos += int(r["count"])
    assert pos == n
    # error behaviour: status code + thread-local message, no exception
    assert lib.htn_bond_update(psi, L, +1, 0, 1, C.byref(o), None) != 0 and b"out of range" in lib.htn_last_error()
    lib.htn_mps_destroy(psi)
    lib.htn_mpo_destroy(mpo)
    lib.htn_ctx_destroy(ctx)


def test_exception_in_the_exchange_hook_is_reraised_by_the_engine_call():
    """the reduction hook runs inside the library (one call per matvec): what it raises cannot cross the C frame, so the
    context provider stores it, the solve is aborted, and the engine call re-raises the ORIGINAL exception -- not a generic
    HtnError -- once the library call has returned; the next call works again"""
    from cpu_ops import CpuOps
    L = 6
    H = models.hamiltonian(models.OB_Sim([1.0], [4.0]), L)
    bonds, tens = mps.random_mps(L, (L, 0), 4, seed=2)
    ops = CpuOps()
    calls = {"n": 0, "fail": True}

    class Boom(RuntimeError):
        pass

    def hook(y):
        calls["n"] += 1
        if calls["fail"] and calls["n"] == 3:
            raise Boom("reduction failed on purpose")
    ops.set_exchange(0, 1, hook)
    eng = engine.DMRG2(ops, H, bonds, tens, chi_full=32)
    with pytest.raises(Boom, match="on purpose"):
        eng.sweep()
    calls["fail"] = False
    bonds, tens = mps.random_mps(L, (L, 0), 4, seed=2)
    eng = engine.DMRG2(ops, H, bonds, tens, chi_full=32)
    E = eng.sweep()
    assert np.isfinite(E) and calls["n"] > 3


def test_mps_import_export_tables_shuffled_order_and_padded_leading_dimensions():
    """Python twin of the Julia side's export_mps / import_mps (integration/HubbardHIP.jl): `htn_mps_create` must accept the
    sub-block table in ANY order, with leading dimensions larger than the block height and gaps between the blocks of the
    flat vector (TensorKit's fusion-tree views are strided windows of one data vector), plus sub-blocks between sectors the
    bond tables do not hold (skipped), and `htn_mps_get_site` must hand every block back bit for bit -- the state is then
    exactly the one the tidy tables describe: same energies from the same sweeps."""
    lib = CpuOps().lib
    L = 8
    Hm = models.hamiltonian(models.OB_Sim([1.0], [4.0]), L)
    bonds, tens = mps.random_mps(L, (L, 0), 5, seed=77)
    ops = CpuOps()
    cm = engine.CMpo(ops, Hm)
    rng = np.random.default_rng(5)

    def tables(shuffle):
        bond_ptr, secs, sub_ptr, subs, data_ptr, chunks = [0], [], [0], [], [0], []
        for b in bonds:
            secs += [(N, j, n) for (N, j), n in sorted(b.items())]
            bond_ptr.append(len(secs))
        for i in range(L):
            items = list(tens[i].items())
            if shuffle:
                items = [items[k] for k in rng.permutation(len(items))]
                # a sub-block whose right sector is not in the bond table: the import must skip it
                items.insert(len(items) // 2, (((0, 0), 0, (99, 0)), np.full((2, 2), 7.0 + 0j)))
            off, flat = 0, []
            for (l, s, r), blk in items:
                m, n = blk.shape
                ld = m + (int(rng.integers(1, 4)) if shuffle else 0)          # padded leading dimension
                gap = int(rng.integers(0, 5)) if shuffle else 0               # unused elements in front of the block
                buf = np.full(gap + ld * n, np.nan + 0j)                      # padding is NaN: it must never be read
                for c in range(n):
                    buf[gap + c * ld:gap + c * ld + m] = blk[:, c]
                subs.append((l[0], l[1], s, r[0], r[1], ld, off + gap))
                flat.append(buf)
                off += buf.size
            chunks.append(np.concatenate(flat))
            sub_ptr.append(len(subs))
            data_ptr.append(data_ptr[-1] + off)
        sb = np.zeros(len(subs), dtype=abi.SUBBLOCK_DT)
        for q, r_ in enumerate(subs):
            sb[q] = r_
        return (np.array(bond_ptr, dtype=np.int32), np.array(secs, dtype=np.int32), np.array(sub_ptr, dtype=np.int32), sb,
                np.array(data_ptr, dtype=np.int64), np.concatenate(chunks))

    def create(t):
        bp, sec, sp, sb, dp, data = t
        h = C.c_void_p()
        assert lib.htn_mps_create(ops.ctx, cm.handle, L, bp.ctypes.data, sec.ctypes.data, sp.ctypes.data, sb.ctypes.data, dp.ctypes.data,
                                  data.ctypes.data, None, None, C.byref(h)) == 0, lib.htn_last_error()
        return h

    def export(h):
        out = []
        for i in range(L):
            nb = lib.htn_mps_get_site(h, i, None, None)
            size = lib.htn_mps_site_size(h, i, None)
            subs = np.zeros(nb, dtype=abi.SUBBLOCK_DT)
            flat = np.zeros(max(size, 1), dtype=np.complex128)
            assert lib.htn_mps_get_site(h, i, subs.ctypes.data, flat.ctypes.data) == nb
            site = {}
            for sb in subs:
                l, r = (int(sb["lN"]), int(sb["lj"])), (int(sb["rN"]), int(sb["rj"]))
                m, n = bonds[i][l], bonds[i + 1][r]
                idx = int(sb["off"]) + np.arange(m)[:, None] + int(sb["ld"]) * np.arange(n)[None, :]
                site[(l, int(sb["s"]), r)] = flat[idx]
            out.append(site)
        return out

    h_tidy, h_mess = create(tables(False)), create(tables(True))
    for i, (a, b) in enumerate(zip(export(h_tidy), export(h_mess))):
        assert a.keys() == b.keys() and set(a) >= set(tens[i])
        for key in a:
            assert np.array_equal(a[key], b[key]), (i, key)
            if key in tens[i]:
                assert np.array_equal(a[key], np.asarray(tens[i][key], dtype=np.complex128)), (i, key)     # bit for bit
            else:
                assert not a[key].any()                                   # structurally allowed block the caller did not send
    o = abi.SweepOpts()
    o.chi_full, o.krylovdim, o.maxrestart, o.lanczos_tol = 48, 30, 3, 1e-12
    for _ in range(2):
        E1, E2 = C.c_double(), C.c_double()
        assert lib.htn_dmrg2_sweep(h_tidy, C.byref(o), None, C.byref(E1)) == 0 and lib.htn_dmrg2_sweep(h_mess, C.byref(o), None, C.byref(E2)) == 0
        assert E1.value == E2.value and np.isfinite(E1.value)
    lib.htn_mps_destroy(h_tidy)
    lib.htn_mps_destroy(h_mess)
