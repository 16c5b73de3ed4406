"""BASELINE.json's configurations at their REAL sizes under assertion:

  * oracle trajectories generated in the build container (tests/golden/golden_r02.json, make_golden_r02.py): L=64 U=4
    chi 32 -> 128, the polyacetylene parameter set chi 32 -> 128, and configs[1] itself, L=64 U=4 chi 32 -> 512
    -- the HIP engine replays the same schedule from the same seeded start: energies after every sweep and truncated
    Schmidt spectra at 1e-8;
  * centre-bond updates of the grown chi=512 states (one band: configs[1]; polyacetylene: configs[3]) checked against the
    numpy oracle run on the SAME tensors and environments (downloaded): the default large-block SVD route, the real
    sector tables and the K-slab pre-split GEMM tiles at their production shapes;
  * size-independent properties at chi=1024 (the bench line's configuration): H_eff Hermitian, written-back isometries
    orthonormal, energy non-increasing over sweeps, sum of (2S+1) schmidt^2 = 1;
  * configs[2]'s code path: the sector-parallel apply (zero-filled y, rank-dealt tiles, RCCL all-reduce inside the
    library) at world size 1 on the one GPU, bit-equal to the unsharded run.
"""
import json
import os

import numpy as np
import pytest

from hubbardtn_amd import engine, models, mps
from oracle import dmrg_su2, mpo as ompo

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_r02.json")))
TRACE = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_r03.json")))       # per-bond, first sweep
POLY = dict(t=np.array([[0.000, 3.803, -0.548, 0.000], [3.803, 0.000, 2.977, -0.501]]),
            u=np.array([[10.317, 6.264, 0.000, 0.000], [6.264, 10.317, 6.162, 0.000]]),
            J=np.array([[0.000, 0.123, 0.000, 0.000], [0.123, 0.000, 0.113, 0.000]]))


def _mpo(rec):
    if rec["model"] == "one_band":
        return models.hamiltonian(models.OB_Sim(rec["t"], rec["u"]), rec["L"])
    return models.hamiltonian(models.MB_Sim(POLY["t"], POLY["u"], POLY["J"], 1, 1, 2.5, 20), rec["L"] // 2)


def first_departure(stats, trace, krylovdim):
    """first bond update of a sweep whose statistics leave the oracle's per-bond trace (golden_r03.json), as
    (position in the sweep, bond, direction, stage, got, want) -- or None.  Stages, in the order they run inside one update:
    'lanczos' (eigenvalue / matvec count), 'svd+truncation' (discarded weight, kept multiplets, TensorKit dim).  The
    product counts the speculatively enqueued step of every restart cycle as a matvec (htn_krylov.hip), the oracle does
    not: 0 <= difference <= number of cycles; one more step either way is allowed for a residual that meets the tolerance
    within rounding (seen between the oracle and the CPU backend: 92 vs 91 matvecs on bond 14 of the polyacetylene run)."""
    assert len(stats) == len(trace), (len(stats), len(trace))
    for pos, (s, t) in enumerate(zip(stats, trace)):
        assert (s.bond, s.direction) == (t["bond"], t["dir"]), (pos, s.bond, s.direction, t)
        cycles = -(-t["nmv"] // krylovdim)
        if abs(s.energy - t["E"]) > 1e-8 * abs(t["E"]) or not -1 <= s.n_matvec - t["nmv"] <= cycles + 1:
            return pos, t["bond"], t["dir"], "lanczos", (s.energy, s.n_matvec), (t["E"], t["nmv"])
        if (s.multiplets, s.chi_full) != (t["mult"], t["chi_full"]) or abs(s.trunc_weight - t["trunc"]) > 1e-8 * max(t["trunc"], 1e-6):
            return (pos, t["bond"], t["dir"], "svd+truncation", (s.multiplets, s.chi_full, s.trunc_weight),
                    (t["mult"], t["chi_full"], t["trunc"]))
    return None


@pytest.mark.parametrize("name", ["L64_U4_chi128", "poly32_chi128", "L64_U4_chi512"])
def test_hip_engine_replays_the_oracle_trajectory(hip_ops, name):
    rec = GOLD[name]
    L = rec["L"]
    bonds, tens = mps.random_mps(L, (L, 0), rec["cap"], rec["seed"])
    eng = engine.DMRG2(hip_ops, _mpo(rec), bonds, tens, chi_full=rec["schedule"][0][0], lanczos_tol=rec["lanczos_tol"],
                       krylovdim=rec["krylovdim"], maxrestart=rec["maxrestart"])
    trace = next(t for t in TRACE.values() if name in t["same_start_as"])
    k = 0
    for chi, nsw in rec["schedule"]:
        eng.chi_full = chi
        for _ in range(nsw):
            n0 = len(eng.stats)
            E = eng.sweep()
            if k == 0:          # bond by bond against the oracle's trace: a red run names the first (bond, stage) that departs
                dep = first_departure(eng.stats[n0:], trace["bonds"], rec["krylovdim"])
                assert dep is None, (name, "first departure (position, bond, direction, stage, got, want)", dep)
            assert abs(E - rec["energies"][k]) <= 1e-8 * abs(rec["energies"][k]), (name, k, E, rec["energies"][k])
            k += 1
    assert eng.bond_dims() == rec["bond_dims"]
    for b, s in rec["spectra_last_sweep"].items():
        got = eng.spectrum(int(b))
        assert set(got) == {tuple(int(x) for x in c.split(",")) for c in s}
        for c, v in s.items():
            v = np.asarray(v)
            g = got[tuple(int(x) for x in c.split(","))]
            assert g.shape == v.shape and np.abs(g - v).max() <= 1e-8 * v.max(), (name, b, c)


def _grow(ops, mpo, L, schedule, tol=1e-6):
    bonds, tens = mps.random_mps(L, (L, 0), 4, seed=1234)
    eng = engine.DMRG2(ops, mpo, bonds, tens, chi_full=16, lanczos_tol=tol)
    for chi, nsw in schedule:
        eng.chi_full = chi
        for _ in range(nsw):
            eng.sweep()
    return eng


def _oracle_on_bond(eng, mpo_sites, i0, placement):
    """one oracle bond update on the tensors / environments downloaded from the engine (centre on site i0)"""
    L = eng.L
    o = object.__new__(dmrg_su2.DMRG2)
    psi = dmrg_su2.MPS(L, (L, 0))
    psi.bonds = [dict(b.dims) for b in eng.bonds]
    psi.tensors = [None] * L
    o.psi, o.L = psi, L
    o.mpo = [{"left": list(W.left), "right": list(W.right), "entries": list(W.entries)} for W in mpo_sites]
    o.chi_full, o.cutoff, o.weighting = eng.chi_full, eng.cutoff, eng.weighting
    o.krylovdim, o.lanczos_tol, o.maxrestart = eng.krylovdim, eng.lanczos_tol, eng.maxrestart
    o.Lenvs, o.Renvs = [None] * (L + 1), [None] * (L + 1)
    o.stats, o.energy = [], None
    for s in (i0, i0 + 1):
        psi.tensors[s] = eng.download_site(s)
    o.Lenvs[i0] = eng.download_env("L", i0)
    o.Renvs[i0 + 2] = {(bra, w, ket): m.T.copy() for (ket, w, bra), m in eng.download_env("R", i0 + 2).items()}
    return o.update_bond(i0, +1, placement)


@pytest.mark.parametrize("model", ["one_band", "polyacetylene"])
def test_centre_bond_at_chi512_matches_the_oracle_on_the_same_tensors(hip_ops, model):
    """configs[1] (one band L=64 chi=512) and configs[3] (polyacetylene, 64 chain sites, chi=512) at production shape"""
    from threadpoolctl import threadpool_limits
    L = 64
    if model == "one_band":
        H = models.hamiltonian(models.OB_Sim([1.0], [4.0]), L)
    else:
        H = models.hamiltonian(models.MB_Sim(POLY["t"], POLY["u"], POLY["J"], 1, 1, 2.5, 20), L // 2)
    eng = _grow(hip_ops, H, L, [(16, 4), (32, 3), (64, 2), (128, 2), (256, 1), (512, 1)])
    eng.lanczos_tol = 1e-12
    i0 = L // 2 - 1
    for i in range(0, i0):                                     # centre (on site 0 after a sweep) -> site i0
        eng.update_bond(i, +1, "right")
    assert max(eng.bond_dims()) == 512
    with threadpool_limits(limits=1):
        Er, spec = _oracle_on_bond(eng, H, i0, "left")
    Eg = eng.update_bond(i0, +1, "left")                       # 'left' placement: the centre stays on i0
    st = eng.stats[-1]
    assert st.n_tiles > 200 and st.jacobi_sweeps >= 1
    assert abs(Eg - Er) <= 1e-8 * abs(Er), (Eg, Er)
    got = eng.spectrum(i0 + 1)
    assert set(got) == set(spec)
    for c, v in spec.items():
        assert got[c].shape == np.asarray(v).shape and np.abs(got[c] - np.asarray(v)).max() <= 1e-8 * max(v), c
    # the other sweep direction from the same point (G0 staged as M^H instead of M)
    with threadpool_limits(limits=1):
        Er2, spec2 = _oracle_on_bond(eng, H, i0, "right")
    Eg2 = eng.update_bond(i0, +1, "right")
    assert abs(Eg2 - Er2) <= 1e-8 * abs(Er2)
    got = eng.spectrum(i0 + 1)
    for c, v in spec2.items():
        assert np.abs(got[c] - np.asarray(v)).max() <= 1e-8 * max(v), c


def test_properties_at_the_bench_configuration_chi1024(hip_ops):
    L = 64
    H = models.hamiltonian(models.OB_Sim([1.0], [4.0]), L)
    eng = _grow(hip_ops, H, L, [(16, 8), (32, 4), (64, 4), (128, 2), (256, 2), (512, 2)])
    eng.chi_full, eng.lanczos_tol = 1024, 1e-10
    Es = [eng.sweep() for _ in range(3)]
    assert max(eng.bond_dims()) == 1024
    assert Es[1] <= Es[0] + 1e-9 * abs(Es[0]) and Es[2] <= Es[1] + 1e-9 * abs(Es[1])        # monotone over sweeps
    assert abs(Es[2] / L - (-0.5680)) < 1e-3                                                   # e_inf + boundary term
    # normalisation of every bond's Schmidt spectrum: sum (2S+1) s^2 = 1
    for b, s in eng.spectra.items():
        tot = sum((c[1] + 1) * float(np.sum(v ** 2)) for c, v in s.items())
        assert abs(tot - 1.0) < 1e-12, b
    i0 = L // 2 - 1
    for i in range(0, i0):
        eng.update_bond(i, +1, "right")
    # H_eff of the centre bond is Hermitian: <z|Hx> = <Hz|x> (plain metric of the tilde normalisation)
    n = len(eng.theta(i0))
    assert n > 100_000
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    z = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    Hx, Hz = eng.apply_heff(i0, x), eng.apply_heff(i0, z)
    assert abs(np.vdot(z, Hx) - np.vdot(Hz, x)) <= 1e-11 * abs(np.vdot(z, Hx))
    # written-back isometries: A_i (left layout) has orthonormal columns per right sector, B_j orthonormal rows
    for site, kind in ((i0 - 1, "L"), (i0 + 2, "R")):
        assert eng.site_kind(site) == kind
        acc = {}
        for (l, s, r), blk in eng.download_site(site).items():
            key = r if kind == "L" else l
            g = blk.conj().T @ blk if kind == "L" else blk @ blk.conj().T
            acc[key] = acc.get(key, 0) + g
        for c, g in acc.items():
            assert np.abs(g - np.eye(g.shape[0])).max() < 1e-12, (site, c)


def test_sharded_apply_world1_is_bit_equal_to_the_unsharded_run(hip_ops):
    """configs[2]'s code path on one GPU: a second context with an RCCL communicator of size 1 -- y is zero-filled,
    the tile list goes through the rank split, ncclAllReduce runs on the library's stream after every matvec"""
    import torch.distributed as dist
    from hubbardtn_amd.device import HipOps
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        ops2 = HipOps(0)
        ops2.set_comm(0, 1)
        L = 16
        H = models.hamiltonian(models.OB_Sim([1.0], [4.0]), L)

        def run(ops):
            bonds, tens = mps.random_mps(L, (L, 0), 8, seed=3)
            eng = engine.DMRG2(ops, H, bonds, tens, chi_full=300, lanczos_tol=1e-10)
            Es = [eng.sweep() for _ in range(2)]
            return Es, eng.spectrum(L // 2), sum(s.n_matvec for s in eng.stats)
        E1, S1, mv1 = run(hip_ops)
        E2, S2, mv2 = run(ops2)
        assert E1 == E2 and mv1 == mv2
        for c in S1:
            assert np.array_equal(S1[c], S2[c])
    finally:
        if created:
            dist.destroy_process_group()


def test_config5_L128_nnn_chi2048_centre_bond_against_the_oracle_and_properties(hip_ops):
    """BASELINE configs[4]: L=128, hopping t = [1.0, 0.1] (examples/One_band.jl:25), U/t=4, chi=2048 -- on one GPU (the
    8-GPU sharding of the same sweep is the code path of test_sharded_apply_world1_... and the gloo world-2 test).  The
    grown state's centre bond is updated by the HIP engine and by the numpy oracle on the same downloaded tensors
    (blocks of ~400 rows: the non-LDS SVD route, multi-slab GEMM tiles, range-2 MPO with the Z stage), then the
    size-independent properties."""
    from threadpoolctl import threadpool_limits
    L = 128
    H = models.hamiltonian(models.OB_Sim([1.0, 0.1], [4.0]), L)
    eng = _grow(hip_ops, H, L, [(16, 6), (32, 3), (64, 2), (128, 2), (256, 1), (512, 1), (1024, 1)])
    eng.chi_full, eng.lanczos_tol = 2048, 1e-10
    E1 = eng.sweep()
    E2 = eng.sweep()
    assert max(eng.bond_dims()) == 2048 and E2 <= E1 + 1e-9 * abs(E1)
    assert abs(E2 / L - (-0.5711)) < 1e-3
    for b, s in eng.spectra.items():
        assert abs(sum((c[1] + 1) * float(np.sum(v ** 2)) for c, v in s.items()) - 1.0) < 1e-12, b
    i0 = L // 2 - 1
    for i in range(0, i0):
        eng.update_bond(i, +1, "right")
    eng.lanczos_tol = 1e-12
    with threadpool_limits(limits=1):
        Er, spec = _oracle_on_bond(eng, H, i0, "left")
    Eg = eng.update_bond(i0, +1, "left")
    assert abs(Eg - Er) <= 1e-8 * abs(Er), (Eg, Er)
    got = eng.spectrum(i0 + 1)
    assert set(got) == set(spec)
    for c, v in spec.items():
        assert got[c].shape == np.asarray(v).shape and np.abs(got[c] - np.asarray(v)).max() <= 1e-8 * max(v), c
    n = len(eng.theta(i0))
    rng = np.random.default_rng(1)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    z = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    assert abs(np.vdot(z, eng.apply_heff(i0, x)) - np.vdot(eng.apply_heff(i0, z), x)) <= 1e-11 * abs(np.vdot(z, eng.apply_heff(i0, x)))
