"""CPU suite for the host logic: model builder, planner task lists and the sweep driver, executed on the
numpy interpreter of the C-ABI primitives (tests/emul.py) and compared with the oracle."""
import json
import os

import numpy as np
import pytest

from emul import NumpyOps
import ref_engine as engine
import ref_planner as pl
from hubbardtn_amd import api, models, mps
from oracle import dmrg_su2, ed, mpo as ompo

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_r01.json")))


def _as_dict(mpo):
    return [{"left": W.left, "right": W.right, "entries": W.entries} for W in mpo]


def test_ob_sim_constructors_follow_reference():
    """two positional forms distinguished by a J vector (src/HubbardFunctions.jl:87-92)"""
    a = models.OB_Sim([1.0, 0.1], [8.0], 0.0, 1, 1, 2.5, 20, spin=False)
    assert (a.P, a.Q, a.svalue, a.bond_dim, a.period, a.J) == (1, 1, 2.5, 20, 0, [0.0])
    b = models.OB_Sim([1.0], [4.0], 0.5, [0.0], 1, 2, 3.0)
    assert (b.mu, b.P, b.Q, b.svalue, b.bond_dim) == (0.5, 1, 2, 3.0, 50)
    c = models.OB_Sim([1.0], [4.0])
    assert (c.mu, c.P, c.Q, c.svalue, c.bond_dim) == (0.0, 1, 1, 2.0, 50)


@pytest.mark.parametrize("t,u,mu,L", [([1.0], [4.0], 0.0, 4), ([1.0, 0.1], [8.0, 0.5, 0.25], 0.3, 5)])
def test_one_band_mpo_equals_dense_hamiltonian(t, u, mu, L):
    H = models.hamiltonian(models.OB_Sim(t, u, mu), L)
    assert np.abs(ompo.mpo_to_dense(_as_dict(H)) - ed.dense_hamiltonian(L, t, u, mu)).max() < 1e-13


def test_multi_band_mpo_equals_dense_hamiltonian():
    """2 bands snaked onto a chain (InfiniteStrip order, src:491): on-site + inter-cell hopping and
    direct terms (src:498, 515, 561, 664), diagonal of t as chemical potential (src:861-864)"""
    t = np.array([[0.3, 1.1, -0.2, 0.0], [1.1, -0.1, 0.7, -0.3]])
    u = np.array([[5.0, 2.0, 0.4, 0.0], [2.0, 6.0, 0.6, 0.1]])
    sim = models.MB_Sim(t, u, np.zeros((2, 4)))
    cells, B = 2, 2
    n = cells * B
    M = ompo.mpo_to_dense(_as_dict(models.hamiltonian(sim, cells)))
    # dense Jordan-Wigner construction
    from oracle import su2
    lm = su2.local_matrices()

    def site_op(mats):
        out = np.eye(1)
        for s in range(n):
            out = np.kron(out, mats.get(s, lm["id"]))
        return out

    def c_op(i, spin):
        mats = {s: lm["F"] for s in range(i)}
        mats[i] = lm["a_up"] if spin == 0 else lm["a_dn"]
        return site_op(mats)
    c = {(i, s): c_op(i, s) for i in range(n) for s in (0, 1)}
    nn = {i: site_op({i: lm["n"]}) for i in range(n)}
    site = lambda b, cell: b + cell * B
    H = np.zeros((4 ** n, 4 ** n))
    for cell in range(cells):
        for b in range(B):
            i = site(b, cell)
            H += u[b, b] * site_op({i: lm["docc"]}) - t[b, b] * nn[i]
        for bi in range(B):
            for bf in range(B):
                if bi != bf:
                    for s in (0, 1):
                        H += -t[bi, bf] * c[(site(bf, cell), s)].T @ c[(site(bi, cell), s)]      # src:498
        H += 0.5 * (u[0, 1] + u[1, 0]) * nn[site(0, cell)] @ nn[site(1, cell)]                    # src:548-561
    for cell in range(cells - 1):
        for bi in range(B):
            for bf in range(B):
                i, j = site(bi, cell), site(bf, cell + 1)
                for s in (0, 1):
                    h = c[(j, s)].T @ c[(i, s)]
                    H += -t[bi, B + bf] * (h + h.T)                                               # src:515
                H += u[bi, B + bf] * nn[i] @ nn[j]                                                # src:664
    assert np.abs(M - H).max() < 1e-12
    assert np.abs(M - M.T).max() < 1e-13


@pytest.mark.parametrize("name", ["L8_U4_chi64", "L12_t2_chi48"])
def test_sweep_driver_on_emulator_matches_golden(name):
    """planner + engine host logic (layouts, 9j coefficients, task lists, truncation, SVD staging) vs the
    oracle's golden energies / spectra"""
    rec = GOLD["oracle_runs"][name]
    L = rec["L"]
    bonds, tens = mps.random_mps(L, (L, 0), rec["cap"], rec["seed"])
    eng = engine.DMRG2(NumpyOps(), models.hamiltonian(models.OB_Sim(rec["t"], rec["u"]), L), bonds, tens,
                       chi_full=rec["chi"])
    for k in range(rec["sweeps"]):
        E = eng.sweep()
        assert abs(E - rec["energies"][k]) <= 1e-9 * abs(E)
    for b, s in rec["spectra_last_sweep"].items():
        for c, v in s.items():
            key = tuple(int(x) for x in c.split(","))
            assert np.abs(eng.spectra[int(b)][key] - np.asarray(v)).max() < 1e-9
    assert eng.bond_dims() == rec["bond_dims"]
    assert eng.cache_hits > 0


def test_apply_plan_is_hermitian_and_matches_oracle_terms():
    """one bond: y = H_eff x from the planner's task list equals the oracle's term-by-term apply"""
    L, t, u = 8, [1.0], [4.0, 0.5]       # (level order of oracle and product MPOs coincides for range-1 terms)
    bonds, tens = mps.random_mps(L, (L, 0), 5, 3)
    ops = NumpyOps()
    eng = engine.DMRG2(ops, models.hamiltonian(models.OB_Sim(t, u), L), bonds, tens, chi_full=40)
    for i in range(3):
        eng.update_bond(i, +1, "right")
    i = 3
    tl = pl.ThetaLayout.build(eng.bonds[i], eng.bonds[i + 2])
    stages, flops, nbytes, ntiles, nsegs = eng._make_apply(i, tl)
    rng = np.random.default_rng(0)
    x = rng.standard_normal(tl.size) + 1j * rng.standard_normal(tl.size)
    z = rng.standard_normal(tl.size) + 1j * rng.standard_normal(tl.size)

    def apply(v):
        y = np.zeros(tl.size, dtype=np.complex128)
        for bufs, tasks in stages:
            b = list(bufs)
            b[pl.BUF_X], b[pl.BUF_Y] = v, y
            ops.grouped_gemm(b, tasks)
        return y
    Hx, Hz = apply(x), apply(z)
    assert abs(np.vdot(z, Hx) - np.vdot(Hz, x)) < 1e-10 * abs(np.vdot(z, Hx))      # Hermitian in the plain metric
    # oracle on the same environments
    Lenv = eng.download_env("L", i)
    Renv = {(bra, w, ket): m.T.copy() for (ket, w, bra), m in eng.download_env("R", i + 2).items()}
    W = ompo.hubbard_mpo(L, t, u)
    for a, b in zip(W, eng.mpo):
        assert a["left"] == b.left and a["right"] == b.right        # identical level order
    theta = {}
    for key, (off, m, n, ld) in tl.blocks.items():
        idx = off + np.arange(m)[:, None] + ld * np.arange(n)[None, :]
        theta[key] = x[idx]
    blocks = sorted(theta)
    terms = dmrg_su2.build_apply_terms(blocks, Lenv, Renv, W[i], W[i + 1])
    y = dmrg_su2.apply_heff(theta, terms, Lenv, Renv)
    for key, (off, m, n, ld) in tl.blocks.items():
        idx = off + np.arange(m)[:, None] + ld * np.arange(n)[None, :]
        assert np.abs(Hx[idx] - y[key]).max() < 1e-11
    assert flops > 0 and nbytes > 0


def test_planner_truncate_matches_oracle_rule():
    rng = np.random.default_rng(1)
    sv = {(10, j): np.sort(rng.random(6))[::-1] for j in (0, 2, 4)}
    sv[(9, 1)] = np.sort(rng.random(5))[::-1]
    for chi in (5, 17, 40, None):
        k1, w1, n1 = pl.truncate(sv, chi)
        k2, w2 = dmrg_su2.truncate_spectrum(sv, chi)
        assert k1 == k2 and abs(w1 - w2) < 1e-14
    k1, _, _ = pl.truncate(sv, None, cutoff=0.3)
    k2, _ = dmrg_su2.truncate_spectrum(sv, None, cutoff=0.3)
    assert k1 == k2


def test_api_surface_keeps_reference_names():
    for name in ("OB_Sim", "MB_Sim", "produce_groundstate", "compute_groundstate", "find_groundstate",
                 "initialize_mps", "hamiltonian", "dim_state", "expectation_value", "truncdim", "truncbelow", "DMRG2",
                 "IDMRG2", "density_state", "TruncState", "produce_TruncState"):
        assert hasattr(api, name)
    sim = api.OB_Sim([1.0], [4.0], 0.0, 1, 1, 2.0, 6)
    H = api.hamiltonian(sim, L=6)
    from cpu_ops import CpuOps
    psi = api.initialize_mps(H, sim.P, sim.bond_dim, ops=CpuOps())        # CPU baseline backend injected: CPU test
    psi, envs, delta = api.find_groundstate(psi, H, api.DMRG2(trscheme=api.truncdim(64), tol=1e-9, maxiter=6))
    E = float(np.sum(api.expectation_value(psi, H)))
    ref, _ = ed.SectorED(6, 3, 3, [1.0], [4.0]).ground_state()
    assert abs(E - ref) < 1e-8 and delta < 1e-9
    assert api.dim_state(psi)[2] == 64 or api.dim_state(psi)[2] <= 64


def test_exchange_terms_equal_dense_kanamori_form():
    """J terms (src:445-451 one band; src:565-611, 668-696 multi band): MPO == dense
    J sum c+_{is} c+_{js'} c_{is'} c_{js} + J (D+_i D_j + h.c.), incl. merged n n coefficients"""
    from oracle import su2
    L, t, u, J = 4, [1.0, 0.2], [4.0, 0.5], [0.3, 0.1]
    M = ompo.mpo_to_dense(_as_dict(models.hamiltonian(models.OB_Sim(t, u, 0.1, J, 1, 1), L)))
    lm = su2.local_matrices()

    def site_op(mats):
        out = np.eye(1)
        for s in range(L):
            out = np.kron(out, mats.get(s, lm["id"]))
        return out

    def c_op(i, spin):
        mats = {s: lm["F"] for s in range(i)}
        mats[i] = lm["a_up"] if spin == 0 else lm["a_dn"]
        return site_op(mats)
    c = {(i, s): c_op(i, s) for i in range(L) for s in (0, 1)}
    H = ed.dense_hamiltonian(L, t, u, 0.1)
    for r, Jr in enumerate(J, start=1):
        for i in range(L - r):
            j = i + r
            for s in (0, 1):
                for sp in (0, 1):
                    H += Jr * c[(i, s)].T @ c[(j, sp)].T @ c[(i, sp)] @ c[(j, s)]
            pd = c[(i, 0)].T @ c[(i, 1)].T @ c[(j, 1)] @ c[(j, 0)]
            H += Jr * (pd + pd.T)
    assert np.abs(M - H).max() < 1e-12 and np.abs(M - M.T).max() < 1e-13


def test_polyacetylene_parameters_build_and_sweep_on_emulator():
    """examples/polyacetylene.jl:29-33 parameters (2 bands, hopping range 1 cell, U, exchange J): the MPO builds
    and the sweep driver lowers the energy monotonically on a short chain (emulator); E equals the dense ground
    state of the same MPO for 2 cells"""
    t = np.array([[0.000, 3.803, -0.548, 0.000], [3.803, 0.000, 2.977, -0.501]])
    U = np.array([[10.317, 6.264, 0.000, 0.000], [6.264, 10.317, 6.162, 0.000]])
    J = np.array([[0.000, 0.123, 0.000, 0.000], [0.123, 0.000, 0.113, 0.000]])
    sim = models.MB_Sim(t, U, J, 1, 1, 2.5, 20, code="polyacetylene")
    cells = 2
    H = models.hamiltonian(sim, cells)
    dense = ompo.mpo_to_dense(_as_dict(H))
    assert np.abs(dense - dense.T).max() < 1e-12
    # ground state in the N = 4, S = 0 sector by dense diagonalisation restricted through the DMRG itself
    bonds, tens = mps.random_mps(2 * cells, (2 * cells, 0), 6, 5)
    eng = engine.DMRG2(NumpyOps(), H, bonds, tens, chi_full=None)
    Es = [eng.sweep() for _ in range(3)]
    assert Es[1] <= Es[0] + 1e-9 and abs(Es[2] - Es[1]) < 1e-9
    w = np.linalg.eigvalsh(dense)
    assert Es[-1] >= w[0] - 1e-9              # variational
    # the N=4 singlet ground state is one of the eigenvalues of the full matrix
    assert np.abs(w - Es[-1]).min() < 1e-8


def test_symbolic_apply_plan_equals_loop_plan():
    """the dimension-independent plan instantiated on perturbed multiplicities is byte-identical to the plan the
    loop nest builds from scratch (one-band NN, NNN and the two-band model)"""
    rng = np.random.default_rng(3)
    tab = {(2, 0): 2, (2, 2): 1, (3, 1): 5, (3, 3): 3, (4, 0): 7, (4, 2): 9, (4, 4): 2, (5, 1): 11, (5, 3): 6,
           (6, 0): 8, (6, 2): 9, (6, 4): 3, (7, 1): 5, (7, 3): 2, (8, 0): 2}

    def same(a, b):
        if a is None or b is None:
            return a is b
        return (a.ntiles == b.ntiles and a.nsegs == b.nsegs and a.flops == b.flops
                and a.tiles.tobytes() == b.tiles.tobytes() and a.segs.tobytes() == b.segs.tobytes())
    tm = np.array([[0.0, 1.0, 0.1, 0.0], [1.0, 0.0, 0.8, 0.05]])
    um = np.array([[4.0, 1.0, 0.0, 0.0], [1.0, 4.0, 0.5, 0.0]])
    mpos = [models.hamiltonian(models.OB_Sim([1.0], [4.0]), 12), models.hamiltonian(models.OB_Sim([1.0, 0.1], [4.0, 0.5]), 12),
            models.hamiltonian(models.MB_Sim(tm, um, np.zeros_like(um), 1, 1, 2.0, 8), 6)]
    for mpo in mpos:
        for trial in range(3):
            pert = lambda n: max(1, int(n + rng.integers(-2, 3)))
            bl = pl.Bond({(N + 2, j): pert(n) for (N, j), n in tab.items()})
            br = pl.Bond({(N + 4, j): pert(n) for (N, j), n in tab.items()})
            tl = pl.ThetaLayout.build(bl, br)
            Ll = pl.EnvLayout.build("L", bl, mpo[5].left)
            Rl = pl.EnvLayout.build("R", br, mpo[6].right)
            ref = pl.plan_apply(tl, Ll, Rl, mpo[5], mpo[6])
            new = pl.plan_apply_cached(tl, Ll, Rl, mpo[5], mpo[6])
            assert same(ref[0], new[0]) and same(ref[1], new[1]) and ref[2:] == new[2:]
    # every bond of a short chain, chain ends (empty environments, tiny sector tables) included
    L = 8
    mpo = models.hamiltonian(models.OB_Sim([1.0, 0.3], [4.0]), L)
    bonds, _ = mps.random_mps(L, (L, 0), 5, seed=2)
    bonds = [pl.Bond(b) for b in bonds]
    for i in range(L - 1):
        tl = pl.ThetaLayout.build(bonds[i], bonds[i + 2])
        Ll = pl.EnvLayout.build("L", bonds[i], mpo[i].left)
        Rl = pl.EnvLayout.build("R", bonds[i + 2], mpo[i + 1].right)
        ref = pl.plan_apply(tl, Ll, Rl, mpo[i], mpo[i + 1])
        new = pl.plan_apply_cached(tl, Ll, Rl, mpo[i], mpo[i + 1])
        assert same(ref[0], new[0]) and same(ref[1], new[1]) and ref[2:] == new[2:], i


def test_vectorised_finalize_plan_equals_general_path():
    """plan_finalize's column-wise fast path (default staging) == the per-block general path, both placements"""
    import inspect
    rng = np.random.default_rng(1)
    tab = {(2, 0): 2, (2, 2): 1, (3, 1): 12, (3, 3): 3, (4, 0): 23, (4, 2): 24, (4, 4): 4, (5, 1): 33, (5, 3): 24,
           (6, 0): 14, (6, 2): 22, (6, 4): 13, (7, 1): 13, (7, 3): 4, (8, 0): 3}
    bl = pl.Bond({(N + 5, j): n for (N, j), n in tab.items()})
    br = pl.Bond({(N + 7, j): n for (N, j), n in tab.items()})
    tl = pl.ThetaLayout.build(bl, br)
    code = inspect.getsource(pl.plan_finalize).replace(
        "    if not any(sp.accumulate):\n        return _plan_finalize_fast(tl, sp, order, keep, layA, layB, placement, offA, offB)\n", "")
    assert "_plan_finalize_fast" not in code
    ns = dict(pl.__dict__)
    exec(code, ns)
    general = ns["plan_finalize"]
    for placement in ("right", "left"):
        sp = pl.plan_svd(tl, placement)
        svals = {c: np.sort(rng.random(int(sp.desc[i]["n"])))[::-1] for i, c in enumerate(sp.mids)}
        order = {c: rng.permutation(len(v)) for c, v in svals.items()}
        keep, tw, nrm = pl.truncate(svals, 200, 0.0, "sqrtdim")
        mid = pl.Bond({c: k for c, k in keep.items() if k > 0})
        layA, layB = pl.SiteLayout.build("L", bl, mid), pl.SiteLayout.build("R", mid, br)
        new = pl.plan_finalize(tl, sp, order, keep, layA, layB, placement, 0, layA.size)
        ref = general(tl, sp, order, keep, layA, layB, placement, 0, layA.size)
        for a, b in zip(ref[:4], new[:4]):
            assert a.dtype == b.dtype and np.array_equal(a, b)
        assert ref[4].tiles.tobytes() == new[4].tiles.tobytes() and ref[4].segs.tobytes() == new[4].segs.tobytes()
        assert ref[4].flops == new[4].flops


def test_symbolic_environment_plans_equal_loop_plans():
    """plan_env_cached (index structure cached by sector sets, instantiated by numpy gathers) == plan_left_env /
    plan_right_env byte for byte: every bond of a short chain (ends included) and perturbed bulk tables"""
    rng = np.random.default_rng(5)

    def same(a, b):
        return (a.ntiles == b.ntiles and a.nsegs == b.nsegs and a.flops == b.flops
                and a.tiles[:a.ntiles].tobytes() == b.tiles[:b.ntiles].tobytes()
                and a.segs[:a.nsegs].tobytes() == b.segs[:b.nsegs].tobytes())

    def check(b0, b1, W):
        layL, layR = pl.SiteLayout.build("L", b0, b1), pl.SiteLayout.build("R", b0, b1)
        Ll, Lnew = pl.EnvLayout.build("L", b0, W.left), pl.EnvLayout.build("L", b1, W.right)
        Rl, Rnew = pl.EnvLayout.build("R", b1, W.right), pl.EnvLayout.build("R", b0, W.left)
        ref, new = pl.plan_left_env(Ll, layL, W, Lnew), pl.plan_env_cached("L", Ll, layL, W, Lnew)
        assert same(ref[0], new[0]) and same(ref[1], new[1]) and ref[2] == new[2]
        ref, new = pl.plan_right_env(Rl, layR, W, Rnew), pl.plan_env_cached("R", Rl, layR, W, Rnew)
        assert same(ref[0], new[0]) and same(ref[1], new[1]) and ref[2] == new[2]
    L = 8
    for tt, uu in (([1.0], [4.0]), ([1.0, 0.3], [4.0, 0.5])):
        mpo = models.hamiltonian(models.OB_Sim(tt, uu), L)
        bonds = [pl.Bond(b) for b in mps.random_mps(L, (L, 0), 5, seed=2)[0]]
        for i in range(L):
            check(bonds[i], bonds[i + 1], mpo[i])
    tab = {(2, 0): 2, (2, 2): 1, (3, 1): 6, (3, 3): 3, (4, 0): 9, (4, 2): 8, (4, 4): 2, (5, 1): 11, (5, 3): 6, (6, 0): 7,
           (6, 2): 9, (6, 4): 3, (7, 1): 5, (7, 3): 2, (8, 0): 2}
    mpo = models.hamiltonian(models.OB_Sim([1.0, 0.1], [4.0]), 24)
    base0 = {(N + 8, j): n for (N, j), n in tab.items()}
    keys1 = sorted({b for c in base0 for s in range(3) for b in pl.fuse(c, s) if b[1] <= 4})
    base1 = {k: int(rng.integers(2, 12)) for k in keys1}
    for trial in range(3):                      # same sector sets, different multiplicities: the cached structure is reused
        pert = lambda n: max(1, int(n + rng.integers(-2, 3)))
        check(pl.Bond({k: pert(v) for k, v in base0.items()}), pl.Bond({k: pert(v) for k, v in base1.items()}), mpo[11])


def _dense_with_u13(L, t, u, mu, pairs_u13):
    """ed.dense_hamiltonian + sum over (a, b, U): U sum_s n_{b,-s} (c+_{a s} c_{b s} + h.c.)  (Jordan-Wigner matrices)"""
    from oracle import su2
    lm = su2.local_matrices()

    def site_op(mats):
        out = np.eye(1)
        for s in range(L):
            out = np.kron(out, mats.get(s, lm["id"]))
        return out

    def c_op(i, spin):
        mats = {s: lm["F"] for s in range(i)}
        mats[i] = lm["a_up"] if spin == 0 else lm["a_dn"]
        return site_op(mats)
    c = {(i, s): c_op(i, s) for i in range(L) for s in (0, 1)}
    H = ed.dense_hamiltonian(L, t, u, mu)
    for (a, b, U) in pairs_u13:
        for s in (0, 1):
            hop = c[(a, s)].T @ c[(b, s)]
            H = H + U * (c[(b, 1 - s)].T @ c[(b, 1 - s)]) @ (hop + hop.T)
    return H


def test_three_equal_index_terms_equal_dense_density_assisted_hopping():
    """U13 of OB_Sim (src:452-458) and Uijjj_OS / Uijjj_IS of MB_Sim (src:617-649, 703-730): the MPO equals the dense
    sum_s n_{b,-s}(c+_{a s} c_{b s} + h.c.) form; reduced elements of the density-assisted ladder operators equal the
    oracle's Wigner-Eckart reduction of their Jordan-Wigner matrices.  (The reference's tests never switch these terms
    on: parity against the reference itself is unpinned, models._assisted_hop.)"""
    from oracle import su2
    ops = su2.site_operators()
    for name in ("cdag_d", "cdagF_d", "c_d", "Fc_d"):
        k, dN, red = ops[name]
        k2, dN2, red2 = models.SITE_OPS[name]
        assert (k, dN) == (k2, dN2) and np.abs(red - red2).max() < 1e-14
    L, t, u, U13 = 4, [1.0, 0.2], [4.0, 0.5], [0.3, 0.1]
    M = ompo.mpo_to_dense(_as_dict(models.hamiltonian(models.OB_Sim(t, u, 0.1, 1, 1, U13=U13), L)))
    pairs = [(i, i + r, Ur) for r, Ur in enumerate(U13, start=1) for i in range(L - r)]
    pairs += [(b, a, U) for (a, b, U) in pairs]
    H = _dense_with_u13(L, t, u, 0.1, pairs)
    assert np.abs(M - H).max() < 1e-12 and np.abs(M - M.T).max() < 1e-13
    # two bands, two cells: on-site inter-band U13_OS (ordered pairs, density on the second index) and inter-cell U13_IS
    B, cells = 2, 2
    tm = np.array([[0.1, 0.7, 0.4, 0.0], [0.7, -0.2, 0.3, 0.2]])
    um = np.array([[3.0, 1.0, 0.0, 0.0], [1.0, 2.5, 0.0, 0.0]])
    U13_OS = np.array([[0.0, 0.25], [-0.15, 0.0]])
    U13_IS = np.zeros((B, B, 4))
    U13_IS[0, 1] = [0.2, 0.1, -0.3, 0.05]
    U13_IS[1, 1] = [0.4, 0.0, 0.0, 0.2]
    sim = models.MB_Sim(tm, um, np.zeros((B, 2 * B)), U13_OS, 1, 1, 2.0, 8, U13_IS=U13_IS)
    M = ompo.mpo_to_dense(_as_dict(models.hamiltonian(sim, cells)))
    site = lambda b, c: b + c * B
    n = B * cells
    # reference Hamiltonian without the three-index terms, from the same builder (already pinned densely elsewhere)
    M0 = ompo.mpo_to_dense(_as_dict(models.hamiltonian(models.MB_Sim(tm, um, np.zeros((B, 2 * B)), 1, 1, 2.0, 8), cells)))
    pairs = []
    for c_ in range(cells):
        for bi in range(B):
            for bf in range(B):
                if bi != bf:
                    pairs.append((site(bi, c_), site(bf, c_), U13_OS[bi, bf]))
    for bi in range(B):
        for bf in range(B):
            i, j = site(bi, 0), site(bf, 1)
            pairs.append((i, j, 0.5 * (U13_IS[bi, bf, 0] + U13_IS[bi, bf, 1])))
            pairs.append((j, i, 0.5 * (U13_IS[bi, bf, 2] + U13_IS[bi, bf, 3])))
    D = _dense_with_u13(n, [0.0], [0.0], 0.0, pairs)
    assert np.abs((M - M0) - D).max() < 1e-12


def test_merged_channel_mpo_is_the_same_operator_on_a_narrower_bond():
    """terms of one channel type that open on the same site share a level (models._build_mpo merge=True, the default):
    same dense Hamiltonian as the reference-style uncompressed sum of per-term MPOs (`H += h`, src:439), fewer levels"""
    t, u, J = [1.0, 0.3, 0.1], [4.0, 0.5, 0.2], [0.2, 0.1]
    L = 5
    sim = models.OB_Sim(t, u, 0.1, J, 1, 1, U13=[0.15, 0.05])
    Hm = models.hamiltonian(sim, L)
    onsite = {s: [("docc", u[0]), ("n", -0.1)] for s in range(L)}
    pairs = []
    for r, tr in enumerate(t, start=1):
        pairs += [(i, i + r, "hop", -tr) for i in range(L - r)]
    for r in range(1, len(u)):
        pairs += [(i, i + r, "nn", u[r]) for i in range(L - r)]
    for r, Jr in enumerate(J, start=1):
        for i in range(L - r):
            models._exchange(pairs, i, i + r, Jr)
    for r, Ur in enumerate([0.15, 0.05], start=1):
        for i in range(L - r):
            models._assisted_hop(pairs, i, i + r, Ur)
            models._assisted_hop(pairs, i + r, i, Ur)
    Hu = models._build_mpo(L, onsite, pairs, merge=False)
    Dm, Du = ompo.mpo_to_dense(_as_dict(Hm)), ompo.mpo_to_dense(_as_dict(Hu))
    assert np.abs(Dm - Du).max() < 1e-12
    wm, wu = max(len(W.right) for W in Hm), max(len(W.right) for W in Hu)
    assert wm < wu and wm <= 0.7 * wu
    # range-3 hopping alone: 3 levels per channel type instead of 6
    H3 = models.hamiltonian(models.OB_Sim([1.0, 0.3, 0.1], [4.0]), 10)
    assert len(H3[5].left) == 2 + 2 * 3


def test_three_and_four_index_terms_as_reduced_operator_strings():
    """U112 / U1111 of MB_Sim (Uijkk / Uijkl, src:732-809) in the SU(2) x U(1) mode: products of two spin-summed hoppings
    become strings of reduced site operators with MPO levels of spin 0, 1/2 and 1 in between -- operators that couple two
    non-trivial spins (hubbardtn_amd/string_table.py, generated and fitted exactly by tests/golden/make_string_table.py).
    Pinned here: the reduced MPO, expanded with explicit Clebsch-Gordan tensors, equals the spinful mode's Jordan-Wigner
    operator strings (pinned against second quantisation in tests/test_spinful_cpu.py) for orbitals in every relative
    order, with gaps between them and across cells; and the C++ engine on that MPO finds the dense ground state of the
    (N, S = 0) sector (mid-string entries: level spin x operator rank -> level spin, all three non-zero)."""
    from oracle import su2
    from cpu_ops import CpuOps
    from test_spinful_cpu import _dense_from_abelian_mpo
    from hubbardtn_amd import engine as cengine
    B, cells = 1, 5                                        # one band: orbital o sits on chain site o - 1 (+ cell)
    n = B * cells
    tm = np.array([[0.1, 1.0, 0.3]])
    um = np.array([[3.0, 0.5, 0.0]])
    rng = np.random.default_rng(0)
    U1111 = {key: float(rng.standard_normal()) for key in [(1, 2, 3, 4), (2, 4, 1, 3), (1, 3, 4, 2), (4, 1, 2, 3), (1, 4, 5, 2), (2, 5, 3, 1),
                                                             (1, 2, 4, 5), (5, 1, 4, 2)]}
    U112 = {(1, 2, 3, 3): 0.25, (1, 3, 3, 4): -0.15, (2, 3, 1, 3): 0.1, (3, 1, 4, 4): 0.3, (4, 2, 2, 1): -0.2, (3, 2, 1, 2): 0.17,
            (1, 5, 5, 2): 0.11, (1, 3, 5, 3): -0.07, (4, 1, 1, 2): 0.21}
    mk = lambda **kw: models.MB_Sim(tm, um, np.zeros((B, 2 * B)), 1, 1, 2.0, 8, **kw)
    Ma = _dense_from_abelian_mpo(models.hamiltonian(mk(spin=True, U1111=U1111, U112=U112), cells))
    Hs = models.hamiltonian(mk(U1111=U1111, U112=U112), cells)
    Ms = ompo.mpo_to_dense(_as_dict(Hs))
    assert np.abs(Ma - Ms).max() < 1e-12
    # Hermitian part only for the eigenvalue check: the random U1111 above is not symmetrised (U[(i,j,k,l)] != U[(l,k,j,i)])
    U1111h = {}
    for (i, j, k, l), v in U1111.items():
        U1111h[(i, j, k, l)] = U1111h.get((i, j, k, l), 0.0) + 0.5 * v
        U1111h[(l, k, j, i)] = U1111h.get((l, k, j, i), 0.0) + 0.5 * v
    H = models.hamiltonian(mk(U1111=U1111h, U112=U112), cells)
    M = ompo.mpo_to_dense(_as_dict(H))
    assert np.abs(M - M.T).max() < 1e-12
    Nel = n - 1                                            # (an even electron number: the singlet sector exists)
    bonds, tens = mps.random_mps(n, (Nel, 0), 64, seed=4)
    eng = cengine.DMRG2(CpuOps(), H, bonds, tens, chi_full=None, lanczos_tol=1e-12)
    for _ in range(4):
        E0 = eng.sweep()
    lm = su2.local_matrices()

    def site_op(mats):
        out = np.eye(1)
        for s in range(n):
            out = np.kron(out, mats.get(s, lm["id"]))
        return out

    def c_op(i, spin):
        mats = {s: lm["F"] for s in range(i)}
        mats[i] = lm["a_up"] if spin == 0 else lm["a_dn"]
        return site_op(mats)
    c = {(i, s): c_op(i, s) for i in range(n) for s in (0, 1)}
    Ntot = sum(c[k].T @ c[k] for k in c)
    Sp = sum(c[(i, 0)].T @ c[(i, 1)] for i in range(n))
    Sz = 0.5 * sum(c[(i, 0)].T @ c[(i, 0)] - c[(i, 1)].T @ c[(i, 1)] for i in range(n))
    S2 = Sp @ Sp.T + Sz @ Sz - Sz
    Id = np.eye(4 ** n)
    w = np.linalg.eigvalsh(M + 50.0 * (Ntot - Nel * Id) @ (Ntot - Nel * Id) + 50.0 * S2)
    assert abs(E0 - w[0]) < 1e-9 * max(abs(w[0]), 1.0)


def test_mpo_compression_keeps_the_operator_and_narrows_the_bond():
    """models._compress_mpo (exact deparallelisation: levels with proportional futures or pasts are one level): the dense
    operator is unchanged to rounding; the plain hopping chain is already minimal in the bulk; operator strings that share
    prefixes / suffixes (U112 / U1111) lose more than half of their levels; level labels stay consistent (every entry
    connects labels the operator's charge allows)"""
    def both(sim, L):
        H = models.hamiltonian(sim, L)
        keep = models._compress_mpo
        models._compress_mpo = lambda H_: H_
        try:
            H0 = models.hamiltonian(sim, L)
        finally:
            models._compress_mpo = keep
        return H0, H
    H0, H = both(models.OB_Sim([1.0], [4.0]), 6)
    assert [len(W.left) for W in H] == [len(W.left) for W in H0] == [1, 4, 4, 4, 4, 4]
    H0, H = both(models.OB_Sim([1.0, 0.3, 0.1], [4.0, 0.5, 0.2], 0.1, [0.3, 0.1], 1, 1), 6)
    assert np.abs(ompo.mpo_to_dense(_as_dict(H)) - ompo.mpo_to_dense(_as_dict(H0))).max() < 1e-13
    assert [len(W.left) for W in H][2:5] == [len(W.left) for W in H0][2:5]              # bulk untouched, edges pruned
    tm, um = np.array([[0.1, 1.0, 0.3]]), np.array([[3.0, 0.5, 0.0]])
    U1111 = {(1, 2, 3, 4): 0.31, (4, 3, 2, 1): 0.31, (1, 3, 4, 2): -0.2, (2, 4, 3, 1): -0.2, (1, 2, 4, 5): 0.15, (5, 4, 2, 1): 0.15}
    U112 = {(1, 2, 3, 3): 0.25, (1, 3, 3, 4): -0.15, (2, 3, 1, 3): 0.1, (1, 5, 5, 2): 0.11}
    H0, H = both(models.MB_Sim(tm, um, np.zeros((1, 2)), 1, 1, 2.0, 8, U1111=U1111, U112=U112), 6)
    assert np.abs(ompo.mpo_to_dense(_as_dict(H)) - ompo.mpo_to_dense(_as_dict(H0))).max() < 1e-13
    assert 2 * max(len(W.left) for W in H) < max(len(W.left) for W in H0)
    for W in H:
        for (wl, wr, op, c) in W.entries:
            k, dN, _ = H.sym.site_ops[op]
            assert W.right[wr][0] == W.left[wl][0] + dN and abs(W.left[wl][1] - k) <= W.right[wr][1] <= W.left[wl][1] + k


@pytest.mark.parametrize("period,L", [(3, 5), (4, 5), (2, 4)])
def test_helix_model_equals_the_dense_hamiltonian_with_two_hopping_ranges(period, L):
    """`period != 0` (src:464-466): -t (cdc + h.c.) on {i, i+1} AND on {i, i+period} -- a chain wound into a helix / cylinder of
    that circumference.  On the open chain this is the dense Hamiltonian with hopping t at range 1 and at range `period`;
    anything but nearest-neighbour parameters is refused with the reference's message (src:468)."""
    t, U, mu = 0.8, 3.0, 0.2
    H = models.hamiltonian(models.OB_Sim([t], [U], mu, 1, 1, 2.0, 50, period), L)
    tvec = [0.0] * period
    tvec[0] += t
    tvec[period - 1] += t
    assert np.abs(ompo.mpo_to_dense(_as_dict(H)) - ed.dense_hamiltonian(L, tvec, [U], mu)).max() < 1e-13
    with pytest.raises(ValueError, match="Extended models in 2D not implemented."):
        models.hamiltonian(models.OB_Sim([1.0, 0.1], [4.0], 0.0, 1, 1, 2.0, 50, period), L)
