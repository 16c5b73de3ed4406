"""Host API mirroring the reference's call sequence on the hot path:

    model = OB_Sim(t, u, mu, P, Q, svalue, bond_dim; spin=false)         src/HubbardFunctions.jl:76-93
    dictionary = produce_groundstate(model)                              src:1145-1166
    psi, H = dictionary["groundstate"], dictionary["ham"]
    E = sum(real(expectation_value(psi, H))) / length(H)                 examples/One_band.jl:42-43
    dim_state(psi)                                                       src:1399-1405

and the plugin boundary MPSKit exposes to it,

    find_groundstate(psi0, H, alg) -> (psi, envs, delta)                 src:1010

with `alg = DMRG2(trscheme=truncdim(D) | truncbelow(eta), tol, maxiter, verbosity)`.

Two drivers share the hot path: with a chain length (`L=` keyword / `simul.kwargs["L"]`) the
finite two-site sweep DMRG2 that BASELINE.json's L=... configs name; without one the infinite-chain IDMRG2 the
reference itself calls (src:1010), here in McCulloch's growing-window form (hubbardtn_amd/idmrg.py) with the
reference's truncation default truncbelow(10^-svalue).  All compute goes through the HIP library; there is no CPU
path.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import engine as _engine
from . import idmrg as _idmrg
from . import models, mps
from .models import MB_Sim, MBC_Sim, OB_Sim, OBC_Sim, OBC_Sim2, Simulation


# ---- truncation schemes (TensorKit names; App. A.6) --------------------------------------------
@dataclass
class truncdim:
    """keep at most D in TensorKit `dim` units (sum (2S+1) n); used at src:1363-1365"""
    D: int


@dataclass
class truncbelow:
    """keep Schmidt values > eta; used at src:1010 with eta = 10^-svalue"""
    eta: float


@dataclass
class DMRG2:
    """two-site DMRG algorithm selector (MPSKit.DMRG2 / IDMRG2 keyword names)"""
    trscheme: object = None
    tol: float = 1e-6              # src:993 default of compute_groundstate
    maxiter: int = 100
    verbosity: int = 0
    krylovdim: int = 30            # KrylovKit / MPSKit defaults (App. A.4)
    eigsolve_tol: float = 1e-10


@dataclass
class IDMRG2:
    """infinite two-site DMRG selector (MPSKit.IDMRG2 keyword names; the call at src:1010)"""
    trscheme: object = None
    tol: float = 1e-6              # on the change of the centre Schmidt spectrum (MPSKit: ||C_old - C_new||)
    maxiter: int = 100
    verbosity: int = 0
    krylovdim: int = 30
    eigsolve_tol: float = 1e-10
    sweeps_per_step: int = 6       # at most this many finite sweeps of the 2-cell window per growth step


@dataclass
class InfiniteHamiltonian:
    """the model of a translation-invariant chain; len() = sites per unit cell (length(H) in the reference)"""
    simul: Simulation

    def __len__(self):
        return _idmrg.cell_sites(self.simul)


@dataclass
class InfiniteMPS:
    """handle of an infinite MPS: before optimisation only the start parameters, afterwards the IDMRGResult"""
    max_dimension: int
    seed: int = 1234
    ops: object = None
    result: object = None

    def bond_dimensions(self):
        if self.result is None:
            raise RuntimeError("run find_groundstate first")
        T = self.result.unit_cell
        return list(self.result.bond_dims[T // 2 + 1:T // 2 + 1 + T])     # the central unit cell of the last window


@dataclass
class FiniteMPS:
    """device-resident finite MPS handle (wraps the sweep engine's state)"""
    engine: object
    L: int

    def bond_dimensions(self):
        return self.engine.bond_dims()


@dataclass
class Environments:
    engine: object


_OPS = None


def _ops(device=0):
    global _OPS
    if _OPS is None:
        from .device import HipOps        # raises if libhubbardtn_hip.so or the GPU is missing
        _OPS = HipOps(device)
    return _OPS


def hamiltonian(simul: Simulation, L: int | None = None):
    """finite open chain of L unit cells -> list of MPO sites; no L -> the infinite chain (src:386-472, 811-910)"""
    L = L or simul.kwargs.get("L")
    if L is None:
        return InfiniteHamiltonian(simul)
    return models.hamiltonian(simul, int(L))


def initialize_mps(H, P: int, max_dimension: int | None = None, spin: bool = False, Q: int = 1, seed: int = 1234, ops=None):
    """random right-canonical start with per-sector cap `max_dimension` (src:917-959); the symmetry mode (SU(2) x U(1),
    or U(1) x U(1) for `spin=true`) is the Hamiltonian's.  The chemical-potential models use the reference's two-argument
    form `initialize_mps(operator, max_dimension)` (src:961-991)."""
    if isinstance(H, InfiniteHamiltonian) and models.symmetry_of(H.simul).kind == 2 or getattr(H, "sym", None) is models.SU2P:
        # (parity 0, S = 0) total sector on a finite chain: an even number of electrons in a singlet
        max_dimension = P if max_dimension is None else max_dimension
        if isinstance(H, InfiniteHamiltonian):
            return InfiniteMPS(int(max_dimension), seed, ops)
        bonds, tensors = mps.random_mps(len(H), (0, 0), int(max_dimension), seed=seed, sym=models.SU2P)
        return FiniteMPS(_engine.DMRG2(ops or _ops(), H, bonds, tensors), len(H))
    if isinstance(H, InfiniteHamiltonian):
        if bool(spin) != (not models.symmetry_of(H.simul).su2):
            raise ValueError("initialize_mps: `spin` does not match the Hamiltonian's symmetry mode")
        return InfiniteMPS(int(max_dimension), seed, ops)
    if max_dimension is None:
        raise TypeError("initialize_mps(H, P, max_dimension, spin, Q): max_dimension is required for the fixed-filling models")
    nsites = len(H)
    sym = getattr(H, "sym", models.SU2U1)
    if bool(spin) != (not sym.su2):
        raise ValueError("initialize_mps: `spin` does not match the Hamiltonian's symmetry mode")
    if (nsites * P) % Q:
        raise ValueError("filling P/Q incompatible with the chain length")
    target = (nsites * P // Q, 0)                     # total spin 0 / total Sz 0
    bonds, tensors = mps.random_mps(nsites, target, max_dimension, seed=seed, sym=sym)
    eng = _engine.DMRG2(ops or _ops(), H, bonds, tensors)
    return FiniteMPS(eng, nsites)


def find_groundstate(psi: FiniteMPS, H, alg: DMRG2, envs=None):
    """-> (psi, envs, delta); delta = |E_sweep - E_previous sweep| / L at exit (MPSKit returns the
    last convergence error).  Sweeps until delta < alg.tol or maxiter.  With an InfiniteMPS / IDMRG2: growth steps
    until the centre Schmidt spectrum changes by less than alg.tol."""
    if isinstance(psi, InfiniteMPS):
        if not isinstance(alg, IDMRG2) or not isinstance(H, InfiniteHamiltonian):
            raise TypeError("an InfiniteMPS is optimised with IDMRG2 on hamiltonian(simul) without a chain length")
        chi, cut = None, 0.0
        if isinstance(alg.trscheme, truncdim):
            chi = int(alg.trscheme.D)
        elif isinstance(alg.trscheme, truncbelow):
            cut = float(alg.trscheme.eta)
        elif alg.trscheme is not None:
            raise TypeError("trscheme must be truncdim(D) or truncbelow(eta)")
        psi.result = _idmrg.idmrg2(psi.ops or _ops(), H.simul, chi_full=chi, cutoff=cut, tol=alg.tol,
                                   maxiter=alg.maxiter, sweeps_per_step=alg.sweeps_per_step,
                                   init_dimension=psi.max_dimension, krylovdim=alg.krylovdim,
                                   lanczos_tol=alg.eigsolve_tol, seed=psi.seed, verbosity=alg.verbosity)
        return psi, Environments(psi.result.engine), psi.result.delta
    eng = psi.engine
    if isinstance(alg.trscheme, truncdim):
        eng.chi_full, eng.cutoff = int(alg.trscheme.D), 0.0
    elif isinstance(alg.trscheme, truncbelow):
        eng.chi_full, eng.cutoff = None, float(alg.trscheme.eta)
    elif alg.trscheme is not None:
        raise TypeError("trscheme must be truncdim(D) or truncbelow(eta)")
    eng.krylovdim, eng.lanczos_tol = alg.krylovdim, alg.eigsolve_tol
    E_prev, delta = None, float("inf")
    for it in range(alg.maxiter):
        E = eng.sweep()
        if E_prev is not None:
            delta = abs(E - E_prev) / psi.L
        if alg.verbosity:
            print(f"DMRG2 sweep {it + 1}: E/L = {E / psi.L:.12f}  delta = {delta:.3e}  chi = {max(eng.bond_dims())}")
        E_prev = E
        if delta < alg.tol:
            break
    return psi, Environments(eng), delta


def compute_groundstate(simul: Simulation, L: int | None = None, tol: float = 1e-6, verbosity: int = 0,
                        maxiter: int = 100, init_state=None, chi: int | None = None):
    """src:993-1030 restated for the finite chain: H = hamiltonian(simul); psi0 = initialize_mps(...);
    find_groundstate(psi0, H, DMRG2(trscheme = truncbelow(10^-svalue))) -- or truncdim(chi) when a
    fixed bond dimension is requested (the configs of BASELINE.json)."""
    H = hamiltonian(simul, L)
    spin = bool(simul.kwargs.get("spin", False))
    psi0 = init_state if init_state is not None else initialize_mps(H, simul.P, simul.bond_dim, spin, simul.Q)
    scheme = truncdim(chi) if chi is not None else truncbelow(10.0 ** (-simul.svalue))
    if isinstance(H, InfiniteHamiltonian):       # src:1010; the VUMPS / GradientGrassmann polish (src:1025-1027) is out of scope
        if chi is None and _idmrg.reference_cell_sites(simul) == 1:
            # the reference's branch for length(H) == 1 (src:1012-1022): VUMPS at fixed space, the space grown by VUMPSSvdCut and
            # cut by SvdCut(truncbelow(10^-svalue)) until it stops changing.  VUMPS-free route: the same self-consistent space is
            # the fixed point of two-site updates with that Schmidt cut -- IDMRG2 -- once the cut is expressed for the doubled cell
            # this library uses (idmrg.schmidt_cut_scale: one sector family per bond instead of both at half weight)
            scheme = truncbelow(_idmrg.schmidt_cut_scale(simul) * 10.0 ** (-simul.svalue))
        alg = IDMRG2(trscheme=scheme, tol=tol, verbosity=verbosity, maxiter=maxiter)
    else:
        alg = DMRG2(trscheme=scheme, tol=tol, verbosity=verbosity, maxiter=maxiter)
    psi, envs, delta = find_groundstate(psi0, H, alg)
    return {"groundstate": psi, "environments": envs, "ham": H, "delta": delta, "config": simul}


def produce_groundstate(simul: Simulation, force: bool = False, **kw):
    """src:1145-1166: computes, or loads the result saved under the reference's cache name (storage.produce_or_load)"""
    from . import storage
    return storage.produce_or_load(compute_groundstate, simul, force=force, **kw)


def expectation_value(psi: FiniteMPS, H):
    """<psi|H|psi> per site as a vector whose sum / length(H) is the energy per site (examples/One_band.jl:42-43 take
    sum(real(E0)) / length(H); test/OB.jl:28-29).  Finite chain: a genuine expectation value of the state AS STORED
    (after truncation), evaluated by a non-optimising pass through the library; entry i = energy of the terms ending on
    site i (engine.DMRG2.site_energies).  Infinite chain: the energy density of the converged window, the same for
    every site of the unit cell: differences of <psi_n|H_n|psi_n> of successive (truncated) window states."""
    if isinstance(psi, InfiniteMPS):
        if psi.result is None:
            raise RuntimeError("run find_groundstate first")
        return np.full(len(H), psi.result.energy_per_site)
    if psi.engine.energy is None:
        raise RuntimeError("run find_groundstate first")
    return psi.engine.site_energies()


def _save_result(res, entry):
    """cache entry of produce_groundstate: the site tensors (storage.save_state format) + what re-creates the handle"""
    import json
    import os
    from . import storage
    psi = res["groundstate"]
    storage.save_state(psi, os.path.dirname(entry), os.path.basename(entry))
    meta = {"delta": float(res["delta"])}
    if isinstance(psi, InfiniteMPS):
        r = psi.result
        meta.update(kind="infinite", energy_per_site=r.energy_per_site, iterations=r.iterations, unit_cell=r.unit_cell,
                    history=[list(h) for h in r.history], max_dimension=psi.max_dimension, seed=psi.seed,
                    spectrum=[[N, j, [float(x) for x in v]] for (N, j), v in sorted(r.spectrum.items())],
                    bL=[[N, j, n] for (N, j), n in sorted(r.boundary["bL"].items())],
                    bR=[[N, j, n] for (N, j), n in sorted(r.boundary["bR"].items())],
                    chi_full=r.engine.chi_full, cutoff=r.engine.cutoff)
        np.save(os.path.join(entry, "left_env.npy"), r.boundary["Lenv"])
        np.save(os.path.join(entry, "right_env.npy"), r.boundary["Renv"])
    else:
        eng = psi.engine
        meta.update(kind="finite", L=psi.L, energy=eng.energy, chi_full=eng.chi_full, cutoff=eng.cutoff)
    with open(os.path.join(entry, "result.json"), "w") as f:
        json.dump(meta, f)


def _load_result(simul, entry, L=None, ops=None, **kw):
    """the result dictionary of compute_groundstate re-created from a cache entry: the state is uploaded, nothing is
    recomputed"""
    import json
    import os
    from . import storage
    meta = json.load(open(os.path.join(entry, "result.json")))
    bonds, sites = storage.load_state(entry)
    tensors = [s["blocks"] for s in sites]
    H = hamiltonian(simul, L)
    ops = ops or _ops()
    if meta["kind"] == "infinite":
        T = meta["unit_cell"]
        big = models.hamiltonian(simul, 8 * max(meta["unit_cell"] // simul.bands, 1))
        window = [big[3 * T + i] for i in range(2 * T)]
        eng = _engine.DMRG2(ops, window, bonds, tensors, chi_full=meta["chi_full"], cutoff=meta["cutoff"],
                            left_env=np.load(os.path.join(entry, "left_env.npy")),
                            right_env=np.load(os.path.join(entry, "right_env.npy")))
        spec = {(N, j): np.asarray(v) for N, j, v in meta["spectrum"]}
        res = _idmrg.IDMRGResult(energy_per_site=meta["energy_per_site"], delta=meta["delta"], iterations=meta["iterations"],
                                 unit_cell=T, bond_dims=eng.bond_dims(), spectrum=spec,
                                 history=[tuple(h) for h in meta["history"]], engine=eng,
                                 boundary={"bL": {(N, j): n for N, j, n in meta["bL"]}, "bR": {(N, j): n for N, j, n in meta["bR"]},
                                           "Lenv": np.load(os.path.join(entry, "left_env.npy")),
                                           "Renv": np.load(os.path.join(entry, "right_env.npy"))})
        psi = InfiniteMPS(meta["max_dimension"], meta["seed"], ops, res)
        return {"groundstate": psi, "environments": Environments(eng), "ham": H, "delta": meta["delta"], "config": simul}
    eng = _engine.DMRG2(ops, H, bonds, tensors, chi_full=meta["chi_full"], cutoff=meta["cutoff"])
    eng.energy = meta["energy"]
    psi = FiniteMPS(eng, meta["L"])
    return {"groundstate": psi, "environments": Environments(eng), "ham": H, "delta": meta["delta"], "config": simul}


def dim_state(psi: FiniteMPS):
    """bond dimensions in TensorKit `dim` units (src:1399-1405)"""
    if isinstance(psi, InfiniteMPS):
        return psi.bond_dimensions()
    return psi.engine.bond_dims()[1:]


def density_state(psi):
    """electrons per site <n_i> (src:1475-1523): all L sites of a finite chain, the unit cell of an infinite one
    (its sum / len equals the filling P/Q, the check at test/OB.jl:97-99)"""
    if isinstance(psi, InfiniteMPS):
        if psi.result is None:
            raise RuntimeError("run find_groundstate first")
        T = psi.result.unit_cell
        n, _ = psi.result.engine.site_occupations()
        return n[T // 2:T // 2 + T]
    return psi.engine.site_occupations()[0]


def density_spin(psi):
    """(n_up, n_dn) per site (src:1412-1456): spinful U(1) x U(1) mode only -- the SU(2) mode raises the reference's
    "This system is spin independent." -- finite chain: all sites, infinite chain: the unit cell"""
    if isinstance(psi, InfiniteMPS):
        if psi.result is None:
            raise RuntimeError("run find_groundstate first")
        T = psi.result.unit_cell
        up, dn = psi.result.engine.spin_occupations()
        return up[T // 2:T // 2 + T], dn[T // 2:T // 2 + T]
    return psi.engine.spin_occupations()


def calc_ms(psi):
    """staggered magnetisation |<n_up - n_dn>| of the first site (src:1458-1473; warns like the reference when the
    magnitude is not uniform)"""
    import warnings
    up, dn = density_spin(psi)
    mag = up - dn
    if not np.allclose(np.abs(mag), abs(mag[0]), rtol=1e-6, atol=1e-12):
        warnings.warn("Spin-density wave?")
    return float(abs(mag[0]))


def double_occupancy(psi):
    """<n_up n_dn> per site, same conventions as density_state"""
    if isinstance(psi, InfiniteMPS):
        T = psi.result.unit_cell
        return psi.result.engine.site_occupations()[1][T // 2:T // 2 + T]
    return psi.engine.site_occupations()[1]


def TruncState(simul: Simulation, trunc_dim: int, trunc_scheme: int = 0, L: int | None = None, **kw):
    """truncated approximation of the ground state at bond dimension `trunc_dim` (TensorKit dim units), src:1351-1367.
    trunc_scheme 1 = SvdCut (truncate by SVD only); 0 = VUMPSSvdCut (truncate, then re-optimise variationally at
    that dimension -- here: further two-site sweeps with truncdim(trunc_dim)).  Finite chains (L given) and the infinite
    chain: scheme 1 cuts the bonds of the converged window, scheme 0 = IDMRG2 at truncdim(trunc_dim)."""
    if trunc_dim <= 0:
        raise ValueError("trunc_dim should be a positive integer.")
    if trunc_scheme not in (0, 1):
        raise ValueError("trunc_scheme should be either 0 (VUMPSSvdCut) or 1 (SvdCut).")
    L = L or simul.kwargs.get("L")
    if L is None:
        if trunc_scheme == 1:
            # SvdCut of the infinite state (test/MB.jl:95-103): every bond inside the converged window -- two unit cells between
            # the environments of the half-infinite blocks -- is cut to truncdim(trunc_dim) by its own Schmidt decomposition,
            # without re-optimisation; the energy density moves by the window's energy change per site
            d = compute_groundstate(simul, **kw)
            psi = d["groundstate"]
            r, eng = psi.result, psi.result.engine
            E_before, _ = eng.bond_energies()
            E_after = eng.svd_cut(trunc_dim)
            r.bond_dims = eng.bond_dims()
            r.energy_per_site += (E_after - E_before) / eng.L
            r.spectrum = eng.spectrum(r.unit_cell)
            return {"ψ_trunc": psi, "envs_trunc": Environments(eng)}
        d = compute_groundstate(simul, chi=trunc_dim, **kw)
        return {"ψ_trunc": d["groundstate"], "envs_trunc": d["environments"]}
    d = produce_groundstate(simul, L=L, **kw)
    psi = d["groundstate"]
    eng = psi.engine
    eng.svd_cut(trunc_dim)
    if trunc_scheme == 0:
        E_prev = None
        for _ in range(kw.get("maxiter", 20)):
            E = eng.sweep()
            if E_prev is not None and abs(E - E_prev) / psi.L < kw.get("tol", 1e-6):
                break
            E_prev = E
    return {"ψ_trunc": psi, "envs_trunc": Environments(eng)}


def produce_TruncState(simul: Simulation, trunc_dim: int, trunc_scheme: int = 0, force: bool = False, **kw):
    """src:1378-1385 without the DrWatson disk cache"""
    return TruncState(simul, trunc_dim, trunc_scheme=trunc_scheme, **kw)
