"""Host API mirroring the reference's call sequence on the hot path:

    model = OB_Sim(t, u, mu, P, Q, svalue, bond_dim; spin=false)         src/HubbardFunctions.jl:76-93
    dictionary = produce_groundstate(model)                              src:1145-1166
    psi, H = dictionary["groundstate"], dictionary["ham"]
    E = sum(real(expectation_value(psi, H))) / length(H)                 examples/One_band.jl:42-43
    dim_state(psi)                                                       src:1399-1405

and the plugin boundary MPSKit exposes to it,

    find_groundstate(psi0, H, alg) -> (psi, envs, delta)                 src:1010

with `alg = DMRG2(trscheme=truncdim(D) | truncbelow(eta), tol, maxiter, verbosity)`.

Differences forced by scope: the reference only runs INFINITE chains (IDMRG2); this
engine runs the finite-chain two-site sweep that BASELINE.json's L=... configs name, so a chain
length must be given (`L=` keyword / `simul.kwargs["L"]`).  All compute goes through the HIP
library; there is no CPU path.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import engine as _engine
from . import models, mps
from .models import MB_Sim, OB_Sim, Simulation


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
    L = L or simul.kwargs.get("L")
    if L is None:
        raise ValueError("finite-chain engine: pass L (number of unit cells)")
    return models.hamiltonian(simul, int(L))


def initialize_mps(H, P: int, max_dimension: int, spin: bool = False, Q: int = 1, seed: int = 1234, ops=None):
    """random right-canonical start with per-sector cap `max_dimension` (src:917-959)"""
    if spin:
        raise NotImplementedError("U(1)xU(1) spinful mode is a 'next' row (SURVEY 8f.2)")
    nsites = len(H)
    if (nsites * P) % Q:
        raise ValueError("filling P/Q incompatible with the chain length")
    target = (nsites * P // Q, 0)
    bonds, tensors = mps.random_mps(nsites, target, max_dimension, seed=seed)
    eng = _engine.DMRG2(ops or _ops(), H, bonds, tensors)
    return FiniteMPS(eng, nsites)


def find_groundstate(psi: FiniteMPS, H, alg: DMRG2, envs=None):
    """-> (psi, envs, delta); delta = |E_sweep - E_previous sweep| / L at exit (MPSKit returns the
    last convergence error).  Sweeps until delta < alg.tol or maxiter."""
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
    psi, envs, delta = find_groundstate(psi0, H, DMRG2(trscheme=scheme, tol=tol, verbosity=verbosity, maxiter=maxiter))
    return {"groundstate": psi, "environments": envs, "ham": H, "delta": delta, "config": simul}


def produce_groundstate(simul: Simulation, force: bool = False, **kw):
    """src:1145-1166 without the DrWatson disk cache (out of scope): always computes"""
    return compute_groundstate(simul, **kw)


def expectation_value(psi: FiniteMPS, H):
    """energy per site as a length-L vector whose sum / L is E/L (examples/One_band.jl:42-43 take
    sum(real(E0)) / length(H)); the finite engine knows the total energy from its last eigensolve."""
    E = psi.engine.energy
    if E is None:
        raise RuntimeError("run find_groundstate first")
    return np.full(psi.L, E / psi.L)


def dim_state(psi: FiniteMPS):
    """bond dimensions in TensorKit `dim` units (src:1399-1405)"""
    return psi.engine.bond_dims()[1:]
