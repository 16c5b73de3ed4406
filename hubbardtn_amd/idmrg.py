"""Infinite-chain two-site DMRG (the `IDMRG2` the reference calls at src/HubbardFunctions.jl:1010), SURVEY 8f.1.

MPSKit's IDMRG2 sweeps the sites of ONE unit cell with environments that grow by transfer at every pass
(no linear solve inside the loop) and stops on ||C_old - C_new|| < tol.  The same fixed point is reached here in
McCulloch's original formulation, which needs nothing but the finite-chain engine:

    system_n = [left block] + window of 2 T sites + [right block]              (T = unit cell, src:408-412)
    optimise the window by ordinary two-site sweeps (engine.DMRG2 with the blocks' environments as boundaries),
    absorb its left half into the left block, its right half into the right block  (= the environments at
    the window's centre bond, which the sweep has just produced), insert a fresh window, repeat.

Every step adds 2 T sites, so  e = (E_n - E_{n-1}) / (2 T)  is the energy per site and the change of the centre
Schmidt spectrum is the convergence measure (the ||dC|| of MPSKit; C is diagonal after the SVD).  All arithmetic
is the finite sweep's: H_eff apply, Lanczos, per-sector SVD, environment transfer -- the HIP hot path.

Sector labels stay absolute (N = particles to the left of the bond): the left block's labels simply grow; the
right block's environment is re-keyed by the window's particle content 2 T P/Q when a window is inserted in front
of it (sorted block order, hence the device buffer, is invariant under that shift).  The reference's shifted charge
k = N Q - P sites (src:251) is this relabelling done once and for all.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import engine as _engine
from . import models, mps
from .planner import Bond, EnvLayout


def unit_cell(P: int, Q: int) -> int:
    """T = Q if P even else 2 Q (src:408-412)"""
    return Q if P % 2 == 0 else 2 * Q


def _shift_right_env(Rlay: EnvLayout, dN: int) -> EnvLayout:
    """the same environment, its bond relabelled N -> N + dN"""
    new = EnvLayout.build("R", Bond({(N + dN, j): n for (N, j), n in Rlay.bond.dims.items()}), Rlay.levels)
    old = list(Rlay.blocks.values())
    assert new.size == Rlay.size and list(new.blocks.values()) == old, "relabelling must keep the block order"
    return new


def _spectrum_distance(a: dict, b: dict, dN: int) -> float:
    """|| S_a - S_b || over sectors, b's labels shifted back by dN, shorter spectra zero padded"""
    keys = set(a) | {(N - dN, j) for (N, j) in b}
    d2 = 0.0
    for k in keys:
        x = np.asarray(a.get(k, []), dtype=float)
        y = np.asarray(b.get((k[0] + dN, k[1]), []), dtype=float)
        n = max(len(x), len(y))
        x = np.pad(x, (0, n - len(x)))
        y = np.pad(y, (0, n - len(y)))
        d2 += (k[1] + 1) * float(np.sum((x - y) ** 2))
    return float(np.sqrt(d2))


@dataclass
class IDMRGResult:
    energy_per_site: float
    delta: float                 # change of the centre Schmidt spectrum in the last step
    iterations: int
    unit_cell: int
    bond_dims: list              # TensorKit dims of the bonds inside the last window
    spectrum: dict               # centre Schmidt spectrum {sector: values} (absolute labels of the last window)
    history: list = field(default_factory=list)     # (energy per site, delta) per growth step
    engine: object = None        # the last window's engine (device-resident tensors of two unit cells)


def idmrg2(ops, sim, chi_full=None, cutoff=0.0, tol=1e-6, maxiter=100, sweeps_per_step=6, init_dimension=8,
           krylovdim=30, lanczos_tol=1e-10, seed=1234, verbosity=0, min_steps=3):
    """-> IDMRGResult.  `sim` is an OB_Sim / MB_Sim (filling P/Q); truncation by truncdim(chi_full) and/or
    truncbelow(cutoff) exactly as in the finite engine."""
    P, Q = int(sim.P), int(sim.Q)
    B = int(sim.bands)
    T = unit_cell(P, Q) * B                  # sites per unit cell
    W = 2 * T
    dNw = (W * P) // Q                       # particles in a window (integer: the cell length is a multiple of Q)
    assert (W * P) % Q == 0
    # translation-invariant MPO sites from the bulk of a long open chain
    ncells = 8 * unit_cell(P, Q)
    big = models.hamiltonian(sim, ncells)
    m0 = 3 * T
    mpo = [big[m0 + i] for i in range(W)]
    key = lambda w: (tuple(w.left), tuple(w.right), tuple(w.entries))
    assert key(big[m0]) == key(big[m0 + T]) == key(big[m0 + W]), "MPO is not periodic with the unit cell"
    # step 0: the window alone, open ends
    bL = {(0, 0): 1}
    bR = {(dNw, 0): 1}
    Llay = EnvLayout.build("L", Bond(bL), mpo[0].left)
    Rlay = EnvLayout.build("R", Bond(bR), mpo[W - 1].right)
    Lbuf = ops.zeros_z(max(Llay.size, 1))
    Rbuf = ops.zeros_z(max(Rlay.size, 1))
    E_prev, spec_prev, e_site, delta = None, None, float("nan"), float("inf")
    history = []
    eng = None
    for it in range(maxiter):
        bonds, tensors = mps.random_window(W, Llay.bond.dims, Rlay.bond.dims, init_dimension, seed=seed + it)
        eng = _engine.DMRG2(ops, mpo, bonds, tensors, chi_full=chi_full, cutoff=cutoff, krylovdim=krylovdim,
                            lanczos_tol=lanczos_tol, left_env=(Llay, Lbuf), right_env=(Rlay, Rbuf))
        # the window starts from a random state: sweep until its energy has settled (at least twice, at most
        # sweeps_per_step times); a window that has not found its ground state poisons the blocks it is absorbed into
        E_sw = None
        for k in range(sweeps_per_step):
            E_new = eng.sweep()
            if k >= 1 and E_sw is not None and abs(E_new - E_sw) <= 1e-11 * max(abs(E_new), 1.0):
                break
            E_sw = E_new
        # the leftward pass visited the centre bond last: its eigenvalue is the energy of system_n
        centre = [s for s in eng.stats if s.bond == T][-1]
        E = centre.energy
        spec = eng.spectra[T]
        if E_prev is not None:
            e_site = (E - E_prev) / W
            delta = _spectrum_distance(spec_prev, spec, dNw // 2)
        history.append((e_site, delta))
        if verbosity:
            print(f"IDMRG2 step {it + 1}: sites {W * (it + 1)}  E/site = {e_site:.10f}  delta = {delta:.3e}  "
                  f"chi = {eng.bonds[T].dim_full}")
        E_prev, spec_prev = E, spec
        if it + 1 >= min_steps and delta < tol:
            break
        # absorb the halves: environments at the centre bond become the new boundaries
        Llay, Lbuf = eng.Llay[T], eng.Lbuf[T]
        Rlay, Rbuf = _shift_right_env(eng.Rlay[T], dNw), eng.Rbuf[T]
    return IDMRGResult(energy_per_site=e_site, delta=delta, iterations=len(history), unit_cell=T,
                       bond_dims=[b.dim_full for b in eng.bonds], spectrum=spec_prev, history=history, engine=eng)
