"""IDMRG2 (hubbardtn_amd/idmrg.py) over the C++ sweep engine on the CPU baseline backend (tests/cpu_ops.py): host logic
of the growing-window driver and one of the reference's own infinite-chain known answers (test/OB.jl:44-54), small
enough for the CPU suite."""
import json
import os

import numpy as np

from cpu_ops import CpuOps
from hubbardtn_amd import api, idmrg, models, mps
from ref_planner import Bond, EnvLayout

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_constants.json")))


def test_unit_cell_rule():
    # T = Q if P even else 2Q (src/HubbardFunctions.jl:408-412)
    assert [idmrg.unit_cell(P, Q) for P, Q in ((1, 1), (1, 2), (3, 2), (2, 3), (2, 1))] == [2, 4, 4, 3, 1]


def test_random_window_keeps_boundaries_and_is_right_canonical():
    bl = {(10, 0): 3, (10, 2): 2, (11, 1): 4, (9, 1): 1}
    br = {(14, 0): 3, (14, 2): 2, (15, 1): 4, (13, 1): 2}
    bonds, tens = mps.random_window(4, bl, br, 5, seed=3)
    assert bonds[0] == bl and bonds[4] == br
    for i in range(1, 4):          # rows of the (c ; (s, b)) matrix orthonormal: right-canonical in the tilde normalisation
        rows = {}
        for (c, s, b), blk in tens[i].items():
            rows.setdefault(c, []).append(blk)
            assert blk.shape == (bonds[i][c], bonds[i + 1][b])
        for c, blks in rows.items():
            M = np.concatenate(blks, axis=1)
            assert np.abs(M @ M.conj().T - np.eye(M.shape[0])).max() < 1e-12
    for (c, s, b), blk in tens[0].items():
        assert blk.shape == (bl[c], bonds[1][b])


def test_right_environment_relabelling_keeps_the_buffer_layout():
    mpo = models.hamiltonian(models.OB_Sim([1.0, 0.2], [4.0]), 16)
    # the right block's bond is relabelled N -> N + dN when a window is inserted in front of it: the environment's
    # block order (sorted sectors), hence its flat data, must not change
    lay = EnvLayout.build("R", Bond({(7, 1): 3, (8, 0): 2, (8, 2): 4, (9, 1): 5}), mpo[8].right)
    new = EnvLayout.build("R", Bond({(N + 4, j): n for (N, j), n in lay.bond.dims.items()}), mpo[8].right)
    assert new.size == lay.size and list(new.blocks.values()) == list(lay.blocks.values())
    assert idmrg._zero_env({(4, 0): 1}, mpo[8].right, "R").size == max(EnvLayout.build("R", Bond({(4, 0): 1}), mpo[8].right).size, 1)
    assert idmrg._spectrum_distance({(3, 1): [0.8, 0.1]}, {(5, 1): [0.8, 0.1]}, 2) == 0.0
    assert abs(idmrg._spectrum_distance({(3, 1): [0.8]}, {(5, 1): [0.8, 0.1]}, 2) - np.sqrt(2) * 0.1) < 1e-15


def test_idmrg2_reproduces_a_reference_test_constant_with_the_reference_truncation():
    """U = 5, half filling, svalue = 2: test/OB.jl:44-54 pins E/site = -0.48460447 (atol 1e-2) for IDMRG2 with
    truncbelow(1e-2) + variational polish; the same truncation scheme here lands 4e-5 above it (no polish)"""
    rec = GOLD["OB_filling"][1]
    sim = api.OB_Sim(rec["t"], rec["u"], 0.0, rec["P"], rec["Q"], rec["svalue"], 8)
    H = api.hamiltonian(sim)
    assert len(H) == 2
    psi = api.initialize_mps(H, sim.P, sim.bond_dim, ops=CpuOps())
    alg = api.IDMRG2(trscheme=api.truncbelow(10.0 ** -sim.svalue), tol=2e-4, maxiter=14, eigsolve_tol=1e-9, sweeps_per_step=3)
    psi, envs, delta = api.find_groundstate(psi, H, alg)
    e = api.expectation_value(psi, H)
    assert e.shape == (2,) and abs(e[0] - e[1]) == 0.0
    assert abs(e[0] - rec["E_per_site"]) < rec["atol"]            # the reference's own tolerance
    assert abs(e[0] - rec["E_per_site"]) < 5e-4                   # what the shared truncation rule actually gives
    assert delta < 1e-3 and max(api.dim_state(psi)) <= 20


def test_predicted_windows_reach_the_fixed_point_of_randomly_started_ones_in_fewer_sweeps():
    """McCulloch's prediction (idmrg._absorb): every window after the first starts from the previous window's halves.
    At a truncation fine enough for the fixed point to be well defined (truncbelow(1e-3)) the warm- and the cold-started
    growth agree on the energy density to 1e-5 and on the centre Schmidt spectrum; the predicted windows need about half
    the sweeps; and a predicted window is already within 1e-4 (relative) of its converged energy after its FIRST sweep."""
    from hubbardtn_amd import engine
    ops = CpuOps()
    sim = models.OB_Sim([1.0], [4.0], 0.0, 1, 1, 2.0, 50)
    first, orig = [], engine.DMRG2.sweep

    def spy(self):
        E = orig(self)
        self._n = getattr(self, "_n", 0) + 1
        if self._n == 1:
            first.append([E])
        else:
            first[-1].append(E)
        return E
    engine.DMRG2.sweep = spy
    try:
        warm = idmrg.idmrg2(ops, sim, cutoff=1e-3, tol=1e-4, maxiter=30, warm_start=True)
        per_window = [list(w) for w in first]
        cold = idmrg.idmrg2(ops, sim, cutoff=1e-3, tol=1e-4, maxiter=30, warm_start=False)
    finally:
        engine.DMRG2.sweep = orig
    assert warm.iterations == cold.iterations
    assert abs(warm.energy_per_site - cold.energy_per_site) < 1e-5
    assert abs(warm.energy_per_site - (-0.5737)) < 1e-3                 # Lieb-Wu: -0.573729
    assert warm.sweeps < 0.7 * cold.sweeps
    for w in per_window[3:]:                                            # (the first windows are far from translation invariant)
        assert abs(w[0] - w[-1]) < 1e-4 * abs(w[-1])
    d = idmrg._spectrum_distance(warm.spectrum, cold.spectrum, 0)
    assert d < 3e-3                                                     # (one multiplet next to the 1e-3 threshold)
