"""Generates tests/golden/golden_r01.json: known answers independent of any tensor-network code
(exact diagonalisation, free fermions) plus oracle outputs for truncated runs.

Nothing here reads /root/reference: the reference (Julia + MPSKit/TensorKit, not runnable in this
image) ships no fixture files; its only pinned numbers are the infinite-chain energies of test/*.jl
(atol 1e-2), recorded below for documentation.  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hubbardtn_amd import mps                       # noqa: E402  (deterministic initial state shared by oracle and HIP runs)
from oracle import dmrg_su2, ed, mpo as ompo        # noqa: E402


def spectra_to_json(spec):
    return {str(b): {f"{c[0]},{c[1]}": [float(x) for x in v] for c, v in s.items()} for b, s in spec.items()}


def oracle_run(L, t, u, chi, nsweeps, cap, seed):
    bonds, tens = mps.random_mps(L, (L, 0), cap, seed)
    psi = dmrg_su2.MPS(L, (L, 0))
    psi.bonds = [dict(b) for b in bonds]
    psi.tensors = [dict(x) for x in tens]
    eng = dmrg_su2.DMRG2(psi, ompo.hubbard_mpo(L, t, u), chi_full=chi)
    energies = []
    for _ in range(nsweeps):
        E, spec = eng.sweep()
        energies.append(float(E))
    return energies, spectra_to_json(spec), [dmrg_su2.bond_dim_full(b) for b in psi.bonds]


def main():
    out = {"reference_test_constants_infinite_chain_atol_1e-2": {
        "test/OB.jl:21 U=0,1,2": [-1.2696767, -1.037173, -0.84163698],
        "test/OB.jl:44 U=5 fillings 1/2,1,3/2": [-0.73920032, -0.48460447, 1.76073968]}}
    # exact diagonalisation (independent of tensor networks)
    edv = {}
    for (L, t, u) in [(4, [1.0], [4.0]), (6, [1.0, 0.1], [8.0, 0.5]), (8, [1.0], [4.0]), (8, [1.0, 0.1], [8.0]), (10, [1.0], [4.0])]:
        e = ed.SectorED(L, L // 2, L // 2, t, u)
        E, psi = e.ground_state()
        rec = {"L": L, "t": t, "u": u, "E0": E}
        if L == 8 and t == [1.0]:
            ms = ed.multiplet_spectrum(e.schmidt_by_sector(psi, 4))
            rec["schmidt_centre"] = {f"{k[0]},{k[1]}": [float(x) for x in v] for k, v in ms.items()}
        edv[f"L{L}_t{t}_u{u}"] = rec
    out["exact_diagonalisation"] = edv
    out["free_fermions_obc"] = {str(L): ed.free_fermion_energy(L, L // 2, L // 2) for L in (8, 64, 128)}
    # oracle truncated runs (deterministic start: mps.random_mps seed / cap recorded)
    runs = {}
    for name, (L, t, u, chi, nsw, cap, seed) in {
            "L8_U4_chi64": (8, [1.0], [4.0], 64, 2, 6, 1234),
            "L12_t2_chi48": (12, [1.0, 0.1], [8.0, 0.5], 48, 2, 6, 1234),
            "L8_U4_chi48_seed7": (8, [1.0], [4.0], 48, 1, 6, 7)}.items():
        E, spec, dims = oracle_run(L, t, u, chi, nsw, cap, seed)
        runs[name] = {"L": L, "t": t, "u": u, "chi": chi, "sweeps": nsw, "cap": cap, "seed": seed,
                      "energies": E, "spectra_last_sweep": spec, "bond_dims": dims}
    out["oracle_runs"] = runs
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_r01.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
