"""Generates tests/golden/golden_r03.json: PER-BOND trace of the first sweep of the oracle trajectories of golden_r02.json.

Round-3 fixture: a trajectory fixture that holds one energy per sweep can only say "sweep 0 is off".
Here the numpy oracle (oracle/dmrg_su2.py) replays the first sweep of each distinct start of golden_r02.json (the two L=64
one-band trajectories share theirs) and records, for every one of its 2L-3 bond updates in order: bond, direction,
Lanczos eigenvalue, matvec count, residual, discarded weight, kept multiplets and TensorKit dim.  The `-m gpu` trajectory
test compares the HIP engine's per-bond statistics with it and names the first (bond, stage) that departs.

Consistency: the energy after the traced sweep must equal energies[0] of the golden_r02 record bit for bit (same oracle,
same start), which the script asserts.  Nothing here reads /root/reference.
Run:  python tests/golden/make_golden_r03.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from hubbardtn_amd import models, mps                  # noqa: E402
from oracle import dmrg_su2, mpo as ompo               # noqa: E402
from make_golden_r02 import POLY, RUNS, product_mpo_as_oracle      # noqa: E402

TRACES = {"L64_U4": "L64_U4_chi128", "poly32": "poly32_chi128"}       # trace name -> golden_r02 record with the same start


def trace(r02name):
    cfg = RUNS[r02name]
    L = cfg["L"]
    bonds, tens = mps.random_mps(L, (L, 0), 4, seed=1234)
    psi = dmrg_su2.MPS(L, (L, 0))
    psi.bonds = [dict(b) for b in bonds]
    psi.tensors = [dict(x) for x in tens]
    if cfg["model"] == "one_band":
        mpo = ompo.hubbard_mpo(L, cfg["t"], cfg["u"])
    else:
        sim = models.MB_Sim(np.array(POLY["t"]), np.array(POLY["u"]), np.array(POLY["J"]), 1, 1, 2.5, 20)
        mpo = product_mpo_as_oracle(models.hamiltonian(sim, L // 2))
    eng = dmrg_su2.DMRG2(psi, mpo, chi_full=cfg["schedule"][0][0], lanczos_tol=1e-12)
    E, _ = eng.sweep()
    return float(E), [dict(bond=int(s["bond"]), dir=int(s["dir"]), E=float(s["E"]), nmv=int(s["nmv"]), res=float(s["res"]),
                           trunc=float(s["trunc"]), mult=int(s["mult"]), chi_full=int(s["chi_full"])) for s in eng.stats]


def main():
    from threadpoolctl import threadpool_limits
    r02 = json.load(open(os.path.join(HERE, "golden_r02.json")))
    out = {}
    with threadpool_limits(limits=1):
        for name, r02name in TRACES.items():
            E, bonds = trace(r02name)
            # (the oracle itself reproduces a trajectory to ~1e-15 relative only: OpenBLAS kernels differ by summation order)
            assert abs(E - r02[r02name]["energies"][0]) <= 1e-12 * abs(E), (name, E, r02[r02name]["energies"][0])
            out[name] = {"same_start_as": [k for k, v in RUNS.items() if v["L"] == RUNS[r02name]["L"] and v["model"] == RUNS[r02name]["model"]],
                         "chi_full": RUNS[r02name]["schedule"][0][0], "krylovdim": 30, "energy_after_sweep": E, "bonds": bonds}
            print(name, E, len(bonds), flush=True)
    with open(os.path.join(HERE, "golden_r03.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote golden_r03.json")


if __name__ == "__main__":
    main()
