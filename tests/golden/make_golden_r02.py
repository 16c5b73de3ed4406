"""Generates tests/golden/golden_r02.json: ORACLE trajectories at BASELINE.json's real chain length.

Round-2 fixtures: the numpy oracle (oracle/dmrg_su2.py) run in the build container on
  * L=64, U/t=4, half filling, chi schedule 32x2, 64x2, 128x2                       ("L64_U4_chi128")
  * L=64, U/t=4, half filling, chi schedule 32x2, 64x2, 128x1, 256x1, 512x1         ("L64_U4_chi512" = configs[1])
  * polyacetylene parameter set (examples/polyacetylene.jl:29-33), 16 cells = 32 sites, chi 32x2, 64x2, 128x1
from the deterministic start mps.random_mps(seed 1234, cap 4).  Recorded: energy after every sweep, the Schmidt
spectra of a few bonds after the last sweep, TensorKit dims of all bonds.  The `-m gpu` tests replay the same
schedule on the HIP engine and compare at 1e-8 relative.

Nothing here reads /root/reference.  One BLAS thread (the oracle's small per-sector GEMMs run 4x slower when
OpenBLAS spreads them over all cores).  Run:  python tests/golden/make_golden_r02.py [name ...]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hubbardtn_amd import models, mps                  # noqa: E402
from oracle import dmrg_su2, mpo as ompo               # noqa: E402

POLY = dict(t=[[0.000, 3.803, -0.548, 0.000], [3.803, 0.000, 2.977, -0.501]],
            u=[[10.317, 6.264, 0.000, 0.000], [6.264, 10.317, 6.162, 0.000]],
            J=[[0.000, 0.123, 0.000, 0.000], [0.123, 0.000, 0.113, 0.000]])

RUNS = {
    "L64_U4_chi128": dict(model="one_band", L=64, t=[1.0], u=[4.0], schedule=[(32, 2), (64, 2), (128, 2)],
                          bonds=[8, 16, 24, 32, 40, 48, 56]),
    "L64_U4_chi512": dict(model="one_band", L=64, t=[1.0], u=[4.0],
                          schedule=[(32, 2), (64, 2), (128, 1), (256, 1), (512, 1)], bonds=[16, 32, 48]),
    "poly32_chi128": dict(model="polyacetylene", L=32, schedule=[(32, 2), (64, 2), (128, 1)], bonds=[8, 16, 24]),
}


def product_mpo_as_oracle(sites):
    """the product's MPOSite list in the oracle's dict form (same content; the oracle has no multi-band builder)"""
    return [{"left": list(s.left), "right": list(s.right), "entries": list(s.entries)} for s in sites]


def run(name, cfg):
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
    energies, times, nmv = [], [], []
    spec = None
    for chi, nsw in cfg["schedule"]:
        eng.chi_full = chi
        for _ in range(nsw):
            t0 = time.perf_counter()
            n0 = len(eng.stats)
            E, spec = eng.sweep()
            energies.append(float(E))
            times.append(time.perf_counter() - t0)
            nmv.append(int(sum(s["nmv"] for s in eng.stats[n0:])))
            print(f"{name}: chi={chi} E={E:.12f} {times[-1]:.1f}s nmv={nmv[-1]}", flush=True)
    return {"model": cfg["model"], "L": L, "t": cfg.get("t"), "u": cfg.get("u"), "schedule": cfg["schedule"],
            "cap": 4, "seed": 1234, "lanczos_tol": 1e-12, "krylovdim": 30, "maxrestart": 3,
            "energies": energies, "oracle_seconds_per_sweep": times, "matvecs_per_sweep": nmv,
            "spectra_last_sweep": {str(b): {f"{c[0]},{c[1]}": [float(x) for x in v] for c, v in spec[b].items()}
                                   for b in cfg["bonds"]},
            "bond_dims": [dmrg_su2.bond_dim_full(b) for b in psi.bonds]}


def oracle_vs_reference_constants():
    """the ORACLE (not the product) against the infinite-chain energies the reference's own tests
    pin (tests/golden/reference_constants.json <- test/OB.jl:15-54), with the reference's own truncation
    truncbelow(10^-svalue) (src:1007-1010): finite chains of L1 < L2 sites at filling P/Q, energy density
    (E(L2) - E(L1)) / (L2 - L1) (the boundary terms cancel to O(1/L^2))."""
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_constants.json")))
    out = []
    for rec in ref["OB_parameters"] + ref["OB_filling"]:
        Es = []
        for L in (24, 32):
            N = L * rec["P"] // rec["Q"]
            psi = dmrg_su2.random_mps(L, (N, 0), 6, seed=5)
            eng = dmrg_su2.DMRG2(psi, ompo.hubbard_mpo(L, rec["t"], rec["u"]), chi_full=16, lanczos_tol=1e-6)
            # state preparation (fixed D, loose eigensolver): two-site DMRG from a random state spreads charge slowly at
            # strong coupling / low density, so many cheap sweeps come first
            for chi, nsw in ((8, 16), (16, 12), (32, 10)):
                eng.chi_full = chi
                for _ in range(nsw):
                    E, _ = eng.sweep()
            eng.chi_full, eng.cutoff, eng.lanczos_tol = None, 10.0 ** -rec["svalue"], 1e-10
            for _ in range(3):                                        # the reference's scheme: truncbelow(10^-svalue)
                E, _ = eng.sweep()
            Es.append(float(E))
            print(f"oracle_vs_reference U={rec['u']} P/Q={rec['P']}/{rec['Q']} L={L}: E={E:.10f} "
                  f"chi={max(dmrg_su2.bond_dim_full(b) for b in psi.bonds)}", flush=True)
        e = (Es[1] - Es[0]) / 8
        out.append({"t": rec["t"], "u": rec["u"], "P": rec["P"], "Q": rec["Q"], "svalue": rec["svalue"], "L": [24, 32],
                    "E_oracle": Es, "e_density_oracle": e, "E_per_site_reference": rec["E_per_site"], "atol": rec["atol"]})
        print(f"   e = {e:.8f}  reference {rec['E_per_site']:.8f}  diff {e - rec['E_per_site']:+.2e}", flush=True)
    return out


def main():
    from threadpoolctl import threadpool_limits
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_r02.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    names = sys.argv[1:] or list(RUNS)
    with threadpool_limits(limits=1):
        for name in names:
            out[name] = oracle_vs_reference_constants() if name == "oracle_vs_reference_constants" else run(name, RUNS[name])
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
