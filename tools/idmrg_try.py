"""IDMRG2 energy densities against the reference test constants (test/OB.jl) with the reference truncation, on the GPU"""
import sys, time
sys.path.insert(0, ".")
from hubbardtn_amd import models, idmrg
from hubbardtn_amd.device import HipOps
ops = HipOps(0)
cases = [(0.0, 1, 1, -1.2696767, -1.2732395447), (1.0, 1, 1, -1.037173, -1.0403686534), (2.0, 1, 1, -0.84163698, -0.8443743411),
         (5.0, 1, 2, -0.73920032, None), (5.0, 1, 1, -0.48460447, None), (5.0, 3, 2, 1.76073968, None)]
for U, P, Q, ref, bethe in cases:
    t0 = time.time()
    r = idmrg.idmrg2(ops, models.OB_Sim([1.0], [U], 0.0, P, Q), chi_full=None, cutoff=1e-2, tol=1e-5, maxiter=60, sweeps_per_step=4)
    print(f"U={U} P/Q={P}/{Q}: e = {r.energy_per_site:.8f}  ref const {ref} diff {r.energy_per_site - ref:+.2e}  delta {r.delta:.2e} steps {r.iterations} "
          f"dims {r.bond_dims} {time.time() - t0:.1f}s", flush=True)
