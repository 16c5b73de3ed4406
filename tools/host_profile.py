"""cProfile of one steady-state sweep (host side) on the GPU box"""
import cProfile
import pstats
import sys
sys.path.insert(0, ".")
from hubbardtn_amd import engine, models, mps
from hubbardtn_amd.device import HipOps

ops = HipOps(0)
L = 64
mpo = models.hamiltonian(models.OB_Sim([1.0], [4.0]), L)
bonds, tens = mps.random_mps(L, (L, 0), 4)
eng = engine.DMRG2(ops, mpo, bonds, tens, chi_full=16, lanczos_tol=1e-6)
for chi, n in [(16, 8), (32, 4), (64, 4), (128, 2), (256, 2), (512, 2)]:
    eng.chi_full = chi
    for _ in range(n):
        eng.sweep()
eng.chi_full, eng.lanczos_tol = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 1e-10
eng.sweep()
eng.sweep()
pr = cProfile.Profile()
pr.enable()
eng.sweep()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
print("cache hits/misses", eng.cache_hits, eng.cache_misses)
