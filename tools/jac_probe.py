import sys, time, json, numpy as np
sys.path.insert(0,'.')
import torch
from hubbardtn_amd import engine, models, mps, planner as pl
from hubbardtn_amd.device import HipOps
ops=HipOps(0)
L=64
mpo=models.hamiltonian(models.OB_Sim([1.0],[4.0]),L)
bonds,tens=mps.random_mps(L,(L,0),4)
eng=engine.DMRG2(ops,mpo,bonds,tens,chi_full=16,lanczos_tol=1e-6)
for chi,n in [(16,8),(32,4),(64,4),(128,2),(256,2)]:
    eng.chi_full=chi
    for _ in range(n): eng.sweep()
eng.chi_full=512; eng.lanczos_tol=1e-10
eng.sweep()
eng.stats.clear(); eng.profile=True
eng.sweep()
st=eng.stats
print("jacobi sweeps: mean %.1f max %d"%(np.mean([s.jacobi_sweeps for s in st]), max(s.jacobi_sweeps for s in st)))
c=[s for s in st if s.bond==32][0]
print(c)
tl=pl.ThetaLayout.build(eng.bonds[31],eng.bonds[33])
print("center blocks rows x cols:",sorted([(tl.mats[c][1],tl.mats[c][2]) for c in tl.mids],reverse=True)[:12])
print("svd per bond ms: mean %.2f  lanczos %.2f plan %.2f env %.2f"%tuple(1e3*np.mean([getattr(s,k) for s in st]) for k in ("t_svd","t_lanczos","t_plan","t_env")))
