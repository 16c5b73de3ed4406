"""the reference's call sequence (examples/One_band.jl:20-46) through hubbardtn_amd.api on the GPU"""
import numpy as np
import pytest

from hubbardtn_amd import api
from oracle import ed

pytestmark = pytest.mark.gpu


def test_one_band_example_sequence_matches_exact_diagonalisation():
    # parameters of examples/One_band.jl:20-27 (s = 2.5, t = [1.0, 0.1], u = [8.0], half filling) on L = 10 sites
    s, P, Q, bond_dim = 2.5, 1, 1, 20
    t, u, mu = [1.0, 0.1], [8.0], 0.0
    model = api.OB_Sim(t, u, mu, P, Q, s, bond_dim, spin=False)
    dictionary = api.produce_groundstate(model, L=10, tol=1e-8)
    psi, H = dictionary["groundstate"], dictionary["ham"]
    E0 = api.expectation_value(psi, H)
    E = float(np.sum(np.real(E0))) / len(H)                      # examples/One_band.jl:42-43
    Eref, _ = ed.SectorED(10, 5, 5, t, u).ground_state()
    assert abs(E - Eref / 10) < 1e-2                             # the reference's own test tolerance (test/OB.jl:12)
    assert abs(E - Eref / 10) < 5e-4                             # Schmidt cut 10^-2.5 bias, E/site
    assert E >= Eref / 10 - 1e-12                                # variational
    dims = api.dim_state(psi)
    assert len(dims) == 10 and max(dims) >= 16 and dictionary["delta"] < 1e-8


def test_fixed_chi_truncdim_scheme():
    model = api.OB_Sim([1.0], [4.0], 0.0, 1, 1, 2.0, 8)
    d = api.compute_groundstate(model, L=8, chi=64, tol=1e-10, maxiter=8)
    E = float(np.sum(api.expectation_value(d["groundstate"], d["ham"])))
    assert abs(E - (-4.235806999130)) < 5e-7                     # chi = 64 truncates the L = 8 chain at ~1e-7
    assert max(api.dim_state(d["groundstate"])) <= 64
