"""IDMRG2 on the GPU against every infinite-chain energy the reference's own tests pin for the SU(2) x U(1) models
(test/OB.jl:15-54, test/MB.jl:59-65; fixture tests/golden/reference_constants.json) and against Bethe ansatz."""
import json
import os

import numpy as np
import pytest

from hubbardtn_amd import api, idmrg, models

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_constants.json")))


@pytest.mark.parametrize("rec", GOLD["OB_parameters"] + GOLD["OB_filling"], ids=lambda r: f"U{r['u'][0]:g}_P{r['P']}Q{r['Q']}")
def test_reference_infinite_chain_constants(rec):
    """the reference's call sequence (hf.OB_Sim -> produce_groundstate -> expectation_value) with the reference's
    truncation truncbelow(10^-svalue).  Its tests allow 1e-2; the shared truncation rule reproduces its numbers to
    a few 1e-4 (the remainder is its VUMPS / GradientGrassmann polish at that bond dimension, src:1025-1027)"""
    model = api.OB_Sim(rec["t"], rec["u"], 0.0, rec["P"], rec["Q"], rec["svalue"])
    d = api.produce_groundstate(model, tol=1e-5, maxiter=60)
    psi, H = d["groundstate"], d["ham"]
    E = float(np.sum(np.real(api.expectation_value(psi, H)))) / len(H)          # test/OB.jl:28-29
    assert len(H) == idmrg.unit_cell(rec["P"], rec["Q"])
    assert abs(E - rec["E_per_site"]) < rec["atol"]
    assert abs(E - rec["E_per_site"]) < 1e-3          # (measured: 3e-5 .. 7e-4; the crude cut makes the value path dependent)
    assert len(api.dim_state(psi)) == len(H)


@pytest.mark.parametrize("rec", GOLD["OB_parameters"], ids=lambda r: f"U{r['u'][0]:g}")
def test_energy_density_converges_to_bethe_ansatz(rec, hip_ops):
    """fixed-D truncation (truncdim) instead of the crude Schmidt cut: the exact Lieb-Wu energy density"""
    r = idmrg.idmrg2(hip_ops, models.OB_Sim(rec["t"], rec["u"]), chi_full=120, tol=1e-4, maxiter=25)
    assert abs(r.energy_per_site - rec["bethe"]) < 4e-4
    assert r.energy_per_site > rec["bethe"] - 1e-6                              # variational in the limit
    assert max(r.bond_dims) <= 120


def test_two_band_reference_constant():
    """test/MB.jl:22-62: two decoupled bands, U = 3, half filling, E/site = -0.630375296 (atol 1e-1)"""
    rec = GOLD["MB_groundstate"]
    model = api.MB_Sim(np.array(rec["t"]), np.array(rec["u"]), np.array(rec["J"]), rec["P"], rec["Q"], rec["svalue"], rec["bond_dim"])
    d = api.produce_groundstate(model, tol=1e-4, maxiter=40)
    H = d["ham"]
    E = float(np.sum(np.real(api.expectation_value(d["groundstate"], H)))) / len(H)
    assert len(H) == 4
    assert abs(E - rec["E_per_site"]) < rec["atol"]


def test_spinful_reference_constants():
    """test/Spin.jl:14-47: the fZ2 x U(1) x U(1) mode (`spin=true`): one band U = 8 -> -0.32637, two decoupled bands
    U = 3 -> -0.63093 (atol 1e-1), through the reference's call sequence; plus the density consistency check of
    test/Spin.jl:76-85"""
    model1 = api.OB_Sim([1.0], [8.0], 0.0, 1, 1, 2.0, spin=True)
    d1 = api.produce_groundstate(model1, tol=1e-4, maxiter=40)
    H1 = d1["ham"]
    E1 = float(np.sum(np.real(api.expectation_value(d1["groundstate"], H1)))) / len(H1)
    assert abs(E1 - (-0.32637)) < 1e-1 and abs(E1 - (-0.32637)) < 2e-3
    t = np.array([[0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    u = np.array([[3.0, 0.0, 0.0, 0.0], [0.0, 3.0, 0.0, 0.0]])
    model2 = api.MB_Sim(t, u, np.zeros((2, 2)), 1, 1, 2.0, 20, code="Spin", spin=True)
    d2 = api.produce_groundstate(model2, tol=1e-4, maxiter=40)
    H2 = d2["ham"]
    E2 = float(np.sum(np.real(api.expectation_value(d2["groundstate"], H2)))) / len(H2)
    # (two chains snaked onto one, multiplets resolved into their Sz components: truncbelow(1e-2) cuts more weight than in
    # the SU(2) mode, -0.611 here against the reference's -0.631 and the SU(2) mode's -0.6304; randomly started windows --
    # the default of this mode, see idmrg.idmrg2)
    assert len(H2) == 4 and abs(E2 - (-0.63093)) < 1e-1 and abs(E2 - (-0.63093)) < 3e-2
    for d in (d1, d2):
        n = api.density_state(d["groundstate"])
        up, dn = api.density_spin(d["groundstate"])
        assert abs(n.sum() - (up + dn).sum()) < 1e-8 and abs(n.sum() / len(d["ham"]) - 1.0) < 5e-3


def test_chemical_potential_models_on_the_gpu():
    """fZ2 x SU(2) sectors without U(1) (OBC_Sim2 / MBC_Sim, src:176-238, 341-382): one band at mu = U/2 through the
    reference's call sequence -- `produce_groundstate` takes the route of the one-site unit cell (src:1012-1022: the
    Schmidt cut expressed for the doubled cell, idmrg.schmidt_cut_scale) -- meets test/OBC.jl:20's constant INSIDE its own
    atol 1e-3; the two-band model of test/MBC.jl:22-59 meets its constant at the reference's tolerance"""
    sim = api.OBC_Sim2([1.0], [1.0], 0.5, 2.0)
    d = api.produce_groundstate(sim, tol=1e-4, maxiter=40)
    H, psi = d["ham"], d["groundstate"]
    n = api.density_state(psi)
    E0 = float(np.sum(api.expectation_value(psi, H))) / len(H) + 0.5 * float(n.mean())
    assert np.abs(n - 1.0).max() < 1e-5
    assert abs(E0 - (-1.03541433)) < 1e-3, E0                                    # test/OBC.jl:20, 30 (7e-4; cf. test_nou1_cpu.py)
    t = np.array([[0.5, 0.0, 1.0, 0.0], [0.0, 0.5, 0.0, 1.0]])
    u = np.array([[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0]])
    simb = api.MBC_Sim(t, u, np.zeros((2, 2)), 2.0, 20, code="MBC")
    db = api.produce_groundstate(simb, tol=1e-4, maxiter=40)
    nb = api.density_state(db["groundstate"])
    Eb = (float(np.sum(api.expectation_value(db["groundstate"], db["ham"]))) + 0.5 * float(nb.sum())) / len(db["ham"])
    assert abs(Eb - (-1.01631556)) < 1e-1 and Eb > -1.0404
