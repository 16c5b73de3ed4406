"""End-to-end GPU parity of the HIP sweep engine against exact results and the CPU oracle."""
import numpy as np
import pytest

from hubbardtn_amd import engine, models, mps
from oracle import dmrg_su2, mpo as ompo

pytestmark = pytest.mark.gpu

E_ED_L8_U4 = -4.235806999130        # SURVEY.md App. B (sparse ED, 4900 states)


def _oracle_run(L, t, u, chi, nsweeps, cap, seed=1234):
    bonds, tens = mps.random_mps(L, (L, 0), cap, seed)
    psi = dmrg_su2.MPS(L, (L, 0))
    psi.bonds = [dict(b) for b in bonds]
    psi.tensors = [dict(x) for x in tens]
    eng = dmrg_su2.DMRG2(psi, ompo.hubbard_mpo(L, t, u), chi_full=chi)
    out = []
    for _ in range(nsweeps):
        E, spec = eng.sweep()
        out.append((E, spec))
    return out


def _hip_run(ops, L, t, u, chi, nsweeps, cap, seed=1234):
    bonds, tens = mps.random_mps(L, (L, 0), cap, seed)
    sim = models.OB_Sim(t, u, 0.0, 1, 1, 2.0, cap)
    eng = engine.DMRG2(ops, models.hamiltonian(sim, L), bonds, tens, chi_full=chi)
    out = []
    for _ in range(nsweeps):
        E = eng.sweep()
        out.append((E, {k: dict(v) for k, v in eng.spectra.items()}))
    return out, eng


def test_L8_untruncated_matches_exact_diagonalisation(hip_ops):
    out, eng = _hip_run(hip_ops, 8, [1.0], [4.0], None, 2, 8)
    assert abs(out[-1][0] - E_ED_L8_U4) < 1e-10
    assert eng.bond_dims()[4] == 256           # full centre bond: no truncation anywhere


@pytest.mark.parametrize("L,t,u,chi", [(8, [1.0], [4.0], 64), (12, [1.0, 0.1], [8.0, 0.5], 48)])
def test_truncated_sweeps_match_oracle(hip_ops, L, t, u, chi):
    """energies and truncated singular spectra within 1e-8 relative (north_star tolerance)"""
    ref = _oracle_run(L, t, u, chi, 2, 6)
    out, eng = _hip_run(hip_ops, L, t, u, chi, 2, 6)
    for (Er, sr), (Eg, sg) in zip(ref, out):
        assert abs(Eg - Er) <= 1e-8 * abs(Er)
    sr, sg = ref[-1][1], out[-1][1]
    for bond in sr:
        assert set(sr[bond]) == set(sg[bond])
        for c in sr[bond]:
            a, b = np.asarray(sr[bond][c]), np.asarray(sg[bond][c])
            assert a.shape == b.shape
            assert np.abs(a - b).max() <= 1e-8 * a.max()
