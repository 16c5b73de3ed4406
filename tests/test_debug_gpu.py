"""The two debug switches of the HIP backend (hubbardtn_amd/csrc/htn_common.h), each exercised by one run.

  HTN_DEBUG_POISON=1       every block the device pool hands out is filled with 0xFF bytes (NaN) first: a kernel that
                           consumes memory nobody wrote turns the energies into NaN (or trips the Lanczos NaN guard)
                           instead of depending on what the box's memory held before.
  HTN_DEBUG_EVENT_WAITS=1  the Lanczos driver waits for a completed HIP event per step instead of polling the
                           host-mapped step record; the record must still validate.

(A third child runs the fallback SVD path of the large blocks, HTN_SVD_PAIRS=1, against the default ring kernel; a fourth,
HTN_RING_NO_XCD=1, the dense workgroup placement in which the large-block kernels hand panels over through memory instead
of one XCD's L2 -- the same arithmetic, so again bit for bit.)
Both are read once per process, so each case runs `python tests/test_debug_gpu.py` as ONE child process (one extra GPU
process at a time) and must reproduce the in-process run of the same schedule BIT FOR BIT: energies after every sweep
and the centre Schmidt spectrum.  The schedule covers the one-workgroup SVD, the forced large-block SVD path
(svd_split), a model with a Z stage (range-2 hopping) and the multi-band polyacetylene parameter set.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POLY = dict(t=[[0.000, 3.803, -0.548, 0.000], [3.803, 0.000, 2.977, -0.501]],
            u=[[10.317, 6.264, 0.000, 0.000], [6.264, 10.317, 6.162, 0.000]],
            J=[[0.000, 0.123, 0.000, 0.000], [0.123, 0.000, 0.113, 0.000]])


def run_schedule():
    """the workload: three short runs through the C ABI on cuda:0 -> plain lists (JSON-able, bit-exact via hex floats)"""
    from hubbardtn_amd import engine, models, mps
    from hubbardtn_amd.device import HipOps
    ops = HipOps(0)
    out = {}
    cases = {
        "one_band_nnn": (models.hamiltonian(models.OB_Sim([1.0, 0.1], [4.0]), 16), 16, [(32, 1), (96, 2)], 0),
        "one_band_large_svd": (models.hamiltonian(models.OB_Sim([1.0], [4.0]), 16), 16, [(64, 1), (200, 2)], 600),
        "polyacetylene": (models.hamiltonian(models.MB_Sim(np.array(POLY["t"]), np.array(POLY["u"]), np.array(POLY["J"]),
                                                           1, 1, 2.5, 20), 8), 16, [(32, 1), (96, 2)], 0),
    }
    for name, (H, L, schedule, split) in cases.items():
        bonds, tens = mps.random_mps(L, (L, 0), 4, seed=11)
        eng = engine.DMRG2(ops, H, bonds, tens, chi_full=schedule[0][0], lanczos_tol=1e-11)
        eng.svd_split = split
        Es = []
        for chi, nsw in schedule:
            eng.chi_full = chi
            for _ in range(nsw):
                Es.append(float(eng.sweep()))
        spec = eng.spectrum(L // 2)
        out[name] = {"E": [e.hex() for e in Es],
                     "spec": {f"{c[0]},{c[1]}": [float(x).hex() for x in v] for c, v in sorted(spec.items())},
                     "jacobi_sweeps_max": max(s.jacobi_sweeps for s in eng.stats)}
    return out


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    print("RESULT " + json.dumps(run_schedule()))
    sys.exit(0)


pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def baseline(hip_ops):
    return run_schedule()


def _child(env_name):
    env = dict(os.environ)
    env[env_name] = "1"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def _same(a, b):
    assert a.keys() == b.keys()
    for name in a:
        Ea = [float.fromhex(x) for x in a[name]["E"]]
        Eb = [float.fromhex(x) for x in b[name]["E"]]
        assert all(np.isfinite(Eb)), (name, Eb)
        assert Ea == Eb, (name, Ea, Eb)
        assert a[name]["spec"] == b[name]["spec"], name


def test_schedule_takes_the_paths_it_is_meant_to(baseline):
    assert baseline["one_band_large_svd"]["jacobi_sweeps_max"] >= 2
    for name, rec in baseline.items():
        assert all(np.isfinite(float.fromhex(x)) for x in rec["E"]), name


def test_poisoned_pool_changes_nothing(baseline):
    """no kernel consumes memory nobody wrote: with every pool block starting as NaN the run is bit-identical"""
    _same(baseline, _child("HTN_DEBUG_POISON"))


def test_pair_visit_svd_path_agrees_with_the_ring_path(baseline):
    """HTN_SVD_PAIRS=1 sends the large blocks through the multi-launch pair-visit block Jacobi (the fallback for blocks the
    ring kernel cannot take) instead of the one-launch ring kernel: a different rotation order, the same SVD -- energies to
    1e-10 relative, Schmidt values to 1e-9 of the sector's largest"""
    other = _child("HTN_SVD_PAIRS")
    assert other.keys() == baseline.keys()
    for name in baseline:
        Ea = [float.fromhex(x) for x in baseline[name]["E"]]
        Eb = [float.fromhex(x) for x in other[name]["E"]]
        assert np.allclose(Ea, Eb, rtol=1e-10, atol=0.0), (name, Ea, Eb)
        assert baseline[name]["spec"].keys() == other[name]["spec"].keys()
        for c, va in baseline[name]["spec"].items():
            a = np.array([float.fromhex(x) for x in va])
            b = np.array([float.fromhex(x) for x in other[name]["spec"][c]])
            assert a.shape == b.shape and np.abs(a - b).max() <= 1e-9 * a.max(), (name, c)


def test_dense_placement_changes_nothing(baseline):
    """HTN_RING_NO_XCD=1: the workgroups of a large block are NOT steered to one XCD, so k_jacobi_ring / k_qr_large use the
    placement-independent hand-off (write-through stores) instead of the one through the shared L2 -- which bytes travel how
    must not change a single bit of the result"""
    _same(baseline, _child("HTN_RING_NO_XCD"))


def test_event_waits_change_nothing(baseline):
    """the polled step record and a completed per-step event deliver the same tridiagonal coefficients"""
    _same(baseline, _child("HTN_DEBUG_EVENT_WAITS"))
