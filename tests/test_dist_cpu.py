"""world_size-2 (gloo, CPU) rehearsal of the sector-parallel apply: each rank executes its share of the
output tiles on the numpy emulator, y is summed with torch.distributed.all_reduce, and both ranks must
reproduce the single-process energies bit for bit among themselves and to 1e-10 against the golden run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["HTN_ROOT"]); sys.path.insert(0, os.path.join(os.environ["HTN_ROOT"], "tests"))
from emul import NumpyOps
import ref_engine as engine
from hubbardtn_amd import models, mps
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
def allreduce(y):
    t = torch.from_numpy(y.view(np.float64))
    dist.all_reduce(t)
L, t, u, chi = 8, [1.0], [4.0], 64
bonds, tens = mps.random_mps(L, (L, 0), 6, 1234)
eng = engine.DMRG2(NumpyOps(), models.hamiltonian(models.OB_Sim(t, u), L), bonds, tens, chi_full=chi,
                   shard=(rank, world, allreduce))
Es = [eng.sweep() for _ in range(2)]
gathered = [None] * world
dist.all_gather_object(gathered, Es)
if rank == 0:
    print(json.dumps({"E": gathered}))
dist.destroy_process_group()
'''


def test_sharded_apply_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, HTN_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    E = json.loads(line)["E"]
    assert E[0] == E[1]                                     # ranks stay in lockstep bit for bit
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_r01.json")))["oracle_runs"]["L8_U4_chi64"]
    for a, b in zip(E[0], gold["energies"]):
        assert abs(a - b) <= 1e-10 * abs(b)


WORKER_CXX = r'''
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["HTN_ROOT"]); sys.path.insert(0, os.path.join(os.environ["HTN_ROOT"], "tests"))
from cpu_ops import CpuOps
from hubbardtn_amd import engine, models, mps
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
calls = [0]
def allreduce(y):                       # y: numpy complex128 view of the library's buffer
    calls[0] += 1
    dist.all_reduce(torch.from_numpy(y.view(np.float64)))
ops = CpuOps()
ops.set_exchange(rank, world, allreduce)
L, t, u, chi = 8, [1.0], [4.0], 64
bonds, tens = mps.random_mps(L, (L, 0), 6, 1234)
eng = engine.DMRG2(ops, models.hamiltonian(models.OB_Sim(t, u), L), bonds, tens, chi_full=chi)
Es = [eng.sweep() for _ in range(2)]
spec = {f"{c[0]},{c[1]}": v.tolist() for c, v in eng.spectrum(4).items()}
tiles = int(eng.plan_apply_dump(3, 1)[0].shape[0])
gathered = [None] * world
dist.all_gather_object(gathered, (Es, spec, calls[0], sum(s.n_matvec for s in eng.stats) + 2 * len(eng.stats)))
if rank == 0:
    print(json.dumps({"out": gathered, "tiles": tiles}))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_cxx_engine_sharded_apply_two_ranks_gloo(tmp_path, world):
    """the product's C++ sweep driver with the sector-parallel apply (tiles dealt over ranks inside the library, y
    zero-filled, ONE reduction per matvec through the exchange hook) and the sector-sharded SVD (every rank decomposes
    the blocks it owns, two reductions per bond hand everybody the complete result) at world size 2 and 3 (an odd deal of
    tiles and blocks) over gloo: ranks stay in lock step bit for bit and reproduce the unsharded golden energies and spectra"""
    script = tmp_path / "worker_cxx.py"
    script.write_text(WORKER_CXX)
    env = dict(os.environ, HTN_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(29541 + world), str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)["out"]
    assert len(res) == world
    (E0, S0, c0, mv0) = res[0]
    for (E1, S1, c1, mv1) in res[1:]:
        assert E0 == E1 and S0 == S1                          # lock step, bit for bit
        # exactly one reduction per matvec, plus two per bond update for the sector-sharded SVD (blocks, singular values)
        assert c0 == c1 == mv0 == mv1 and c0 > 0
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_r01.json")))["oracle_runs"]["L8_U4_chi64"]
    for a, b in zip(E0, gold["energies"]):
        assert abs(a - b) <= 1e-10 * abs(b)
    for c, v in gold["spectra_last_sweep"]["4"].items():
        assert max(abs(x - y) for x, y in zip(S0[c], v)) < 1e-9


def test_bench_launch_path_with_two_ranks_on_the_cpu_backend():
    """`bench.py --gpus 2` outside torchrun starts its own two ranks (python -m torch.distributed.run) and relays rank 0's JSON
    line: the launch path the driver would use on a multi-GPU node, rehearsed on the CPU baseline library over gloo with a tiny
    chain.  The line must say n_gpus = 2, carry the sweep-level roofline and come from exactly one rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--backend", "cpu", "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--L", "8", "--chi", "32", "--grow", "16x1", "--master-port", "29617"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 1 and rec["config"]["backend"] == "cpu"
    assert rec["config"]["parallelism"].endswith("x2") and rec["scaling"] == "strong"
    assert rec["roofline"]["sweep"]["bound_s"] > 0 and abs(rec["energy_per_site"]) < 10.0
