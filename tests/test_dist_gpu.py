"""World-size-2 run of the HIP engine's SHARDED paths on ONE GPU: two processes, each with its own context on cuda:0, the
reduction hook (`htn_ctx_set_exchange`) carried over gloo through host memory.  What a multi-GPU node would run with RCCL
(`htn_ctx_set_comm`) -- the H_eff apply with its output tiles dealt over the ranks and y zero-filled, one reduction per
matvec; the sector-sharded SVD, every rank running the pivoted QR and the ring block Jacobi only on the blocks it owns, two
reductions per bond -- is exercised here with the same library code; only the transport of the sums differs.  The ranks must
stay in lock step bit for bit and reproduce the unsharded run of the same schedule."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r'''
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["HTN_ROOT"]); sys.path.insert(0, os.path.join(os.environ["HTN_ROOT"], "tests"))
from hubbardtn_amd import engine, models, mps
from hubbardtn_amd.device import HipOps
world = int(os.environ["WORLD_SIZE"])
rank = int(os.environ["RANK"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
ops = HipOps(0)
calls = [0]
class DevView:                       # a device pointer as a torch tensor (CUDA array interface), 2 n float64
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (2 * n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
def allreduce(y_ptr, n):             # the library has ENQUEUED the producer of y on its own stream: device-wide syncs order this
    calls[0] += 1
    torch.cuda.synchronize()
    t = torch.as_tensor(DevView(y_ptr, n), device="cuda:0")
    h = t.cpu()
    dist.all_reduce(h)
    t.copy_(h)
    torch.cuda.synchronize()
if world > 1:
    ops.set_exchange(rank, world, allreduce)
L = 12
H = models.hamiltonian(models.OB_Sim([1.0, 0.1], [4.0]), L)      # range-2 hopping: the apply has a Z stage
bonds, tens = mps.random_mps(L, (L, 0), 6, 77)
eng = engine.DMRG2(ops, H, bonds, tens, chi_full=64, lanczos_tol=1e-11)
eng.svd_split = 400                  # blocks above 400 elements take the large-block path: pivoted QR + ring Jacobi on several CUs
Es = []
for chi, nsw in ((64, 1), (160, 2)):
    eng.chi_full = chi
    for _ in range(nsw):
        Es.append(float(eng.sweep()))
spec = {f"{c[0]},{c[1]}": [float(x).hex() for x in v] for c, v in sorted(eng.spectrum(L // 2).items())}
res = {"E": [e.hex() for e in Es], "spec": spec, "calls": calls[0], "jac": max(s.jacobi_sweeps for s in eng.stats),
       "expected_calls": sum(s.n_matvec for s in eng.stats) + 2 * len(eng.stats)}
if world > 1:
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        print("RESULT " + json.dumps(gathered))
    dist.destroy_process_group()
else:
    print("RESULT " + json.dumps([res]))
'''


def _run(tmp_path, world):
    script = tmp_path / f"worker_gpu_{world}.py"
    script.write_text(WORKER)
    env = dict(os.environ, HTN_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if world == 1:
        env.update(RANK="0", WORLD_SIZE="1")
        cmd = [sys.executable, str(script)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", "29563", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_sharded_apply_and_sharded_svd_two_ranks_on_one_gpu(tmp_path):
    one = _run(tmp_path, 1)[0]
    two = _run(tmp_path, 2)
    assert len(two) == 2
    assert two[0]["E"] == two[1]["E"] and two[0]["spec"] == two[1]["spec"]                # lock step, bit for bit
    assert two[0]["jac"] >= 2                                                              # the large-block path did run
    # one reduction per matvec plus two per bond update (blocks, singular values) on every rank
    assert two[0]["calls"] == two[1]["calls"] == two[0]["expected_calls"] and one["calls"] == 0
    Ea = np.array([float.fromhex(x) for x in one["E"]])
    Eb = np.array([float.fromhex(x) for x in two[0]["E"]])
    assert np.allclose(Ea, Eb, rtol=1e-10, atol=0.0), (Ea, Eb)
    assert one["spec"].keys() == two[0]["spec"].keys()
    for c, va in one["spec"].items():
        a = np.array([float.fromhex(x) for x in va])
        b = np.array([float.fromhex(x) for x in two[0]["spec"][c]])
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-9 * a.max(), c
