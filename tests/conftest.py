import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# A red run must say as much as possible as early as possible (pytest -x stops at the first failure): the cheap,
# diagnostic tests run first -- kernels vs numpy / LAPACK, then the debug-switch runs, the engine vs oracle / ED, the API,
# IDMRG2 -- and the long full-size trajectories last.  Files not named here keep their place in front.
_ORDER = ["test_kernels_gpu", "test_debug_gpu", "test_engine_gpu", "test_dist_gpu", "test_api_gpu", "test_idmrg_gpu", "test_fullsize_gpu"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else -1
    items.sort(key=rank)          # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def hip_ops():
    from hubbardtn_amd.device import HipOps
    return HipOps(0)


@pytest.fixture(autouse=True)
def _isolated_project_dir(tmp_path, monkeypatch):
    """produce_groundstate caches results under <project>/data/sims (src/HubbardFunctions.jl:1134-1166): every test gets
    its own project directory, so no test loads another one's cache entry and nothing is written into the repository"""
    monkeypatch.setenv("HTN_PROJECT_DIR", str(tmp_path))
