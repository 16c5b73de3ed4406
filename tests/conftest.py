import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_ops():
    from hubbardtn_amd.device import HipOps
    return HipOps(0)


@pytest.fixture(autouse=True)
def _isolated_project_dir(tmp_path, monkeypatch):
    """produce_groundstate caches results under <project>/data/sims (src/HubbardFunctions.jl:1134-1166): every test gets
    its own project directory, so no test loads another one's cache entry and nothing is written into the repository"""
    monkeypatch.setenv("HTN_PROJECT_DIR", str(tmp_path))
