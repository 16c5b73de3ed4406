"""hubbardtn_amd -- MI355X-native two-site DMRG sweep engine behind HubbardTN's host API.

Only the hot path of SURVEY.md section 8 lives here: Hamiltonian builder (models), planner,
device primitives (C ABI over hand-written HIP kernels) and the sweep engine.
"""
from .models import MB_Sim, OB_Sim, hamiltonian  # noqa: F401

__all__ = ["OB_Sim", "MB_Sim", "hamiltonian"]
