"""hubbardtn_amd -- MI355X-native two-site DMRG sweep engine behind HubbardTN's host API.

Only the hot path of SURVEY.md section 8 lives here: the host side the reference keeps in its own language (model structs
and Hamiltonian builder `models`, initial states `mps`, the model-level API `api`, the iDMRG growth loop `idmrg`, the
result cache `storage`) and the binding of the C ABI (`abi`, `engine`, `device`) behind which the planner, the sweep
driver and the hand-written HIP kernels live (hubbardtn_amd/csrc -> libhubbardtn_hip.so).
"""
from .models import MB_Sim, OB_Sim, hamiltonian  # noqa: F401

__all__ = ["OB_Sim", "MB_Sim", "hamiltonian"]
