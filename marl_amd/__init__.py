"""marl_amd: MI355X-native batched JammerEnv step (HIP kernels behind a C ABI).

Public surface:
    BatchedJammerEnv   E environments in lockstep on one GPU (device tensors in/out)
    JammerEnv          single-env facade with the reference's Gymnasium-style API
    EnvConfig          constructor arguments of the reference env
"""
from .env import BatchedJammerEnv, EnvConfig, JammerEnv  # noqa: F401
from ._lib import MarlError, load as load_library  # noqa: F401

__all__ = ["BatchedJammerEnv", "JammerEnv", "EnvConfig", "MarlError", "load_library"]
