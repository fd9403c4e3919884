"""nav_gym_amd: MI355X-native batched NavGym step() behind the reference's gym API.

Importing the package registers 'NavGym-v0' (nav_gym_env/__init__.py:4-40) and touches neither the
GPU nor the shared library; those load on first use and fail loudly when missing.
"""
from .env import DEFAULT_KWARGS, NavGymEnv  # noqa: F401
from .registry import make, register, spaces  # noqa: F401

register(id='NavGym-v0', kwargs=DEFAULT_KWARGS, entry_point='nav_gym_amd.env:NavGymEnv')
