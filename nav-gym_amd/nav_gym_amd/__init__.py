"""nav_gym_amd: MI355X-native batched NavGym step() behind the reference's gym API.

Importing the package registers 'NavGym-v0' (nav_gym_env/__init__.py:4-40) and 'CrowdSim-v0'
(crowd_sim/__init__.py:3-6) and touches neither the GPU nor the shared library; those load on first use and fail loudly when missing.
"""
from .env import DEFAULT_KWARGS, NavGymEnv  # noqa: F401
from .registry import make, register, spaces  # noqa: F401

register(id='NavGym-v0', kwargs=DEFAULT_KWARGS, entry_point='nav_gym_amd.env:NavGymEnv')
register(id='CrowdSim-v0', entry_point='nav_gym_amd.crowd:CrowdSimEnv')
