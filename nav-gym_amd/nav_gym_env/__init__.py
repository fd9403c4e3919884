"""Drop-in alias: `import nav_gym_env` registers NavGym-v0 exactly like the reference package
(nav_gym_env/__init__.py:4-40), backed by the MI355X-native implementation."""
from nav_gym_amd import DEFAULT_KWARGS, NavGymEnv, make, register, spaces  # noqa: F401
