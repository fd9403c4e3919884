"""Drop-in alias: `import crowd_sim` registers CrowdSim-v0 like the reference package (crowd_sim/__init__.py:3-6), backed
by nav_gym_amd.crowd.CrowdSimEnv (host-side reset with the reference's own draws, the step on the device)."""
from nav_gym_amd import make, register  # noqa: F401
from nav_gym_amd.crowd import CROWD_DEFAULTS, CROWD_INFO, CrowdSimEnv, CrowdSimStepper, crowd_reset_scenario  # noqa: F401
