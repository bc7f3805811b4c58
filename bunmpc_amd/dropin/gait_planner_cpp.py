"""PYTHONPATH shim: `import gait_planner_cpp` resolves to bunmpc_amd.gait_planner_cpp."""
from bunmpc_amd.gait_planner_cpp import *  # noqa: F401,F403
