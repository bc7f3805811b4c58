"""PYTHONPATH shim: `import inverse_kinematics_cpp` resolves to bunmpc_amd.inverse_kinematics_cpp."""
from bunmpc_amd.inverse_kinematics_cpp import *  # noqa: F401,F403
