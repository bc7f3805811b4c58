"""PYTHONPATH shim: `import biconvex_mpc_cpp` resolves to bunmpc_amd.biconvex_mpc_cpp."""
from bunmpc_amd.biconvex_mpc_cpp import *  # noqa: F401,F403
