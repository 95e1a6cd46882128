"""Import shim: lets the reference's scripts keep `from physics import ...` (train.py:17-19)
while the implementation lives in pinn_depthestimation_amd.physics.  Put this directory
first on sys.path."""
from pinn_depthestimation_amd.physics import *  # noqa: F401,F403
from pinn_depthestimation_amd import physics as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
