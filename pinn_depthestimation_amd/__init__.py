"""pinn_depthestimation_amd — MI355X-native engine for the PINN depth-inversion hot path
(tanh-MLP forward + first-order PDE residual + parameter gradient inside Adam / L-BFGS).
See DESIGN.md.  The compute lives in libpinn_hip.so (include/pinn_hip.h)."""
from ._lib import PinnError, build, load  # noqa: F401
from .engine import Engine, NetDesc, ResidualSpec  # noqa: F401

__version__ = "0.1.0"
