"""dcreg_amd: MI355X-native point-to-plane ICP inner loop with DCReg's degeneracy analysis.

The product is dcreg_amd/lib/libdcreg_hip.so (C-ABI: include/dcreg.h); this package is its thin
Python binding for tests, bench.py and multi-GPU launch.
"""
from . import api  # noqa: F401
from .api import Context, DcregError, METHODS, default_config, default_lin_params  # noqa: F401

__all__ = ["api", "Context", "DcregError", "METHODS", "default_config", "default_lin_params"]
