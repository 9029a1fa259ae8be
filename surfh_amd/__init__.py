"""surfh_amd: MI355X-native implementation of surfh's forward/adjoint MRS operator chain and
the regularised least-squares CG loop, behind the reference's operator API.

    from surfh_amd import instru
    from surfh_amd.models import spectroSigRLSCT
"""
from . import instru  # noqa: F401
from .linop import LinOp, dottest, dotgap  # noqa: F401

__version__ = "0.1.0"
