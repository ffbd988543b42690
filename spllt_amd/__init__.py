"""spllt_amd -- MI355X-native supernodal Cholesky factorize engine behind SpLLT's API.

Product code lives in ``csrc/`` (HIP kernels, stream-DAG scheduler, C-ABI) and is
reached only through ``libspllt_hip.so``; ``api`` marshals numpy arrays to it and
``matgen`` builds the benchmark matrices.  Nothing here imports ``oracle/``.
"""
from . import api, matgen  # noqa: F401
from .api import Factorization, SplltError, csc_lower_1based, residual  # noqa: F401
