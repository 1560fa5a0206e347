"""poulpy_amd — MI355X (gfx950) FFT64 backend for poulpy-hal.

The product is ``libpoulpy_hip.so`` (C ABI in ``include/poulpy_hip.h``, HIP kernels in
``poulpy_amd/csrc``).  This package is the thin host-side mirror used by the tests and
``bench.py``: numpy-backed layout containers (``layouts``) and a ``Module`` whose methods
carry the names and argument order of the reference's ``poulpy-hal`` api traits
(``hal``).  There is no CPU fallback: importing ``hal`` works anywhere (so that the
symbol-export test can run without a GPU) but constructing a ``Module`` without the
built library or without a HIP device raises.
"""
from .layouts import (MatZnx, ScalarZnx, SvpPPol, VecZnx, VecZnxBig, VecZnxDft, VmpPMat)  # noqa: F401
from .hal import Module, PoulpyHipError, load_library  # noqa: F401
