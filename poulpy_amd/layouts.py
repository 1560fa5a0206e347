"""Host-side containers mirroring poulpy-hal/src/layouts (shape metadata + a byte buffer).

Memory layout is the reference's (poulpy-hal/src/layouts/znx_base.rs:52-82): limb-major,
column-minor, i.e. a C-contiguous array of shape (size, cols, n).  ``VecZnxDft`` /
``SvpPPol`` / ``VmpPMat`` hold backend-private f64 bytes ("device order") of the
same byte size as the reference's (module.rs:51-65).
"""
from __future__ import annotations

import numpy as np


class _Znx:
    dtype = np.int64

    def __init__(self, n: int, cols: int, size: int, data: np.ndarray | None = None):
        self.n, self.cols, self.size = int(n), int(cols), int(size)
        self.max_size = self.size
        if data is None:
            data = np.zeros((self.size, self.cols, self.n), dtype=self.dtype)
        assert data.dtype == self.dtype and data.flags["C_CONTIGUOUS"]
        assert data.size == self.n * self.cols * self.size
        self.data = data.reshape(self.size, self.cols, self.n)

    @classmethod
    def alloc(cls, n, cols, size):
        return cls(n, cols, size)

    def at(self, col: int, limb: int) -> np.ndarray:
        assert col < self.cols and limb < self.size
        return self.data[limb, col]

    def view(self, size: int):
        """Same buffer with a smaller current size (``set_size``, vec_znx.rs)."""
        assert size <= self.max_size
        out = type(self)(self.n, self.cols, size, np.ascontiguousarray(self.data.reshape(-1)[: self.n * self.cols * size]))
        return out

    def copy(self):
        return type(self)(self.n, self.cols, self.size, self.data.copy())

    def fill_uniform(self, log_bound: int, rng: np.random.Generator):
        """Uniform in [-2^(log_bound-1), 2^(log_bound-1)) (vec_znx.rs:282-295)."""
        if log_bound >= 64:
            self.data[...] = rng.integers(np.iinfo(np.int64).min, np.iinfo(np.int64).max, self.data.shape, dtype=np.int64, endpoint=True)
        else:
            h = 1 << (log_bound - 1)
            self.data[...] = rng.integers(-h, h, self.data.shape, dtype=np.int64)
        return self


class VecZnx(_Znx):
    """poulpy-hal/src/layouts/vec_znx.rs:33-41"""


class VecZnxBig(_Znx):
    """poulpy-hal/src/layouts/vec_znx_big.rs:23-32 (ScalarBig = i64)"""


class ScalarZnx(_Znx):
    """poulpy-hal/src/layouts/scalar_znx.rs — one limb per column"""

    def __init__(self, n, cols, size=1, data=None):
        super().__init__(n, cols, 1, data)


class VecZnxDft(_Znx):
    """poulpy-hal/src/layouts/vec_znx_dft.rs:25-34 (ScalarPrep = f64, opaque device order)"""
    dtype = np.float64

    def into_big(self) -> VecZnxBig:
        """Re-type the same bytes (vec_znx_dft.rs:57-59)."""
        return VecZnxBig(self.n, self.cols, self.size, self.data.view(np.int64))


class SvpPPol(_Znx):
    """poulpy-hal/src/layouts/svp_ppol.rs:21-28"""
    dtype = np.float64

    def __init__(self, n, cols, size=1, data=None):
        super().__init__(n, cols, 1, data)


class CnvPVecL(_Znx):
    """poulpy-hal/src/layouts/convolution.rs — prepared left operand of the bivariate convolution (ScalarPrep = f64, opaque bytes:
    n * cols * size scalars, module.rs:66-69)."""
    dtype = np.float64


class CnvPVecR(_Znx):
    """poulpy-hal/src/layouts/convolution.rs — prepared right operand (same byte size)."""
    dtype = np.float64


class MatZnx:
    """poulpy-hal/src/layouts/mat_znx.rs:28-35,161-181: entry (row, col_in) is a VecZnx(cols_out, size)."""

    def __init__(self, n, rows, cols_in, cols_out, size, data=None):
        self.n, self.rows, self.cols_in, self.cols_out, self.size = map(int, (n, rows, cols_in, cols_out, size))
        shape = (self.rows, self.cols_in, self.size, self.cols_out, self.n)
        if data is None:
            data = np.zeros(shape, dtype=np.int64)
        assert data.dtype == np.int64 and data.flags["C_CONTIGUOUS"]
        self.data = data.reshape(shape)

    @classmethod
    def alloc(cls, n, rows, cols_in, cols_out, size):
        return cls(n, rows, cols_in, cols_out, size)

    def at(self, row, col_in) -> VecZnx:
        return VecZnx(self.n, self.cols_out, self.size, self.data[row, col_in])

    def fill_uniform(self, log_bound, rng):
        h = 1 << (log_bound - 1)
        self.data[...] = rng.integers(-h, h, self.data.shape, dtype=np.int64)
        return self


class VmpPMat:
    """poulpy-hal/src/layouts/vmp_pmat.rs:23-33 — opaque prepared matrix."""

    def __init__(self, n, rows, cols_in, cols_out, size, data=None):
        self.n, self.rows, self.cols_in, self.cols_out, self.size = map(int, (n, rows, cols_in, cols_out, size))
        cnt = self.n * self.rows * self.cols_in * self.cols_out * self.size
        if data is None:
            data = np.zeros(cnt, dtype=np.float64)
        assert data.dtype == np.float64 and data.size == cnt
        self.data = data.reshape(-1)

    @classmethod
    def alloc(cls, n, rows, cols_in, cols_out, size):
        return cls(n, rows, cols_in, cols_out, size)
