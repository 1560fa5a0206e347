"""Shared helpers for the parity tests (tests/ only)."""
from __future__ import annotations

import numpy as np

from poulpy_amd.layouts import MatZnx, ScalarZnx, VecZnx, VecZnxBig, VecZnxDft


def rng_for(*key) -> np.random.Generator:
    return np.random.default_rng(abs(hash(tuple(key))) % (1 << 63))


def seeded(seed: int) -> np.random.Generator:
    return np.random.default_rng(seed)


# The suite's bound on the rounding margin - max |x - round(x)| over every value an inverse transform rounds (0.5 = a wrong limb) - of the
# calls it runs: "fp tolerance must be documented" (poulpy-hal/docs/backend_safety_contract.md:25-27), SURVEY.md 7 "exactness margin".
MARGIN_MAX = 0.05


def probed_margin(hip, run) -> float:
    """`run()` once more with the module's rounding-margin probe on (pz_module_set_margin_probe): the same dispatch with the probing
    instantiations of the rounding kernels.  Returns the margin; the caller compares the outputs of this run as well."""
    return hip.rounding_margin_of(run)


def normalize_all(mod, res_big: VecZnxBig, base2k: int, res_size: int | None = None, res_base2k: int | None = None) -> VecZnx:
    res = VecZnx(res_big.n, res_big.cols, res_size or res_big.size)
    res.data[...] = 0x5A5A5A5A  # garbage: every limb must be written (test_suite/vmp.rs:81-82)
    for j in range(res_big.cols):
        mod.vec_znx_big_normalize(res, res_base2k or base2k, 0, j, res_big, base2k, j)
    return res


def garbage_dft(n, cols, size, rng) -> VecZnxDft:
    d = VecZnxDft(n, cols, size)
    d.data[...] = rng.standard_normal(d.data.shape) * 1e6
    return d
