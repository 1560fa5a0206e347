"""Golden vectors (exact big-int arithmetic, tests/golden/make_golden.py) against the CPU oracle
(not gpu) and against the HIP path through the C ABI (gpu)."""
import os

import pytest

from tests.golden_runner import fixtures, run_fixture


@pytest.mark.parametrize("path", fixtures(), ids=lambda p: os.path.basename(p))
def test_oracle_matches_golden(path):
    from oracle.ref import RefModule
    run_fixture(path, RefModule)


@pytest.mark.gpu
@pytest.mark.parametrize("path", fixtures(), ids=lambda p: os.path.basename(p))
def test_hip_matches_golden(path):
    from poulpy_amd.hal import Module
    run_fixture(path, Module)
