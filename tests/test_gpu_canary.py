"""Workspace guards: POULPY_DBG_CANARY=1 puts a 256-byte guard behind every workspace segment and
verifies it when each API call returns.  (1) the mechanism catches a deliberate one-byte overrun and is silent without one;
(2) the GPU parity suites run once more in that mode, in a subprocess (HIP graphs off there, every call synchronised)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from poulpy_amd.hal import Module
m = Module(4096, device=0)
m.lib.pz_debug_workspace_overrun.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
rc = m.lib.pz_debug_workspace_overrun(m.handle, 1 << 20, int(sys.argv[1]))
print("rc", rc, flush=True)
"""


def _probe(overrun, canary):
    env = dict(os.environ)
    env.pop("POULPY_DBG_CANARY", None)
    if canary:
        env["POULPY_DBG_CANARY"] = "1"
    return subprocess.run([sys.executable, "-c", PROBE % ROOT, str(overrun)], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)


@pytest.mark.gpu
def test_guard_catches_a_one_byte_overrun():
    ok = _probe(0, True)
    assert ok.returncode == 0 and "rc 0" in ok.stdout, ok.stderr[-1500:]
    bad = _probe(1, True)
    assert bad.returncode != 0, "a one-byte overrun went unnoticed"
    assert "WORKSPACE OVERRUN" in bad.stderr and "pz_debug_workspace_overrun" in bad.stderr
    off = _probe(1, False)                      # mode off: nothing armed, nothing checked
    assert off.returncode == 0 and "rc 0" in off.stdout


@pytest.mark.gpu
def test_parity_suites_with_workspace_guards():
    if os.environ.get("POULPY_DBG_CANARY") == "1":
        pytest.skip("already inside the guarded run")
    env = dict(os.environ, POULPY_DBG_CANARY="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                          "tests/test_gpu_parity.py", "tests/test_gpu_cnv.py", "tests/test_gpu_lwe.py", "tests/test_gpu_scale.py"],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=3000)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "WORKSPACE OVERRUN" not in out.stderr
    assert " passed" in out.stdout, tail
