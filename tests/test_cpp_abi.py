"""The C ABI used from compiled code: tests/cpp/test_abi.cpp (a C++ client written against include/poulpy_hip.hpp, the way a
backend shim in the reference's own language would be) is built with g++, linked against libpoulpy_hip.so and the CPU oracle,
and must reproduce the oracle bit for bit.  CPU part: the client compiles and links (no GPU call); GPU part: it runs."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_abi.cpp")


def _build(tmp_path) -> str:
    import __graft_entry__ as g
    lib = os.path.join(ROOT, "poulpy_amd", "libpoulpy_hip.so")
    if not os.path.exists(lib):
        g.build()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    exe = os.path.join(str(tmp_path), "test_abi")
    cmd = ["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), SRC, "-o", exe,
           lib, os.path.join(ROOT, "oracle", "_build", "libpoulpy_oracle.so"),
           "-Wl,-rpath," + os.path.join(ROOT, "poulpy_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle", "_build"),
           "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return exe


def test_cpp_client_builds_and_links(tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_cpp_client_runs_bit_exact(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "test_abi: OK" in out.stdout
