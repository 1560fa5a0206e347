"""The tree collects from its root: a bare `pytest -m "not gpu"` in a clean checkout (the reference's CI runs `cargo test --workspace`
from the root, .github/workflows/ci.yml:46-80) must import nothing outside tests/ and must not need a device."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tracked_files():
    out = subprocess.run(["git", "ls-files", "-co", "--exclude-standard"], cwd=ROOT, capture_output=True, text=True)
    if out.returncode != 0:
        return None
    return [f for f in out.stdout.splitlines() if f and os.path.isfile(os.path.join(ROOT, f))]


@pytest.mark.skipif(shutil.which("git") is None or not os.path.isdir(os.path.join(ROOT, ".git")), reason="needs the git checkout")
def test_bare_pytest_collects_in_a_clean_checkout(tmp_path):
    files = _tracked_files()
    assert files, "git ls-files returned nothing"
    for f in files:
        if f.startswith("gpurun_out/"):
            continue
        dst = tmp_path / f
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy2(os.path.join(ROOT, f), dst)
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTEST") and not k.startswith("POULPY")}
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    run = subprocess.run([sys.executable, "-m", "pytest", "-m", "not gpu", "--collect-only", "-q", "-p", "no:cacheprovider"],
                         cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    tail = (run.stdout + run.stderr)[-3000:]
    assert run.returncode == 0, tail
    collected = [ln for ln in run.stdout.splitlines() if "::" in ln]
    assert len(collected) >= 200, tail
    outside = [ln for ln in collected if not ln.startswith("tests/")]
    assert not outside, outside[:5]


def test_no_test_named_modules_outside_tests():
    """pytest's default patterns (test_*.py, *_test.py) must match nothing under tools/ — scripts there touch the GPU at import."""
    bad = []
    for base in ("tools", "poulpy_amd", "oracle"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            bad += [os.path.join(dp, f) for f in fs if f.endswith(".py") and (f.startswith("test_") or f.endswith("_test.py"))]
    assert not bad, bad
