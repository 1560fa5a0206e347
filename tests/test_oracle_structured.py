"""CPU half of the structured-input check (tests/test_gpu_structured.py): the oracle against the EXACT integer product (tools/structured.py) on
coherent inputs at configs[1] (N = 2^12, 4 limbs, base2k 17: still exact, margin 0.125) and one bit above (alternating signs: the oracle - cpu-ref's
arithmetic - leaves the exact product), plus the exact product itself against the schoolbook big-int product of oracle/exact.py at a small size."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _case(n, size, k, name):
    import structured as st
    from oracle.ref import RefModule
    from poulpy_amd.layouts import MatZnx, VecZnx
    ref = RefModule(n)
    mat, a, res = MatZnx(n, size, 2, 2, size), VecZnx(n, 2, size), VecZnx(n, 2, size)
    mat.data[...] = st.fill(mat.data.shape[:-1], name, n, k)
    a.data[...] = st.fill(a.data.shape[:-1], name, n, k)
    pm = ref.vmp_pmat_alloc(size, 2, 2, size)
    ref.vmp_prepare(pm, mat)
    margin = ref.rounding_margin_of(lambda: ref.glwe_external_product(res, k, a, k, pm, 1, k))
    return bool(np.array_equal(st.exact_external_product(a.data, mat.data, k), res.data)), margin


def test_exact_product_against_schoolbook():
    import structured as st
    from oracle.exact import negacyclic_mul
    rng = np.random.default_rng(5)
    n, rows = 64, 6
    a = rng.integers(-(1 << 18), 1 << 18, (rows, n), dtype=np.int64)
    b = rng.integers(-(1 << 18), 1 << 18, (rows, n), dtype=np.int64)
    want = sum(negacyclic_mul(a[r], b[r]) for r in range(rows))
    assert np.array_equal(st.exact_negacyclic_sum(a, b).astype(object), want)


@pytest.mark.parametrize("name", ["min", "alt", "tone:1", "tone:N/4", "tone:N/2-1", "delta"])
def test_oracle_is_exact_on_structured_inputs_at_config1(name):
    ok, margin = _case(4096, 4, 17, name)
    assert ok and margin < 0.25, (name, margin)
    if name != "delta":
        assert margin > 0.01, "a coherent input must sit far closer to a wrong limb than uniform digits do (6e-4 at this shape)"


def test_oracle_leaves_the_exact_product_one_bit_above_config1():
    ok, margin = _case(4096, 4, 18, "alt")
    assert not ok and margin >= 0.25, (ok, margin)
