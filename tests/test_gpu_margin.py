"""The rounding margin as a first-class output, for every kernel family that rounds ("fp tolerance must be documented",
poulpy-hal/docs/backend_safety_contract.md:25-27; SURVEY.md 7 "exactness margin").

pz_module_set_margin_probe switches each rounding kernel of a call to its probing form IN THE SAME DISPATCH (compile-time PROBE
instantiations in their own translation units for every form of the fused tail, k_br_fused and k_inv_pass1 - launch_tail_probe.hip,
launch_br_probe.hip; a run-time wave-uniform test in k_small_inv and k_small_idft): the results must stay bit-identical
to the oracle's, and the margin max |x - round(x)| must be (a) non-zero - the probe really ran on this path, (b) far from 0.5 at the
reference's parameters, (c) growing with base2k the way the error model says (about x4 per bit)."""
import numpy as np
import pytest

from tests.helpers import MARGIN_MAX

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from oracle.ref import RefModule
    from poulpy_amd.hal import Module
    cache = {}

    def get(n):
        if n not in cache:
            cache[n] = (RefModule(n), Module(n))
        return cache[n]
    return get


class probing:
    """with probing(hip) as box: ...   -> box.margin after the block"""

    def __init__(self, hip):
        self.hip, self.margin = hip, None

    def __enter__(self):
        self.hip.sync()
        self.hip.set_margin_probe(True)
        return self

    def __exit__(self, *exc):
        try:
            self.hip.sync()
            self.margin = self.hip.get_margin()
        finally:
            self.hip.set_margin_probe(False)
        return False


GLWE_CASES = [
    # n, ks, size, base2k, auto, note
    (64, False, 3, 17, None, "tiny ring: per-op style plans"),
    (1024, False, 2, 17, None, "configs[0] shape on the small-ring pipeline (k_small_inv)"),
    (2048, True, 3, 15, None, "small-ring key switch (body operand in k_small_inv)"),
    (2048, True, 3, 15, (5, "automorphism"), "k_small_inv<.., AU>"),
    (4096, False, 4, 17, None, "configs[1]: two-kernel pipeline at N = 4096"),
    (8192, False, 4, 14, None, "three-kernel pipeline, plain tail"),
    (8192, True, 4, 14, None, "key switch: tail with the body operand"),
    (8192, True, 4, 14, (5, "automorphism"), "spectral automorphism: sign-only tail + operand tail"),
    (8192, True, 4, 14, (-1, "add"), "automorphism_add: gathered operand"),
    (65536, False, 8, 12, None, "metric shape"),
]


@pytest.mark.parametrize("n,ks,size,base2k,auto,note", GLWE_CASES, ids=[f"n{c[0]}-{'ks' if c[1] else 'ep'}-{c[4][1] if c[4] else 'plain'}" for c in GLWE_CASES])
def test_glwe_ops_probe_same_bits_and_margin(mods, n, ks, size, base2k, auto, note):
    from tests.test_gpu_parity import _run_glwe_op
    ref, hip = mods(n)
    args = (hip, ref, ks, n, 1, 1, size, base2k, size, base2k, size, 1, size, base2k)
    got0, want = _run_glwe_op(*args, batch=5, seed=n + size, auto=auto)
    assert np.array_equal(got0, want), note
    with probing(hip) as box:
        got1, _ = _run_glwe_op(*args, batch=5, seed=n + size, auto=auto)
    assert np.array_equal(got1, want), f"probing instantiation changed the result ({note})"
    assert 0.0 < box.margin < MARGIN_MAX, (note, box.margin)


@pytest.mark.parametrize("n,rank,blk,dnum,bsz,rsz,k,note", [
    (512, 3, 3, 1, 2, 1, 18, "`ref` shape of the reference bench (blind_rotation.rs:39-57): one-kernel rotation, base2k 18"),
    (1024, 1, 7, 3, 3, 3, 13, "`cbt` shape: one-kernel rotation, two ciphertexts per workgroup"),
    (1024, 1, 7, 3, 3, 3, 19, "base2k 19 (BASELINE.md section 3 lists 18 - 19 for configs[3])"),
    (1024, 2, 7, 3, 4, 4, 13, "rank 2: block step + small-ring tail"),
    (2048, 1, 7, 3, 3, 3, 18, "block step + k_small_inv<NOPROD, FWD>, base2k 18"),
    (4096, 1, 7, 3, 3, 3, 13, "block step on the pipeline, k_inv_tail with the accumulator operand"),
    (16384, 1, 7, 3, 3, 3, 18, "configs[3] N = 2^14, base2k 18"),
])
def test_blind_rotation_probe_same_bits_and_margin(mods, n, rank, blk, dnum, bsz, rsz, k, note):
    from tests.test_gpu_parity import _run_blind_rotation
    ref, hip = mods(n)
    n_lwe = 2 * blk + 1
    got0, want = _run_blind_rotation(hip, ref, n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch=3, seed=n + rank + k)
    assert np.array_equal(got0, want), note
    with probing(hip) as box:
        got1, _ = _run_blind_rotation(hip, ref, n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch=3, seed=n + rank + k)
    assert np.array_equal(got1, want), f"probing changed the result ({note})"
    assert 0.0 < box.margin < MARGIN_MAX, (note, box.margin)


@pytest.mark.parametrize("n,size,mode", [(8192, 4, "apply"), (8192, 4, "square"), (65536, 16, "apply")])
def test_tensoring_probe_same_bits_and_margin(mods, n, size, mode):
    """the tensoring tails (k_inv_tail<.., NZ = 1 / 2, PROBE>): diagonal and pairwise launches"""
    from tests.test_gpu_cnv import _run_tensor
    ref, hip = mods(n)
    k = 12
    args = (hip, ref, n, 1, size, size, size, k, k, size * k - 20, mode)
    got0, want = _run_tensor(*args, batch=2, seed=n + size)
    assert np.array_equal(got0, want)
    with probing(hip) as box:
        got1, _ = _run_tensor(*args, batch=2, seed=n + size)
    assert np.array_equal(got1, want)
    assert 0.0 < box.margin < MARGIN_MAX, box.margin


def test_margin_grows_with_base2k_as_modelled(mods):
    """error ~ N * rows * 2^(2 base2k) * 2^-53: two more bits of base2k cost about 16x (between 6x and 40x measured on a maximum)."""
    from tests.test_gpu_parity import _run_glwe_op
    n = 4096
    ref, hip = mods(n)
    margins = {}
    for k in (15, 17, 19):
        with probing(hip) as box:
            got, want = _run_glwe_op(hip, ref, False, n, 1, 1, 4, k, 4, k, 4, 1, 4, k, batch=4, seed=k)
        assert np.array_equal(got, want)
        margins[k] = box.margin
    assert 6 < margins[17] / margins[15] < 40 and 6 < margins[19] / margins[17] < 40, margins
    assert margins[19] < MARGIN_MAX, margins


def test_probe_off_leaves_no_margin(mods):
    """the word is reset when the probe is switched, and nothing writes it while the probe is off"""
    from tests.test_gpu_parity import _run_glwe_op
    n = 1024
    ref, hip = mods(n)
    hip.set_margin_probe(True)
    hip.set_margin_probe(False)
    _run_glwe_op(hip, ref, False, n, 1, 1, 2, 17, 2, 17, 2, 1, 2, 17, batch=2, seed=5)
    assert hip.get_margin() == 0.0
