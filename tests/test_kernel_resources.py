"""The register / scratch table as a test: read the code-object metadata of the BUILT libpoulpy_hip.so
(tools/kres_so.py: llvm-objdump --offloading + llvm-readelf --notes) and fail when a kernel carries scratch (private segment) outside the
allow-list below.  Scratch is HBM traffic per lane (profiles/r04_tensor_traffic.json: 88 B per lane were 3.4 GB per launch); a new spill must be a
decision, not an accident of the last commit.  CPU test: nothing here needs a GPU."""
import fnmatch
import os
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tools"))

# pattern (fnmatch on the demangled name without arguments) -> (max scratch bytes per lane, why it is tolerated)
SCRATCH_ALLOWED = {
    "k_br_fused<4, 2, 512, 1, 6, 3, *>": (32, "one-kernel rotation, N = 512 rank 3 with two ciphertexts per workgroup, at the 256-register cap: 3 - 6 "
                                               "registers spilled since the key values arrive through buffer loads (round 5, 79aa705); stored once per block"),
    "k_mid128r<4, 16, false, 16, false, 6, true, false>": (24, "digit-selected (dsize > 1) middle kernel: 5 registers of the per-term tables, stored once in "
                                                         "the prologue and re-read once per tile (NOTEBOOK 12.7)"),
    "k_mid128<4, 16, *, false, false, true, 0, 0>": (104, "digit-selected k_mid128: only reached with POULPY_DBG_MID_R=0 (the cross-check path of k_mid128r<..,DS>)"),
    "k_inv_tail<8, 8, 16, true, false, false, 2, *>": (12, "pairwise tensoring tail (mode-5 prefetch of the diagonal digits) at the 168-register cap of its 3 waves per SIMD"),
    "k_inv_tail<16, 16, 16, true, false, false, 2, *>": (16, "pairwise tensoring tail at N = 2^16: as above, 2 registers"),
}


@pytest.fixture(scope="module")
def table():
    import kres_so
    lib = os.environ.get("POULPY_HIP_LIB") or os.path.join(ROOT, "poulpy_amd", "libpoulpy_hip.so")
    if not os.path.exists(lib):
        pytest.skip("libpoulpy_hip.so not built")
    if not os.path.exists(os.path.join(kres_so.LLVM, "llvm-readelf")):
        pytest.skip("llvm-readelf not available")
    rows = kres_so.kernel_table(lib)
    assert len(rows) > 100, "the library's code objects carry hundreds of kernels; the metadata parse found %d" % len(rows)
    return rows


def allowed(name):
    for pat, (cap, _why) in SCRATCH_ALLOWED.items():
        if fnmatch.fnmatchcase(name, pat):
            return cap
    return 0


def test_no_scratch_outside_the_allow_list(table):
    bad = [(r["name"], r["scratch"]) for r in table if r["scratch"] > allowed(r["name"])]
    assert not bad, "kernels with scratch beyond the allow-list (tests/test_kernel_resources.py): %s" % bad


def test_allow_list_has_no_dead_entries(table):
    # an entry that no longer matches a spilling kernel is a fixed spill: delete it, so the cap cannot be re-used silently
    for pat in SCRATCH_ALLOWED:
        hit = [r for r in table if fnmatch.fnmatchcase(r["name"], pat) and r["scratch"] > 0]
        assert hit, "allow-list entry without a spilling kernel: %s" % pat


def test_headline_kernels_are_spill_free(table):
    by = {r["name"]: r for r in table}
    for name in ("k_mid128r<4, 16, false, 16, false, 6, false, false>", "k_mid128r<4, 16, false, 16, false, 6, false, true>", "k_fwd_pass1<16, 16, 16, true, false>",
                 "k_inv_tail<16, 16, 16, true, false, false, 0, false, false, false>", "k_mid_cnv3<16, 16, false, 12>", "k_mid_cnv3<16, 16, true, 12>", "k_mid_cnv3<16, 16, false, 0>", "k_mid_cnv3<16, 16, true, 0>",
                 "k_inv_tail<16, 16, 16, true, false, false, 3, false, false, false>"):
        assert name in by, "kernel not found in the library: %s" % name
        assert by[name]["scratch"] == 0 and by[name]["vgpr_spill"] == 0, (name, by[name])


def test_docs_quote_the_table(table):
    # DESIGN.md's sentence about spills must name the count this table shows
    n = sum(1 for r in table if r["scratch"] > 0)
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert ("%d instantiations carry scratch" % n) in text, "DESIGN.md must state: '%d instantiations carry scratch' (tests/test_kernel_resources.py)" % n
