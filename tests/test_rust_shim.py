"""The Rust crate rust/poulpy-hip-mi355x cannot be compiled in this image (no cargo / rustc).  These CPU tests pin what can be
pinned without a compiler:

* `src/ffi.rs` is exactly what tools/gen_rust_ffi.py generates from include/poulpy_hip.h, and — checked independently of the
  generator — every `extern "C"` declaration has the header's arity and parameter types;
* `src/hal_impl.rs` defines every required fn of `unsafe trait HalImpl` (poulpy-hal/src/oep/hal_impl.rs:25-755; the names are the
  fixture tests/golden/hal_impl_fns.txt, re-derived from the reference checkout when it is present), none elided, none
  `unimplemented!`, and only calls C functions the header declares;
* `src/core_impl.rs` covers every required fn of `unsafe trait CoreImpl` (fixture tests/golden/core_impl_fns.txt): three families
  written out, four through the reference's `impl_core_*_default_methods!` macros;
* `src/tests.rs` instantiates every cross-backend test of the reference's HAL suite for the families this backend implements.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "poulpy-hip-mi355x")
REF = "/root/reference"


def read(*p):
    return open(os.path.join(*p)).read()


def strip_rust_comments(s):
    s = re.sub(r"//[^\n]*", "", s)
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def fixture(name):
    return [l.strip() for l in read(ROOT, "tests", "golden", name).splitlines() if l.strip() and not l.startswith("#")]


def test_ffi_rs_is_generated_from_the_header():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


C2R = {"size_t": "usize", "int64_t": "i64", "uint64_t": "u64", "uint32_t": "u32", "int": "c_int", "double": "f64", "float": "f32",
       "char": "c_char", "void": "c_void"}


def header_functions():
    """{name: (ret, [normalised C parameter types])} — a small independent parser of the header."""
    text = re.sub(r"/\*.*?\*/", " ", read(ROOT, "include", "poulpy_hip.h"), flags=re.S)
    text = re.sub(r"^\s*#[^\n]*", " ", text, flags=re.M)   # preprocessor lines (include guards, PZ_ABI_VERSION)
    text = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    text = re.sub(r"(typedef\s+)?enum\s*\{.*?\}\s*\w*\s*;", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b([A-Za-z_][\w\s\*]*?)\b(pz_\w+)\s*\(([^()]*)\)\s*;", text):
        params = []
        raw = " ".join(m.group(3).split())
        if raw and raw != "void":
            for a in raw.split(","):
                toks = a.replace("*", " * ").split()
                if toks[-1] not in ("*",) and toks[-1] not in C2R and not toks[-1].startswith("pz_") and len(toks) > 1:
                    toks = toks[:-1]                      # drop the parameter name
                params.append(" ".join(toks))
        out[m.group(2)] = (" ".join(m.group(1).split()), params)
    return out


def c_to_rust(ctype):
    toks = ctype.replace("*", " * ").split()
    consts, base, stars = [], None, []
    i = 0
    const_next = False
    if toks[0] == "const":
        const_next, i = True, 1
    base = toks[i]
    i += 1
    if i < len(toks) and toks[i] == "const":
        const_next, i = True, i + 1
    r = C2R.get(base, base)
    while i < len(toks):
        assert toks[i] == "*"
        i += 1
        r = ("*const " if const_next else "*mut ") + r
        const_next = False
        if i < len(toks) and toks[i] == "const":
            const_next, i = True, i + 1
    return r


def test_every_extern_declaration_matches_the_header():
    hdr = header_functions()
    ffi = strip_rust_comments(read(CRATE, "src", "ffi.rs"))
    block = ffi[ffi.index('unsafe extern "C" {'):]
    decls = {}
    for m in re.finditer(r"pub fn (pz_\w+)\(([^)]*)\)(?:\s*->\s*([^;]+))?;", block):
        params = [p.split(":", 1)[1].strip() for p in m.group(2).split(",") if p.strip()]
        decls[m.group(1)] = ((m.group(3) or "").strip(), params)
    assert set(decls) == set(hdr), (sorted(set(hdr) - set(decls)), sorted(set(decls) - set(hdr)))
    assert len(decls) >= 125
    for name, (ret, params) in hdr.items():
        rret, rparams = decls[name]
        assert len(params) == len(rparams), name
        for cp, rp in zip(params, rparams):
            assert c_to_rust(cp) == rp, (name, cp, rp)
        assert (rret == "" and ret == "void") or c_to_rust(ret) == rret, (name, ret, rret)
    # repr(C) structs: field-for-field
    for sname, body in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", re.sub(r"/\*.*?\*/", " ", read(ROOT, "include", "poulpy_hip.h"), flags=re.S), flags=re.S)[::1]:
        pass
    text = re.sub(r"/\*.*?\*/", " ", read(ROOT, "include", "poulpy_hip.h"), flags=re.S)
    for body, sname in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        cfields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if decl:
                ty, rest = decl.split(" ", 1)
                cfields += [(n.strip(), C2R.get(ty, ty)) for n in rest.split(",")]
        rs = re.search(r"pub struct %s \{(.*?)\}" % sname, ffi, flags=re.S).group(1)
        rfields = [(a.strip(), b.strip()) for a, b in re.findall(r"pub (\w+): ([\w_]+),", rs)]
        assert rfields == cfields, sname


def rust_fn_names(src):
    return re.findall(r"\n\s*(?:pub\s+)?(?:unsafe\s+)?fn (\w+)", strip_rust_comments(src))


def test_hal_impl_defines_every_required_method_of_the_trait():
    want = fixture("hal_impl_fns.txt")
    assert len(want) == 105
    src = read(CRATE, "src", "hal_impl.rs")
    code = strip_rust_comments(src)
    body = code[code.index("unsafe impl HalImpl<FFT64Hip> for FFT64Hip {"):]
    have = re.findall(r"\n    fn (\w+)", body)
    assert sorted(have) == sorted(want), (sorted(set(want) - set(have)), sorted(set(have) - set(want)))
    assert len(have) == len(set(have))
    for bad in ("unimplemented!", "todo!", "unreachable!", "hal_impl_scratch!", "hal_impl_vec_znx!"):
        assert bad not in code, bad
    # every forwarded call exists in the header, and every ScalarPrep-touching family is forwarded (not a CPU default)
    hdr = header_functions()
    used = set(re.findall(r"ffi::(pz_\w+)\(", code))
    assert used <= set(hdr), used - set(hdr)
    for name in want:
        if name.startswith(("vec_znx_dft_", "vec_znx_idft_", "svp_", "vmp_", "cnv_")):
            m = re.search(r"\n    fn %s\b.*?\n    \}\n" % name, body, flags=re.S)
            assert m and "ffi::pz_" + name in m.group(0), name
    assert code.count("{") == code.count("}") and code.count("(") == code.count(")")


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
def test_fixtures_are_the_reference_traits():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_shim as g
    names = [m[0] for m in g.parse_trait(os.path.join(REF, "poulpy-hal", "src", "oep", "hal_impl.rs"))]
    assert names == fixture("hal_impl_fns.txt")
    core = [m[0] for m in g.parse_trait_named(os.path.join(REF, "poulpy-core", "src", "oep", "core_impl.rs"), "pub unsafe trait CoreImpl")]
    assert core == fixture("core_impl_fns.txt")
    # and the committed sources are what the generator writes today
    before = {f: read(CRATE, "src", f) for f in ("hal_impl.rs", "core_impl.rs")}
    g.main()
    g.core_main(REF)
    for f, txt in before.items():
        assert read(CRATE, "src", f) == txt, f + " is stale: run python tools/gen_rust_shim.py"


def test_core_impl_covers_the_trait():
    want = fixture("core_impl_fns.txt")
    code = strip_rust_comments(read(CRATE, "src", "core_impl.rs"))
    body = code[code.index("unsafe impl CoreImpl<FFT64Hip> for FFT64Hip {"):]
    written = re.findall(r"\n    fn (\w+)", body)
    assert len(written) == len(set(written)) == 35
    macros = re.findall(r"poulpy_core::(impl_core_\w+_default_methods)!\(FFT64Hip\);", body)
    assert sorted(macros) == ["impl_core_conversion_default_methods", "impl_core_decryption_default_methods",
                              "impl_core_encryption_default_methods", "impl_core_operations_default_methods"]
    assert set(written) <= set(want)
    fused = [n for n in written if re.search(r"\n    fn %s\b.*?ffi::pz_glwe_\w+_batched" % n, body[:body.index("\n    fn " + n) + 4000] if False else
                                              re.search(r"\n    fn %s\b.*?\n    \}\n" % n, body, flags=re.S).group(0), flags=re.S)]
    assert sorted(fused) == sorted(["glwe_external_product", "glwe_external_product_assign", "glwe_keyswitch", "glwe_keyswitch_assign",
                                    "glwe_automorphism", "glwe_automorphism_assign", "glwe_automorphism_add", "glwe_automorphism_add_assign",
                                    "glwe_automorphism_sub", "glwe_automorphism_sub_assign", "glwe_automorphism_sub_negate",
                                    "glwe_automorphism_sub_negate_assign",
                                    # matrix-level forms: one batched call over the (row, column) entries
                                    "gglwe_keyswitch", "gglwe_keyswitch_assign", "gglwe_external_product", "gglwe_external_product_assign",
                                    "ggsw_external_product", "ggsw_external_product_assign"])
    if os.path.isdir(REF):   # the four macro families + the 35 written-out fns are the whole trait
        n_macro = 0
        for fam in ("decryption", "conversion", "operations", "encryption"):
            s = read(REF, "poulpy-core", "src", "oep", fam + ".rs")
            n_macro += len(re.findall(r"\n        fn (\w+)", s[s.index("macro_rules!"):]))
        assert n_macro + len(written) == len(want)
    assert code.count("{") == code.count("}") and code.count("(") == code.count(")")


def test_lib_znx_and_tests_rs():
    lib = strip_rust_comments(read(CRATE, "src", "lib.rs"))
    for needle in ("impl Backend for FFT64Hip", "type ScalarPrep = f64", "type ScalarBig = i64", "type OwnedBuf = PinnedBuf",
                   "unsafe fn destroy", "mod hal_impl;", "mod znx;", "mod core_impl;", "mod tests;"):
        assert needle in lib, needle
    znx = strip_rust_comments(read(CRATE, "src", "znx.rs"))
    traits = re.findall(r"\n    (Znx\w+)::", znx)
    assert len(traits) == len(set(traits)) == 27          # the 27 Znx* impls of poulpy-cpu-ref/src/fft64/znx.rs
    if os.path.isdir(REF):
        ref = read(REF, "poulpy-cpu-ref", "src", "fft64", "znx.rs")
        assert sorted(traits) == sorted(re.findall(r"\nimpl (Znx\w+) for FFT64Ref", ref))
    t = strip_rust_comments(read(CRATE, "src", "tests.rs"))
    listed = set(re.findall(r"\b(test_\w+)\b", t))
    assert "cross_backend_test_suite!" in t and "backend_ref = poulpy_cpu_ref::FFT64Ref" in t and "backend_test = crate::FFT64Hip" in t
    if os.path.isdir(REF):
        for mod_ in ("vec_znx", "vec_znx_big", "vec_znx_dft", "svp", "vmp"):
            s = read(REF, "poulpy-hal", "src", "test_suite", mod_ + ".rs")
            cross = [m.group(1) for m in re.finditer(r"pub fn (test_\w+)(?:<[^>]*>)?\s*\(\s*params", s)]
            assert cross and set(cross) <= listed, (mod_, set(cross) - listed)
    for f in ("lib.rs", "znx.rs", "tests.rs"):
        c = strip_rust_comments(read(CRATE, "src", f))
        assert c.count("{") == c.count("}") and c.count("(") == c.count(")"), f


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
def test_cross_backend_suite_lists_every_test_the_avx_backend_lists():
    """module by module, `src/tests.rs` instantiates at least the tests poulpy-cpu-avx/src/fft64/tests.rs instantiates for FFT64Avx
    (Missing 4 read the vec_znx_dft list as shorter: it is the AVX list plus test_vec_znx_copy)."""
    avx = read(REF, "poulpy-cpu-avx", "src", "fft64", "tests.rs")
    ours = strip_rust_comments(read(CRATE, "src", "tests.rs"))
    mine = {m.group(1): set(re.findall(r"test_\w+", m.group(2))) for m in re.finditer(r"hal_suite!\((\w+):(.*?)\);", ours, re.S)}
    mine["sampling"] = set(re.findall(r"(test_\w+)\s*=>", ours))
    seen = 0
    for m in re.finditer(r"mod (\w+),.*?tests = \{(.*?)\n    \}", avx, re.S):
        names = set(re.findall(r"(test_\w+)\s*=>", m.group(2)))
        assert names and names <= mine.get(m.group(1), set()), (m.group(1), names - mine.get(m.group(1), set()))
        seen += 1
    assert seen >= 6   # vec_znx, svp, vec_znx_big, vec_znx_dft, vmp, sampling
    for conv in ("test_convolution", "test_convolution_by_const", "test_convolution_pairwise"):
        assert conv in avx and conv in ours


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 3 (VERDICT r02 item 6 / ADVICE r02): the DEFAULT feature set must compile against the untouched reference.


def _strip_ks_fused_blocks(src):
    """core_impl.rs without the bodies that exist only under `#[cfg(feature = "ks-fused")]` (brace-matched)."""
    out, i = [], 0
    tag = '#[cfg(feature = "ks-fused")]'
    while True:
        j = src.find(tag, i)
        if j < 0:
            out.append(src[i:])
            return "".join(out)
        out.append(src[i:j])
        k = src.index("{", j)
        depth, p = 0, k
        while True:
            if src[p] == "{":
                depth += 1
            elif src[p] == "}":
                depth -= 1
                if depth == 0:
                    break
            p += 1
        i = p + 1


def test_default_features_only_touch_public_items_of_the_reference():
    cargo = read(CRATE, "Cargo.toml")
    feats = cargo[cargo.index("[features]"):]
    assert re.search(r'^default\s*=\s*\["core-fused"\]', feats, flags=re.M)
    assert re.search(r'^ks-fused\s*=\s*\["core-fused"\]', feats, flags=re.M)
    assert "ks-fused" not in re.search(r'^default\s*=.*$', feats, flags=re.M).group(0)
    assert os.path.exists(os.path.join(ROOT, "rust", "patches", "gglwe_prepared_data.patch"))
    core = strip_rust_comments(read(CRATE, "src", "core_impl.rs"))
    default_src = _strip_ks_fused_blocks(core)
    # a GGLWEPrepared's VmpPMat has no public accessor upstream (poulpy-core/src/layouts/prepared/gglwe.rs:20): the default build
    # must not reach for it; the key of every forward that remains is a GGSWPrepared (`g`), whose `data()` is public
    assert "k.data()" not in default_src
    assert "k.data()" in core                                             # ... and the gated forwards are still there
    assert default_src.count("g.data().as_ptr()") >= 6                    # glwe / gglwe / ggsw external product (+ _assign)
    for name in ("glwe_keyswitch", "glwe_keyswitch_assign", "gglwe_keyswitch", "gglwe_keyswitch_assign", "glwe_automorphism",
                 "glwe_automorphism_add_assign", "glwe_automorphism_sub_negate"):
        assert f"::{name}_default(" in default_src, name                  # without the feature: the reference algorithm
    if os.path.isdir(REF):
        ggsw = read(REF, "poulpy-core", "src", "layouts", "prepared", "ggsw.rs")
        gglwe = read(REF, "poulpy-core", "src", "layouts", "prepared", "gglwe.rs")
        assert re.search(r"pub fn data\(&self\)", ggsw)
        assert not re.search(r"pub fn data\(&self\)", gglwe) and "pub(crate) data" in gglwe   # the reason for the feature split
        # every inherent method the default build calls on a prepared key / layout view is `pub` in the reference
        for meth, files in (("data", ("layouts/prepared/ggsw.rs",)), ("at", ("layouts/gglwe.rs", "layouts/ggsw.rs")),
                            ("at_mut", ("layouts/gglwe.rs", "layouts/ggsw.rs"))):
            assert any(re.search(r"pub fn %s\b" % meth, read(REF, "poulpy-core", "src", f)) for f in files), meth


def test_sibling_modules_are_leased_from_a_bounded_pool():
    """no grow-only map keyed by ThreadId; a sibling goes back to the pool when its thread exits."""
    lib = strip_rust_comments(read(CRATE, "src", "lib.rs"))
    assert "HashMap<std::thread::ThreadId" not in lib and "HashMap<ThreadId" not in lib
    assert "struct SiblingPool" in lib and "struct Lease" in lib and "impl Drop for Lease" in lib
    assert "thread_local!" in lib
    assert re.search(r"self\.pool\.free\.lock\(\)\.unwrap\(\)\.pop\(\)", lib)        # reuse before cloning
    assert lib.count("ffi::pz_module_clone(") == 1
    assert "pz_abi_version()" in lib and "PZ_ABI_VERSION" in lib                     # checked when a handle is made
    from poulpy_amd.hal import PZ_ABI_VERSION
    assert re.search(r"pub const PZ_ABI_VERSION: u32 = (\d+);", lib).group(1) == str(PZ_ABI_VERSION)
    assert re.search(r"#define PZ_ABI_VERSION (\d+)u", read(ROOT, "include", "poulpy_hip.h")).group(1) == str(PZ_ABI_VERSION)   # what the library returns
    assert "return PZ_ABI_VERSION;" in read(ROOT, "poulpy_amd", "csrc", "api.hip")
    assert "pz_abi_version() != PZ_ABI_VERSION" in read(ROOT, "include", "poulpy_hip.hpp")                      # the C++ mirror refuses another revision


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 4: the additive batched API (src/batched.rs, generated by tools/gen_rust_batched.py)


def _ffi_calls(code):
    """[(name, number of top-level arguments)] of every `ffi::pz_*(...)` call (balanced parentheses / brackets / braces)."""
    out = []
    for m in re.finditer(r"ffi::(pz_\w+)\(", code):
        i, depth, args, cur = m.end(), 1, 0, False
        while depth:
            ch = code[i]
            if ch in "([{":
                depth += 1
                cur = True
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                args += 1
                cur = False
            elif not ch.isspace():
                cur = True
            i += 1
        out.append((m.group(1), args + (1 if cur else 0)))
    return out


def test_batched_rs_is_generated_and_every_ffi_call_has_the_headers_arity():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_batched.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    hdr = header_functions()
    code = strip_rust_comments(read(CRATE, "src", "batched.rs"))
    calls = _ffi_calls(code)
    assert len(calls) >= 55
    for name, nargs in calls:
        assert name in hdr, name
        assert nargs == len(hdr[name][1]), (name, nargs, len(hdr[name][1]))
    assert code.count("{") == code.count("}") and code.count("(") == code.count(")") and code.count("[") == code.count("]")
    # the same pin for the files written in earlier rounds
    for f in ("hal_impl.rs", "core_impl.rs", "lib.rs"):
        for name, nargs in _ffi_calls(strip_rust_comments(read(CRATE, "src", f))):
            assert name in hdr and nargs == len(hdr[name][1]), (f, name, nargs)


def test_batched_api_reaches_every_batched_entry_point_without_unsafe_in_its_signatures():
    hdr = header_functions()
    code = strip_rust_comments(read(CRATE, "src", "batched.rs"))
    used = {n for n, _ in _ffi_calls(code)}
    want = {n for n in hdr if n.endswith("_batched")} | {"pz_ggsw_external_product", "pz_device_alloc", "pz_device_free", "pz_memcpy_h2d", "pz_memcpy_d2h",
                                                         "pz_module_sync", "pz_module_pin_key", "pz_module_unpin_key", "pz_bcast_key", "pz_comm_unique_id",
                                                         "pz_comm_init_rank", "pz_comm_destroy", "pz_vmp_prepare"}
    assert want <= used, sorted(want - used)
    trait = code[code.index("pub trait HipBatched {"):code.index("impl HipBatched for Module<FFT64Hip> {")]
    sigs = re.findall(r"\n    fn (\w+)[^;]*;", trait)
    assert len(sigs) == len(set(sigs)) >= 45
    assert "unsafe" not in trait and "*const" not in trait and "*mut" not in trait          # nothing a caller must uphold by hand
    core = code[code.index("pub trait HipBatchedCore"):code.index("impl HipBatchedCore for Module<FFT64Hip>")]
    assert "unsafe" not in core and "*const" not in core and "*mut" not in core
    for needle in ("upload_glwe_batch", "download_glwe_batch", "glwe_external_product_batched", "glwe_keyswitch_batched", "blind_rotation_execute_batched",
                   "circuit_bootstrapping_execute_to_constant_batched", "circuit_bootstrapping_execute_to_exponent_batched", "bcast_key", "vmp_prepare_ggsws_on_device"):
        assert re.search(r"fn %s\b" % needle, code), needle
    # every `unsafe` block is a single FFI call or a byte view of a container whose length was asserted just above it
    for m_ in re.finditer(r"unsafe \{([^{}]*)\}", code):
        body = m_.group(1)
        assert "ffi::pz_" in body or "from_raw_parts" in body or ".add(" in body, body[:120]
    lib = strip_rust_comments(read(CRATE, "src", "lib.rs"))
    assert 'pub mod batched;' in lib and "HipBatched" in lib
    # containers keep their shapes private (the safety of the wrappers rests on it)
    for st in ("DeviceBuf", "DeviceVecZnx", "DeviceVecZnxDft", "DeviceVmpPMat"):
        body = re.search(r"pub struct %s<'m> \{(.*?)\n\}" % st, code, flags=re.S).group(1)
        assert "pub " not in body, st


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
def test_batched_api_only_uses_public_items_of_the_reference():
    """every accessor of a reference type that batched.rs calls is `pub` upstream; and the facts INTEGRATION.md states about what is NOT."""
    checks = [
        ("poulpy-core/src/layouts/glwe.rs", r"pub fn data\(&self\) -> &VecZnx<D>"), ("poulpy-core/src/layouts/glwe.rs", r"pub fn data_mut\(&mut self\)"),
        ("poulpy-core/src/layouts/lwe.rs", r"pub fn data\(&self\) -> &VecZnx<D>"), ("poulpy-core/src/layouts/lwe.rs", r"pub fn data_mut\(&mut self\)"),
        ("poulpy-core/src/layouts/prepared/ggsw.rs", r"pub fn data\(&self\) -> &VmpPMat<D, B>"),
        ("poulpy-core/src/layouts/gglwe.rs", r"pub fn data\(&self\) -> &MatZnx<D>"),
        ("poulpy-core/src/layouts/ggsw.rs", r"pub fn at\(&self, row: usize, col: usize\) -> GLWE<&\[u8\]>"),
        ("poulpy-hal/src/layouts/mat_znx.rs", r"pub fn cols_in\(&self\)"), ("poulpy-hal/src/layouts/mat_znx.rs", r"pub fn cols_out\(&self\)"),
        ("poulpy-hal/src/layouts/mat_znx.rs", r"impl<D: DataRef> ZnxView for MatZnx<D>"),
        ("poulpy-hal/src/layouts/vmp_pmat.rs", r"ZnxView for VmpPMat<D, B>"),
        ("poulpy-hal/src/layouts/module.rs", r"pub fn n\(&self\) -> usize"),
        ("poulpy-hal/src/layouts/znx_base.rs", r"fn as_ptr\(&self\) -> \*const Self::Scalar"),
    ]
    for path, pat in checks:
        assert re.search(pat, read(REF, path)), (path, pat)
    # why the API is additive and takes device keys instead of the bin-fhe key types
    assert re.search(r"impl<BE: Backend> BlindRotationExecute<CGGI, BE> for Module<BE>|impl<BE: Backend, .*BlindRotationExecute", read(REF, "poulpy-bin-fhe/src/blind_rotation/algorithms/cggi/algorithm.rs"))
    assert "pub(crate) data: Vec<GGSWPrepared<D, B>>" in read(REF, "poulpy-bin-fhe/src/blind_rotation/layouts/key_prepared.rs")
    assert "pub(crate) keys: Vec<GGSW<D>>" in read(REF, "poulpy-bin-fhe/src/blind_rotation/layouts/key.rs")
    assert re.search(r"impl<D: DataRef, BRT: BlindRotationAlgo> WriterTo for BlindRotationKey<D, BRT>", read(REF, "poulpy-bin-fhe/src/blind_rotation/layouts/key.rs"))
    assert "pub(crate) data: Vec<VecZnx<Vec<u8>>>" in read(REF, "poulpy-bin-fhe/src/blind_rotation/lut.rs")
    assert "pub(crate) data" in read(REF, "poulpy-core/src/layouts/prepared/gglwe.rs")


def test_batched_per_op_wrappers_check_the_ring_degree_against_the_module():
    """the C ABI's per-op `pz_vec_znx_*_batched` entry points stride by the MODULE's n; a safe wrapper that only
    compared the containers with each other let an LWE batch (n = n_lwe + 1) or a container of another module through.  Every wrapper
    that passes a DeviceVecZnx to such an entry point asserts `owns(self, ..)` for each container; `owns` pins n, byte length and device."""
    code = strip_rust_comments(read(CRATE, "src", "batched.rs"))
    m = re.search(r"fn owns\(module: &Module<FFT64Hip>, v: &DeviceVecZnx\) -> bool \{(.*?)\n\}", code, re.S)
    assert m, "owns() helper missing"
    body = m.group(1)
    assert "v.n == module.n()" in body and "v.buf.len()" in body and "pz_module_device" in body
    per_op = ["add_into", "sub", "add_assign", "sub_assign", "sub_negate_assign", "negate", "copy", "zero", "rotate", "lsh", "rsh", "normalize", "big_normalize"]
    for name in per_op:
        fm = re.search(r"\n    fn vec_znx_%s_batched\(&self([^\n;]*)\) \{(.*?)\n    \}" % name, code, re.S)
        assert fm, name
        sig, fbody = fm.group(1), fm.group(2)
        containers = re.findall(r"(\w+): &(?:mut )?DeviceVecZnx", sig)
        assert containers, name
        for c in containers:
            assert f"owns(self, {c})" in fbody, (name, c)
        assert fbody.index("assert!") < fbody.index("ffi::pz_vec_znx_"), name
    tests_rs = strip_rust_comments(read(CRATE, "src", "tests.rs"))
    assert tests_rs.count("#[should_panic") >= 2 and "lwe_batch_alloc(4, 636" in tests_rs and "vec_znx_copy_batched(&mut res, 0, &a, 0)" in tests_rs
