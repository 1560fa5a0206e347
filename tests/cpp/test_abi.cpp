// Compiled-code client of the C ABI through the header-only C++ mirror (include/poulpy_hip.hpp): the op sequence of the
// reference's external product (poulpy-core/src/external_product/glwe.rs:99-141,197-271) and of config 1 (DFT + SVP) written
// the way a C++ / Rust backend shim would, checked bit for bit against the CPU oracle (oracle/fft64_ref.h).  Test
// infrastructure: built and run by tests/test_cpp_abi.py on the GPU box (g++, links libpoulpy_hip.so and the oracle).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "fft64_ref.h"
#include "poulpy_hip.hpp"

static void fill_uniform(std::vector<int64_t>& v, int log_bound, std::mt19937_64& rng) {
    const int64_t h = int64_t(1) << (log_bound - 1);
    std::uniform_int_distribution<int64_t> d(-h, h - 1);
    for (auto& x : v) x = d(rng);
}
#define REQUIRE(cond, msg)                                      \
    do {                                                        \
        if (!(cond)) {                                          \
            std::fprintf(stderr, "FAILED: %s (%s)\n", msg, #cond); \
            return 1;                                           \
        }                                                       \
    } while (0)

// config 1 (BASELINE configs[0] shape): N = 2^10, 2 limbs, DFT + SVP apply + IDFT + normalize through host pointers
static int test_config1_svp(std::mt19937_64& rng) {
    const size_t n = 1024, cols = 2, size = 2, base2k = 17;
    pz::Module mod(n);
    pzr_tables* t = pzr_tables_new(n);
    std::vector<int64_t> s(n * cols), b(n * cols * size), res_h(n * cols * size, -3), res_r(n * cols * size, 5);
    fill_uniform(s, base2k, rng);
    fill_uniform(b, base2k, rng);
    std::vector<double> pp_h(n * cols), pp_r(n * cols), d_h(n * cols * size), d_r(n * cols * size);
    pz::ScalarZnx sz{s.data(), n, cols};
    pz::VecZnx bv{b.data(), n, cols, size}, rh{res_h.data(), n, cols, size};
    pz::SvpPPol pph{pp_h.data(), n, cols};
    pz::VecZnxDft dh{d_h.data(), n, cols, size};
    for (size_t c = 0; c < cols; ++c) {
        mod.svp_prepare(pph, c, sz, c);
        pzr_svp_prepare(t, pp_r.data(), cols, c, s.data(), cols, c);
    }
    for (size_t c = 0; c < cols; ++c) {
        mod.svp_apply_dft(dh, c, pph, c, bv, c);
        pzr_svp_apply_dft(t, d_r.data(), cols, size, c, pp_r.data(), cols, c, b.data(), cols, size, c);
    }
    pz::VecZnxBig big_h = mod.vec_znx_idft_apply_consume(dh);
    pzr_vec_znx_idft_apply_consume(t, d_r.data(), cols, size);
    for (size_t c = 0; c < cols; ++c) {
        mod.vec_znx_big_normalize(rh, base2k, 0, c, big_h, base2k, c);
        pzr_vec_znx_normalize(n, res_r.data(), cols, size, base2k, 0, c, reinterpret_cast<int64_t*>(d_r.data()), cols, size, base2k, c);
    }
    REQUIRE(std::memcmp(res_h.data(), res_r.data(), res_h.size() * 8) == 0, "config 1: DFT + SVP + IDFT + normalize differs from the oracle");
    // error behaviour: a column out of range is reported, never UB
    bool threw = false;
    try {
        mod.vec_znx_big_normalize(rh, base2k, 0, cols, big_h, base2k, 0);
    } catch (const pz::Error& e) {
        threw = e.status == PZ_ERR_INVALID;
    }
    REQUIRE(threw, "out-of-range column must throw pz::Error(PZ_ERR_INVALID)");
    pzr_tables_free(t);
    return 0;
}

// batched external product on device-resident ciphertexts (the CoreImpl-level boundary), N = 4096, 4 limbs (configs[1] shape)
static int test_external_product_batched(std::mt19937_64& rng) {
    const size_t n = 4096, rank = 1, cols = rank + 1, size = 4, dnum = 4, base2k = 17, batch = 6;
    pz::Module mod(n);
    pzr_tables* t = pzr_tables_new(n);
    std::vector<int64_t> mat(n * dnum * cols * cols * size);
    fill_uniform(mat, base2k, rng);
    std::vector<double> pm_h(mat.size()), pm_r(mat.size());
    pz::MatZnx mz{mat.data(), n, dnum, cols, cols, size};
    pz::VmpPMat pmh{pm_h.data(), n, dnum, cols, cols, size};
    mod.vmp_prepare(pmh, mz);
    pzr_vmp_prepare(t, pm_r.data(), mat.data(), dnum, cols, cols, size);
    const size_t ct = n * cols * size;
    std::vector<int64_t> a(batch * ct), want(batch * ct), got(batch * ct, 0x55);
    fill_uniform(a, base2k, rng);
    for (size_t b = 0; b < batch; ++b)
        pzr_glwe_external_product(t, rank, want.data() + b * ct, size, base2k, a.data() + b * ct, size, base2k, pm_r.data(), dnum, size, 1, base2k);
    void *d_a = nullptr, *d_k = nullptr, *d_r = nullptr;
    pz::check(pz_device_alloc(mod.raw(), a.size() * 8, &d_a), "device_alloc");
    pz::check(pz_device_alloc(mod.raw(), pm_h.size() * 8, &d_k), "device_alloc");
    pz::check(pz_device_alloc(mod.raw(), got.size() * 8, &d_r), "device_alloc");
    pz::check(pz_memcpy_h2d(mod.raw(), d_a, a.data(), a.size() * 8), "h2d");
    pz::check(pz_memcpy_h2d(mod.raw(), d_k, pm_h.data(), pm_h.size() * 8), "h2d");
    pz_glwe_op_params p{};
    p.rank = rank; p.dnum = dnum; p.dsize = 1; p.key_size = size; p.key_base2k = base2k; p.a_size = size; p.a_base2k = base2k;
    p.res_size = size; p.res_base2k = base2k; p.rank_out = rank;
    mod.glwe_external_product_batched((int64_t*)d_r, (const int64_t*)d_a, (const double*)d_k, p, batch);
    mod.sync();
    pz::check(pz_memcpy_d2h(mod.raw(), got.data(), d_r, got.size() * 8), "d2h");
    REQUIRE(std::memcmp(got.data(), want.data(), got.size() * 8) == 0, "batched external product differs from the oracle");
    pz_device_free(mod.raw(), d_a);
    pz_device_free(mod.raw(), d_k);
    pz_device_free(mod.raw(), d_r);
    pzr_tables_free(t);
    return 0;
}

int main() {
    std::mt19937_64 rng(0x5eed);
    if (test_config1_svp(rng)) return 1;
    if (test_external_product_batched(rng)) return 1;
    std::printf("test_abi: OK\n");
    return 0;
}
