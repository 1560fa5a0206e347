// api_lwe.hip — the LWE glue of the gate bootstrap (BASELINE configs[3]: blind rotation + key switch) on device-resident batches:
//   mod_switch_2n        poulpy-bin-fhe/src/blind_rotation/algorithms/mod.rs:136-176
//   lwe_sample_extract   poulpy-core/src/api/conversion.rs:15-40
//   lwe_keyswitch        poulpy-core/src/keyswitching/lwe.rs:49-94
//   glwe_from_lwe        poulpy-core/src/conversion/lwe_to_glwe.rs:46-121
//   lwe_from_glwe        poulpy-core/src/conversion/glwe_to_lwe.rs:42-90
// An LWE is the reference's container: VecZnx(n = n_lwe + 1, one column, `size` limbs), limb i = [b, a_0 .. a_{n_lwe-1}]; a batch is
// `batch` of them back to back.  The embeddings / extractions are index kernels, everything else is the batched key switch.
#include "api_common.hpp"

namespace pz {

struct LweIdx {
    const long long* src;
    long long* dst;
    long long total;        // elements (or element pairs) this launch covers
    int n;                  // ring degree of the GLWE side
    int len;                // n_lwe + 1
    int lwe_size, glwe_size, glwe_cols;
    int min_size;           // limbs that carry data (the rest is zero)
    int base2k, log2n, negate;
};

// mod.rs:136-171: limb 0 (negated for rot_dir = Left), then either rounded down to log2n - 1 bits (base2k > log2n) or extended by
// the following limbs to log2n bits (the low limbs are appended unsigned-shifted and NOT negated, as in the reference)
__global__ void __launch_bounds__(256) k_lwe_mod_switch(LweIdx g) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= g.total) return;
    const long long b = idx / g.len;
    const int j = (int)(idx % g.len);
    const long long* lwe = g.src + b * (long long)g.lwe_size * g.len;
    unsigned long long y = (unsigned long long)lwe[j];
    if (g.negate) y = 0ull - y;
    if (g.base2k > g.log2n) {
        const int diff = g.base2k - (g.log2n - 1);
        y = (unsigned long long)((long long)(y + (1ull << (diff - 1))) >> diff);
    } else {
        const int rem = g.base2k - (g.log2n % g.base2k);
        const int size = (g.log2n + g.base2k - 1) / g.base2k;
        for (int i = 1; i < size; ++i) {
            const long long x = lwe[(long long)i * g.len + j];
            if (i == size - 1 && rem != g.base2k) y = (y << (g.base2k - rem)) + (unsigned long long)(x >> rem);
            else y = (y << g.base2k) + (unsigned long long)x;
        }
    }
    g.dst[idx] = (long long)y;
}

// lwe.rs:69-80 / lwe_to_glwe.rs:71-80: a zeroed two-column container of glwe_size limbs; limb i < min_size: b -> X^0 of column 0,
// a_0.. -> the first n_lwe coefficients of column 1.  Two coefficients per thread.
__global__ void __launch_bounds__(256) k_lwe_embed(LweIdx g) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // pair index
    if (idx >= g.total) return;
    const int half = g.n >> 1;
    const int jp = (int)(idx % half);
    long long t = idx / half;
    const int col = (int)(t % 2); t /= 2;
    const int limb = (int)(t % g.glwe_size);
    const long long b = t / g.glwe_size;
    long long v0 = 0, v1 = 0;
    if (limb < g.min_size) {
        const long long* lwe = g.src + (b * g.lwe_size + limb) * (long long)g.len;
        const int j = 2 * jp;
        if (col == 0) { if (j == 0) v0 = lwe[0]; }
        else {
            if (j < g.len - 1) v0 = lwe[1 + j];
            if (j + 1 < g.len - 1) v1 = lwe[2 + j];
        }
    }
    *reinterpret_cast<longlong2*>(g.dst + 2 * idx) = make_longlong2(v0, v1);
}

// conversion.rs:28-39: limb i < min_size of the LWE = [X^0 of column 0, the first n_lwe coefficients of column 1]; the other limbs zero
__global__ void __launch_bounds__(256) k_lwe_extract(LweIdx g) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= g.total) return;
    const int j = (int)(idx % g.len);
    long long t = idx / g.len;
    const int limb = (int)(t % g.lwe_size);
    const long long b = t / g.lwe_size;
    long long v = 0;
    if (limb < g.min_size) {
        const long long* glwe = g.src + (b * g.glwe_size + limb) * (long long)g.glwe_cols * g.n;
        v = j == 0 ? glwe[0] : glwe[(long long)g.n + (j - 1)];
    }
    g.dst[idx] = v;
}

static int launch_lwe_embed(pz_module* M, long long* dst, int glwe_size, const long long* lwe, int n_lwe, int lwe_size, int min_size, size_t batch) {
    LweIdx g{};
    g.src = lwe; g.dst = dst; g.n = (int)M->n; g.len = n_lwe + 1; g.lwe_size = lwe_size; g.glwe_size = glwe_size; g.glwe_cols = 2;
    g.min_size = min_size;
    g.total = (long long)batch * glwe_size * 2 * (M->n / 2);
    KTimer kt(M, PZ_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_lwe_embed, dim3((unsigned)((g.total + 255) / 256)), dim3(256), 0, M->stream, g);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}
static int launch_lwe_extract(pz_module* M, long long* res, int res_n_lwe, int res_size, const long long* glwe, int glwe_cols, int glwe_size,
                              size_t batch) {
    LweIdx g{};
    g.src = glwe; g.dst = res; g.n = (int)M->n; g.len = res_n_lwe + 1; g.lwe_size = res_size; g.glwe_size = glwe_size; g.glwe_cols = glwe_cols;
    g.min_size = std::min(res_size, glwe_size);
    g.total = (long long)batch * res_size * g.len;
    KTimer kt(M, PZ_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_lwe_extract, dim3((unsigned)((g.total + 255) / 256)), dim3(256), 0, M->stream, g);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}

// api.hip: the batched key switch without taking the module lock (the caller holds it)
int glwe_keyswitch_nolock(pz_module* M, int64_t* res, const int64_t* a, const double* key_pmat, const pz_glwe_op_params* p, size_t batch);

}  // namespace pz

static int lwe_ptr_checks(pz_module* M, const void* a, const void* b, size_t n_lwe) {
    PZ_REQUIRE(is_device_ptr(a) && is_device_ptr(b), "batched entry points take device pointers");
    PZ_REQUIRE(n_lwe >= 1 && n_lwe <= M->n, "LWE dimension %zu must be in [1, n = %llu]", n_lwe, (unsigned long long)M->n);
    return PZ_OK;
}

extern "C" {

int pz_lwe_mod_switch_2n_batched(pz_module* M, int64_t* res, const int64_t* lwe, size_t n_lwe, size_t lwe_size, size_t base2k, size_t n2,
                                 int negate, size_t batch) {
    PZ_ENTER(M);
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(lwe), "batched entry points take device pointers");
    PZ_REQUIRE(n_lwe >= 1 && lwe_size >= 1, "mod_switch_2n: empty LWE");
    PZ_REQUIRE(base2k >= 1 && base2k <= 62, "base2k %zu out of range", base2k);
    PZ_REQUIRE(n2 >= 2 && n2 <= ((size_t)1 << 40), "mod_switch_2n: n = %zu out of range", n2);
    int bits = 0;
    for (size_t v = n2 - 1; v; v >>= 1) ++bits;
    const int log2n = bits + 1;   // usize::BITS - (n - 1).leading_zeros() + 1  (mod.rs:139)
    if ((int)base2k <= log2n)
        PZ_REQUIRE(lwe_size >= (size_t)((log2n + base2k - 1) / base2k), "mod_switch_2n: %zu limbs of %zu bits do not reach %d bits", lwe_size,
                   base2k, log2n);
    if (batch == 0) return PZ_OK;
    LweIdx g{};
    g.src = (const long long*)lwe; g.dst = (long long*)res; g.len = (int)n_lwe + 1; g.lwe_size = (int)lwe_size;
    g.base2k = (int)base2k; g.log2n = log2n; g.negate = negate ? 1 : 0;
    g.total = (long long)batch * g.len;
    KTimer kt(M, PZ_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_lwe_mod_switch, dim3((unsigned)((g.total + 255) / 256)), dim3(256), 0, M->stream, g);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}

int pz_lwe_sample_extract_batched(pz_module* M, int64_t* res, size_t res_n_lwe, size_t res_size, const int64_t* a, size_t a_cols,
                                  size_t a_size, size_t batch) {
    PZ_ENTER(M);
    PZ_TRY(lwe_ptr_checks(M, res, a, res_n_lwe));
    PZ_REQUIRE(a_cols >= 2 && res_size >= 1 && a_size >= 1, "lwe_sample_extract: the GLWE needs a mask column and both sides a limb");
    if (batch == 0) return PZ_OK;
    return launch_lwe_extract(M, (long long*)res, (int)res_n_lwe, (int)res_size, (const long long*)a, (int)a_cols, (int)a_size, batch);
}

// lwe.rs:49-94.  p: the rank-1 -> rank-1 key switch (rank = rank_out = 1; a_size / a_base2k = the input LWE's, res_size / res_base2k =
// the output LWE's).  The two intermediate GLWEs live in the module's second workspace.
int pz_lwe_keyswitch_batched(pz_module* M, int64_t* res, size_t res_n_lwe, const int64_t* a, size_t a_n_lwe, const double* ksk_pmat,
                             const pz_glwe_op_params* p, size_t batch) {
    PZ_ENTER(M);
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_TRY(lwe_ptr_checks(M, res, a, res_n_lwe));
    PZ_REQUIRE(a_n_lwe >= 1 && a_n_lwe <= M->n, "LWE dimension %zu must be in [1, n]", a_n_lwe);
    PZ_REQUIRE(p->rank == 1 && p->rank_out == 1, "lwe_keyswitch: the key maps rank 1 -> rank 1 (lwe.rs:70-92)");
    PZ_REQUIRE(p->a_size >= 1 && p->res_size >= 1, "lwe_keyswitch: empty shape");
    if (batch == 0) return PZ_OK;
    const size_t n = M->n;
    const size_t in_b = align256(batch * n * 2 * p->a_size * 8), out_b = align256(batch * n * 2 * p->res_size * 8);
    PZ_TRY(ws2_reserve(M, in_b + out_b));
    char* wbase = (char*)M->ws2;
    long long* glwe_in; long long* glwe_out;
    PZ_TRY(ws_take(M, wbase, in_b, &glwe_in));
    PZ_TRY(ws_take(M, wbase, out_b, &glwe_out));
    PZ_TRY(launch_lwe_embed(M, glwe_in, (int)p->a_size, (const long long*)a, (int)a_n_lwe, (int)p->a_size, (int)p->a_size, batch));
    PZ_TRY(glwe_keyswitch_nolock(M, (int64_t*)glwe_out, (const int64_t*)glwe_in, ksk_pmat, p, batch));
    return launch_lwe_extract(M, (long long*)res, (int)res_n_lwe, (int)p->res_size, glwe_out, 2, (int)p->res_size, batch);
}

// lwe_to_glwe.rs:46-121.  p: the rank 1 -> rank_out key switch; p->a_size = limbs of the intermediate GLWE = ceil(lwe_size * lwe_base2k /
// key_base2k), p->a_base2k must be the key's base (the reference builds it so, :65-70).
int pz_glwe_from_lwe_batched(pz_module* M, int64_t* res, const int64_t* lwe, size_t n_lwe, size_t lwe_size, size_t lwe_base2k,
                             const double* ksk_pmat, const pz_glwe_op_params* p, size_t batch) {
    PZ_ENTER(M);
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_TRY(lwe_ptr_checks(M, res, lwe, n_lwe));
    PZ_REQUIRE(p->rank == 1, "glwe_from_lwe: the key's input rank is 1");
    PZ_REQUIRE(p->a_base2k == p->key_base2k, "glwe_from_lwe: the embedded GLWE has the key's base (lwe_to_glwe.rs:65-70)");
    PZ_REQUIRE(lwe_size >= 1 && lwe_base2k >= 1 && lwe_base2k <= 62, "glwe_from_lwe: bad LWE shape");
    if (batch == 0) return PZ_OK;
    const size_t n = M->n, gsz = p->a_size;
    const bool same = lwe_base2k == p->key_base2k;
    PZ_REQUIRE(!same || lwe_size <= gsz, "glwe_from_lwe: %zu LWE limbs do not fit the %zu limbs of the embedded GLWE", lwe_size, gsz);
    const size_t g_b = align256(batch * n * 2 * gsz * 8), c_b = same ? 0 : align256(batch * n * 2 * lwe_size * 8);
    PZ_TRY(ws2_reserve(M, g_b + c_b));
    char* wbase = (char*)M->ws2;
    long long* glwe; long long* conv;
    PZ_TRY(ws_take(M, wbase, g_b, &glwe));
    PZ_TRY(ws_take(M, wbase, c_b, &conv));
    if (same) {
        PZ_TRY(launch_lwe_embed(M, glwe, (int)gsz, (const long long*)lwe, (int)n_lwe, (int)lwe_size, (int)lwe_size, batch));
    } else {
        // :82-116: each column embedded in the LWE's base, then vec_znx_normalize into the key's base
        PZ_TRY(launch_lwe_embed(M, conv, (int)lwe_size, (const long long*)lwe, (int)n_lwe, (int)lwe_size, (int)lwe_size, batch));
        DV dg{glwe, (long long)(n * 2 * gsz), 2, (int)gsz}, dc{conv, (long long)(n * 2 * lwe_size), 2, (int)lwe_size};
        for (int c = 0; c < 2; ++c) PZ_TRY(dev_normalize(M, (int)batch, dg, (int)p->key_base2k, 0, c, dc, (int)lwe_base2k, c));
    }
    return glwe_keyswitch_nolock(M, res, (const int64_t*)glwe, ksk_pmat, p, batch);
}

// glwe_to_lwe.rs:42-90.  p: the rank -> 1 key switch (rank_out = 1; res_size / res_base2k = the LWE's); a_idx: the coefficient to
// extract (the GLWE is multiplied by X^-a_idx first)
int pz_lwe_from_glwe_batched(pz_module* M, int64_t* res, size_t res_n_lwe, const int64_t* a, size_t a_idx, const double* ksk_pmat,
                             const pz_glwe_op_params* p, size_t batch) {
    PZ_ENTER(M);
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_TRY(lwe_ptr_checks(M, res, a, res_n_lwe));
    PZ_REQUIRE(p->rank >= 1 && p->rank_out == 1, "lwe_from_glwe: the key maps rank -> 1");
    PZ_REQUIRE(a_idx < M->n, "lwe_from_glwe: coefficient index %zu >= n", a_idx);
    if (batch == 0) return PZ_OK;
    const size_t n = M->n, cols = p->rank + 1;
    const size_t in_b = a_idx ? align256(batch * n * cols * p->a_size * 8) : 0, out_b = align256(batch * n * 2 * p->res_size * 8);
    PZ_TRY(ws2_reserve(M, in_b + out_b));
    const long long* src = (const long long*)a;
    char* wbase = (char*)M->ws2;
    long long* rot; long long* glwe1;
    PZ_TRY(ws_take(M, wbase, in_b, &rot));
    PZ_TRY(ws_take(M, wbase, out_b, &glwe1));
    if (a_idx) {
        const int npolys = (int)(batch * cols * p->a_size);
        PolyMap mp{1, 1, (long long)n, 0, 0, 0};
        PZ_TRY(launch_rotate(M, npolys, src, mp, rot, mp, 0, 1, nullptr, 0, 0, -(long long)a_idx));
        src = rot;
    }
    PZ_TRY(glwe_keyswitch_nolock(M, (int64_t*)glwe1, (const int64_t*)src, ksk_pmat, p, batch));
    return launch_lwe_extract(M, (long long*)res, (int)res_n_lwe, (int)p->res_size, glwe1, 2, (int)p->res_size, batch);
}

}  // extern "C"
