// api_glwe.hip — the batched GLWE product behind the C ABI: GLWE (x) GGSW external product, GLWE key switch, the glwe_automorphism family,
// tensor relinearization, and the composites that are loops of those (ggsw_external_product, ggsw_expand_row, ggsw_from_gglwe, glwe_trace).
// Reference call stacks: poulpy-core/src/external_product/glwe.rs:99-271, keyswitching/glwe.rs:53-380, automorphism/glwe_ct.rs:51-275,
// operations/glwe.rs:541-607.
//
// glwe_op = validate -> pick the pipeline for the shape -> per wave of ciphertexts, the launch sequence of that pipeline:
//   fused      pass 1 | row pass + VMP + inverse row pass | tail         (plans with 128 / 256-point rows; the measured path)
//                per family: plain / spectral automorphism / cross-base output / digits (dsize > 1, a table for the middle kernel) / N = 4096 two-kernel
//   small ring whole forward transform | product + whole inverse + carry chain   (N = 1024 / 2048)
//   five-kernel the reference's op sequence, one kernel per HAL op                (every other shape; fusion switched off)
// One definition of "which pipeline" serves the call and the workspace query (pz_glwe_op_workspace_bytes).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "api_common.hpp"
#include "api_glwe.hpp"

using namespace pz;

// ------------------------------------------------------------------------------
// host containers at the batched GLWE entry points (what the Rust shim's CoreImpl overrides pass: poulpy-hal buffers are host
// addressable by contract).  Ciphertexts are staged like any per-op argument; prepared keys get a device mirror (api.hip).
// ------------------------------------------------------------------------------
struct GlweArgs {
    Stage sa, sr;
    int64_t* res = nullptr;
    const int64_t* a = nullptr;
    const double* key = nullptr;
    bool host = false;
};
static int glwe_args_in(pz_module* M, GlweArgs& g, int64_t* res, const int64_t* a, const double* pmat, size_t res_bytes, size_t a_bytes,
                        size_t key_bytes) {
    PZ_REQUIRE(res != nullptr && a != nullptr && pmat != nullptr, "null argument");
    PZ_TRY(resolve_key(M, pmat, key_bytes, &g.key));
    PZ_TRY(g.sa.in(a, a_bytes, true, false, M));
    if ((const void*)res == (const void*)a) {   // *_assign forms
        PZ_REQUIRE(res_bytes == a_bytes, "in-place call with different layouts for a and res");
        g.sr.M = M; g.sr.dev = g.sa.dev; g.sa.out = true;
    } else {
        PZ_TRY(g.sr.in(res, res_bytes, false, true, M));
    }
    g.res = (int64_t*)g.sr.dev; g.a = (const int64_t*)g.sa.dev;
    g.host = g.sa.owned || g.sr.owned;
    return PZ_OK;
}
static int glwe_args_out(pz_module* M, GlweArgs& g) {
    PZ_TRY(g.sr.finish());
    PZ_TRY(g.sa.finish());
    return finish_call(M, g.host);
}

// ------------------------------------------------------------------------------
// shapes, workspaces, pipeline choice
// ------------------------------------------------------------------------------
struct OpShape {
    int cols_a, cols_in, cols_out;  // columns of `a`, VMP input columns, output columns
    int a_col0;                     // first column of `a` that enters the product
    int a_size_eff;                 // limbs of `a` in the key's base (after optional conversion)
    bool convert;
};
// kind: 0 external product, 1 key switch (mask columns 1.. of a GLWE), 2 tensor relinearization (operations/glwe.rs:541-607: `a` is
// a GLWETensor of cols + pairs columns, the pairs = rank (rank + 1) / 2 columns behind the first cols = rank + 1 are key-switched
// and the first cols are added to every column of the big value)
static OpShape op_shape(const pz_glwe_op_params* p, bool ks, bool tensor = false) {
    OpShape s;
    if (tensor) {
        const int cols = (int)p->rank + 1, pairs = (int)(p->rank * (p->rank + 1) / 2);
        s.cols_a = cols + pairs; s.cols_in = pairs; s.cols_out = cols; s.a_col0 = cols;
        s.convert = p->a_base2k != p->key_base2k;
        s.a_size_eff = s.convert ? (int)((p->a_size * p->a_base2k + p->key_base2k - 1) / p->key_base2k) : (int)p->a_size;
        return s;
    }
    s.a_col0 = ks ? 1 : 0;
    s.cols_a = (int)p->rank + 1;
    s.cols_in = ks ? (int)p->rank : (int)p->rank + 1;
    s.cols_out = ks ? (int)p->rank_out + 1 : (int)p->rank + 1;
    s.convert = p->a_base2k != p->key_base2k;
    s.a_size_eff = s.convert ? (int)((p->a_size * p->a_base2k + p->key_base2k - 1) / p->key_base2k) : (int)p->a_size;
    return s;
}

struct OpWs {
    size_t a_conv, a_dft, res_dft, tmp_dft, T, res_tmp, total;
};
static OpWs op_ws(const pz_module* M, const pz_glwe_op_params* p, const OpShape& s, size_t chunk, bool ks, bool au = false) {
    OpWs w;
    const size_t n8 = (size_t)M->n * 8;
    const size_t dsz = p->dsize;
    w.a_conv = s.convert ? align256(chunk * n8 * s.cols_a * s.a_size_eff) : 0;
    w.a_dft = align256(chunk * n8 * s.cols_in * (size_t)s.a_size_eff);
    w.res_dft = align256(chunk * n8 * s.cols_out * p->key_size);
    w.tmp_dft = dsz > 1 ? align256(chunk * n8 * (s.cols_out * p->key_size + (ks ? s.cols_in * (size_t)s.a_size_eff : 0))) : 0;
    const size_t tp = std::max((size_t)s.cols_in * s.a_size_eff, (size_t)s.cols_out * p->key_size);
    w.T = align256(chunk * tp * (size_t)M->m * sizeof(cplx));
    w.res_tmp = au ? align256(chunk * n8 * s.cols_out * p->res_size) : 0;  // normalized result before the final permutation
    w.total = w.a_conv + w.a_dft + w.res_dft + w.tmp_dft + w.T + w.res_tmp;
    return w;
}
static size_t pick_chunk(const pz_module* M, const pz_glwe_op_params* p, const OpShape& s, size_t batch) {
    if (M->chunk) return std::min(M->chunk, batch);
    // Measured on MI355X (profiles/r01_chunk_sweep.txt, r01_batch_sweep.txt): the intermediates do not stay in the Infinity
    // Cache anyway and every wave re-streams the key and pays the pipeline fill of the persistent middle kernel, so larger
    // waves win (128 -> 1024 ciphertexts per wave: +13 %); cap the workspace at ~24 GiB of the 288 GB.
    const size_t per_ct = (size_t)M->n * 8 * ((size_t)s.cols_in * s.a_size_eff + 2 * (size_t)s.cols_out * p->key_size);
    size_t c = ((size_t)24 << 30) / std::max<size_t>(per_ct, 1);
    c = std::max<size_t>(c & ~(size_t)7, 8);
    return std::min(c, batch);
}

// which pipeline glwe_op takes for a shape, and what it reserves there (one definition for the call and for the workspace query)
static bool fused_applies(const pz_module* M, const pz_glwe_op_params* p, const OpShape& s, bool tensor, bool au) {
    const int npi = s.cols_in * s.a_size_eff, npo = s.cols_out * (int)p->key_size;
    const bool digits = p->dsize > 1, cross_out = p->res_base2k != p->key_base2k;
    return M->fuse_mid && M->fuse_tail && tail_supported(M) && mid_supported(M, npi, npo) && !(tensor && s.convert) &&
           (!(digits || cross_out) || (M->plan.m2 == 128 && !au && (int)p->dnum * s.cols_in <= 255 && npo <= 255));
}
struct FusedWs {
    size_t key, conv, t, t2, rtmp, small2, total;
};
// Placement of T2' relative to the result: the tail of ciphertext b reads T2' + X and writes res + X and res + X + N*4 bytes (the two
// coefficient halves), the same X for every workgroup; with both buffers on the same 1 MiB phase (large allocations are 2 MiB aligned)
// the read and the two write streams of every workgroup meet on the same HBM channels: tail 3.55 ms per 1024 ciphertexts in most
// processes, 3.14 in some, depending on the physical pages (profiles/r02_t2_placement.txt).  T2' therefore sits 768 KiB out of phase
// with the result (mod 4 MiB).  Rounds 2-3 MEASURED the phase per call shape (eight candidates, an event pair each); on every box of
// round 3 and round 4 the tuned and the fixed placement measured the same (99 952 vs 99 181/s, 99 142 vs 98 445/s: noise), so the tuner,
// its shape cache and its two ABI functions were removed in round 4 (ABI version 4).
static constexpr size_t kT2Phase = 0xC0000, kT2PhaseMask = 0x3FFFFF;
static FusedWs fused_ws(const pz_module* M, const pz_glwe_op_params* p, const OpShape& s, size_t chunk, bool au) {
    FusedWs w;
    const size_t n8 = (size_t)M->n * 8, ksz = p->key_size;
    const size_t npi = (size_t)s.cols_in * s.a_size_eff, npo = (size_t)s.cols_out * ksz;
    w.key = align256((size_t)p->dnum * s.cols_in * npo * n8);
    w.conv = s.convert ? align256(chunk * n8 * s.cols_a * s.a_size_eff) : 0;
    w.t = align256(chunk * npi * M->m * sizeof(cplx));
    w.t2 = align256(chunk * npo * M->m * sizeof(cplx));
    // res_tmp holds the normalized result before the final permutation (mode 0 / gather scheme) OR, in the spectral form, the
    // body-column operand (min(a_size, key_size) limbs of one column): sized for the larger of the two
    const size_t body_limbs = std::min<size_t>((size_t)s.a_size_eff, ksz);
    w.rtmp = au ? align256(chunk * n8 * std::max((size_t)s.cols_out * p->res_size, body_limbs)) : 0;
    // cross-base output: the tail's key-base digits (cols_out x key_size limbs per ciphertext) before the cross-base pass
    w.small2 = p->res_base2k != p->key_base2k ? align256(chunk * n8 * s.cols_out * ksz) : 0;
    w.total = w.key + w.conv + w.t + w.t2 + w.rtmp + w.small2 + kMidDummyBytes + align256(M->ws_shift) + (kT2PhaseMask + 1);
    return w;
}
// N = 1024 / 2048: the two-kernel pipeline of device_small.hpp (plain products, key switches and the automorphism family; dsize 1, one
// base2k, <= 4 key limbs); `packed` = no OpLayout (the automorphism family needs it)
static bool small_ring_applies(const pz_module* M, const pz_glwe_op_params* p, const OpShape& s, bool ks, bool tensor, bool au, bool packed) {
    static const int small_env = exp_knob("POULPY_DBG_SMALL", 1);
    static const int small_au = exp_knob("POULPY_DBG_SMALL_AUTO", 1);
    const bool cross_out = p->res_base2k != p->key_base2k;   // (with an automorphism: phi and the cross-base pass do not commute)
    return small_env && M->small_path && M->fuse_mid && M->fuse_tail && M->n < 4096 && (!au || (small_au && ks && packed && !cross_out)) &&
           !tensor && p->dsize == 1 && M->dbg_stages == 7 && small_supported(M, s.cols_in * s.a_size_eff, (int)p->key_size);
}
struct SmallWs {
    size_t key, spectra, conv, digits, total;
};
static SmallWs small_ws(const pz_module* M, const pz_glwe_op_params* p, const OpShape& s, size_t chunk) {
    SmallWs w;
    const size_t n8 = (size_t)M->n * 8;
    w.key = align256((size_t)p->dnum * s.cols_in * s.cols_out * p->key_size * n8);                       // the key re-sliced
    w.spectra = align256(chunk * (size_t)(s.cols_in * s.a_size_eff) * (size_t)M->m * sizeof(cplx));      // the spectra of one wave
    w.conv = s.convert ? align256(chunk * n8 * s.cols_a * s.a_size_eff) : 0;
    w.digits = p->res_base2k != p->key_base2k ? align256(chunk * n8 * s.cols_out * p->key_size) : 0;    // key-base digits before the cross-base pass
    w.total = w.key + w.spectra + w.conv + w.digits;
    return w;
}

// everything a call of glwe_op derives from its arguments before it launches anything
struct GlweCall {
    pz_module* M;
    const pz_glwe_op_params* p;
    OpShape s;
    bool ks, tensor;
    const AutoSpec* au;
    const OpLayout* lay;
    int64_t* res; const int64_t* a; const double* pmat;
    size_t batch, chunk;
    long long n;
    int dsize, dnum, ksz;
    int npi, npo;          // polynomials per ciphertext entering / leaving the product
    int nrows, ncols;      // the key matrix: dnum * cols_in rows, cols_out * key_size columns
    long long a_ct, res_ct, a_bs, res_bs;   // packed sizes and actual strides of one ciphertext (i64 elements)
    int body_col;
    bool au_big;           // add / sub / sub_negate: phi acts on the big value
    unsigned au_p, au_g;   // Galois element mod 2n and its inverse
    bool digits, cross_out;
    bool want_rsh;         // glwe_trace asked for the one-bit shift on the way out
    bool* post_rsh;        // ... and is told here whether it happened
    // relinearization of a GLWETensor that exists only as 16-bit digits in the fused tail's tile order (glwe_relin_t16 below):
    // a16[column][ciphertext of this call][limb][n], a16_cs int16 elements between columns; `a` is not read
    const short* a16 = nullptr;
    long long a16_cs = 0;
    int cols_in() const { return s.cols_in; }
    int64_t* res_at(size_t b0) const { return res + (long long)b0 * res_bs; }
};

extern "C" {
// keyswitch: 0 external product, 1 key switch, 2 automorphism family, 3 tensor relinearization.  The figure is what the call reserves
// in the module's grow-only workspace (+ the 12.5 % growth slack of its first allocation); a key that is neither pinned nor mirrored
// costs its row-sliced copy, which is included.
size_t pz_glwe_op_workspace_bytes(const pz_module* M, const pz_glwe_op_params* p, size_t batch, int keyswitch) {
    if (!M || !p || p->key_size == 0 || p->a_size == 0) return 0;
    const bool tensor = keyswitch == 3, ks = keyswitch != 0, au = keyswitch == 2;
    const OpShape s = op_shape(p, ks, tensor);
    const size_t chunk = pick_chunk(M, p, s, batch);
    size_t bytes;
    if (fused_applies(M, p, s, tensor, au)) bytes = fused_ws(M, p, s, chunk, au).total;
    else if (small_ring_applies(M, p, s, ks, tensor, au, true)) bytes = small_ws(M, p, s, chunk).total;
    else bytes = op_ws(M, p, s, chunk, ks, au).total;
    return bytes + (bytes >> 3);
}
}

// ------------------------------------------------------------------------------
// pieces shared by the pipelines
// ------------------------------------------------------------------------------
// the wave's input in the key's base: `a` itself, or glwe_normalize into a_conv (external_product/glwe.rs:124-132)
static int wave_input(const GlweCall& c, size_t b0, int nb, int64_t* a_conv, DV* av) {
    *av = DV{(void*)(c.a + (long long)b0 * c.a_bs), c.a_bs, c.s.cols_a, (int)c.p->a_size};
    if (!c.s.convert) return PZ_OK;
    DV cv{a_conv, c.n * c.s.cols_a * c.s.a_size_eff, c.s.cols_a, c.s.a_size_eff};
    for (int col = 0; col < c.s.cols_a; ++col)
        PZ_TRY(dev_normalize(c.M, nb, cv, (int)c.p->key_base2k, 0, col, *av, (int)c.p->a_base2k, col));
    *av = cv;
    return PZ_OK;
}
// the permuted copy of the key a pipeline reads: the pinned / mirrored one if the caller declared the key immutable, else built now
static int wave_key(const GlweCall& c, cplx* scratch, bool small_ring, const cplx** out) {
    const size_t bytes = (size_t)c.nrows * c.ncols * (size_t)c.M->n * 8;
    for (auto& pk : c.M->pinned)
        if (pk.key == (const void*)c.pmat && pk.sliced && pk.bytes == bytes) { *out = pk.sliced; return PZ_OK; }
    // the key arrives in the standard device layout; its row-sliced copy is rebuilt per call (2 x 128 MiB of traffic at the metric
    // shape, ~4 % of a 128-ciphertext call) so that no stale copy can ever be used
    if (small_ring) PZ_TRY(launch_small_permute(c.M, c.pmat, scratch, c.nrows * c.ncols));
    else if (c.M->dbg_stages & 2) PZ_TRY(launch_permute_pmat(c.M, c.pmat, scratch, c.nrows * c.ncols));
    *out = scratch;
    return PZ_OK;
}
// the tail of a wave with everything at its defaults for this call (row-major T2', key-base limbs in, res out)
static TailCall wave_tail(const GlweCall& c, int nb, const cplx* T2, size_t b0) {
    TailCall t;
    t.batch = nb; t.T = T2; t.rowmajor = true; t.nlimbs = c.ksz; t.ncols = c.s.cols_out;
    t.res = (long long*)c.res_at(b0); t.res_bs = c.res_bs; t.res_cols = c.s.cols_out; t.res_size = (int)c.p->res_size;
    t.base2k = (int)c.p->res_base2k;
    t.body_col = c.body_col;
    return t;
}
static void tail_operand(TailCall& t, const DV& av, bool every_column) {
    t.small = (const long long*)av.p; t.small_bs = av.bs; t.small_cols = av.cols; t.small_size = av.size; t.small_all = every_column;
}

// dsize > 1 (external_product/glwe.rs:235-267, keyswitching/glwe.rs:332-379) as a table for the middle kernel: limb l of `a` is digit
// di = (dsize - 1 - l) mod dsize, element k = (l - (dsize - 1 - di)) / dsize of that digit's vector (vec_znx_dft_apply with step dsize,
// offset dsize - 1 - di); the vector has (a_size + di) / dsize elements (at most dnum for a key switch) and multiplies key rows k (all
// input columns) with limb_offset di, into a result of key_size - max(dsize - di - 2, 0) limbs (zero-tail semantics of SURVEY.md A.2)
static MidDigits digit_terms(const GlweCall& c) {
    MidDigits dg;
    const int dsize = c.dsize;
    for (int l = 0; l < c.s.a_size_eff; ++l) {
        const int di = ((dsize - 1 - l) % dsize + dsize) % dsize;
        const int k = (l - (dsize - 1 - di)) / dsize;
        int a_sz = (c.s.a_size_eff + di) / dsize;
        if (c.ks) a_sz = std::min(a_sz, c.dnum);
        if (k < 0 || k >= a_sz || k >= c.dnum) continue;
        const int r_sz = c.ksz - std::max(dsize - di - 2, 0);
        const int off = di * c.s.cols_out;
        const int cb = off < c.ncols ? std::min(c.s.cols_out * r_sz, c.ncols - off) : 0;
        if (cb <= 0) continue;
        for (int col = 0; col < c.s.cols_in; ++col) {
            dg.in[dg.n] = (unsigned char)(l * c.s.cols_in + col);
            dg.row[dg.n] = (unsigned char)(k * c.s.cols_in + col);
            dg.coff[dg.n] = (unsigned char)off;
            dg.cb[dg.n] = (unsigned char)cb;
            ++dg.n;
        }
    }
    return dg;
}

// ------------------------------------------------------------------------------
// fused pipeline
// ------------------------------------------------------------------------------
struct FusedBufs {
    cplx* key_scratch; int64_t* a_conv; cplx* T; cplx* T2; int64_t* res_tmp; int64_t* key_digits; cplx* mid_dummy;
    const cplx* Pp;
    short* side16 = nullptr;   // this wave's 16-bit side copy of pass 1's input (add / sub automorphism forms at rank 1: wave_spectral_tail), or null
};
static int fused_carve(const GlweCall& c, FusedBufs* f) {
    pz_module* M = c.M;
    const FusedWs fw = fused_ws(M, c.p, c.s, c.chunk, c.au != nullptr);
    PZ_TRY(ws_reserve(M, fw.total));
    char* base = (char*)M->ws;
    PZ_TRY(ws_take(M, base, fw.key, &f->key_scratch));
    PZ_TRY(ws_take(M, base, fw.conv, &f->a_conv));
    PZ_TRY(ws_take(M, base, fw.t, &f->T));
    base += (kT2Phase - (size_t)(((uintptr_t)base - (uintptr_t)c.res) & kT2PhaseMask)) & kT2PhaseMask;   // see kT2Phase
    base += align256(M->ws_shift);
    PZ_TRY(ws_take(M, base, fw.t2, &f->T2));
    PZ_TRY(ws_take(M, base, fw.rtmp, &f->res_tmp));
    PZ_TRY(ws_take(M, base, fw.small2, &f->key_digits));
    PZ_TRY(ws_take(M, base, kMidDummyBytes, &f->mid_dummy));
    return PZ_OK;
}

// N = 4096, plain external product / key switch / automorphism with <= 4 key limbs: two kernels, the spectra cross HBM once
// (device_small.hpp).  (round 3: the 8-slot tile of k_mid128r - 8 polynomials in, 8 out, 8 product rows: the external product with 4
// limbs, BASELINE configs[1] - beats the two-kernel form, 3.25 vs 3.16 M/s, profiles/r03_ab_small_vs_pipeline.txt; POULPY_DBG_SMALL=2
// forces the two-kernel form there too)
static bool n4096_two_kernel(const GlweCall& c) {
    static const int small_env = exp_knob("POULPY_DBG_SMALL", 1);
    static const int small_au4 = exp_knob("POULPY_DBG_SMALL_AUTO", 1);
    const pz_module* M = c.M;
    const bool mid8 = !c.ks && !c.au && c.npi == 8 && c.npo == 8 && std::min(c.nrows, c.npi) == 8 && small_env != 2;
    return small_env && M->small_path && (!c.au || (small_au4 && c.ks && !c.lay)) && !c.tensor && !c.digits && !c.cross_out &&
           M->dbg_stages == 7 && small_supported(M, c.npi, c.ksz) && !mid8;
}
static int wave_n4096_two_kernel(const GlweCall& c, const FusedBufs& f, size_t b0, int nb, const DV& av, const PolyMap& sm) {
    const bool rsh = c.want_rsh && c.au && c.au->mode != 0 && c.p->res_base2k <= 29;
    PZ_TRY(launch_small_fwd(c.M, nb * c.npi, (const long long*)av.p, sm, f.T));
    PZ_TRY(launch_small_inv(c.M, nb, f.T, f.Pp, c.npi, c.nrows, c.ncols, c.s.cols_out, c.ksz, (long long*)c.res_at(b0), c.res_bs, c.s.cols_out,
                            (int)c.p->res_size, c.ks ? (const long long*)av.p : nullptr, av.bs, c.s.cols_a, av.size, (int)c.p->res_base2k,
                            c.body_col, false, nullptr, 0, c.au != nullptr, c.au_p, c.au ? c.au->mode : 0, rsh));
    if (rsh) *c.post_rsh = true;
    return PZ_OK;
}

// Spectral form of the automorphism family (m2 = 128 plans).  X -> X^p with p = 1 mod 4: DFT(phi(a))[q] = DFT(a)[p q + (p-1)/4 mod m] is an
// affine map of the spectrum index that sends rows of the four-step layout to rows, so the middle kernel writes its product at the
// permuted position (k_mid128<.., PERM>) and the tail's inverse transform is phi(big) itself.  p = 3 mod 4 (X -> X^-1, the first step of
// every trace, among them): the spectrum of phi(a) is the CONJUGATE of a permuted spectrum (MidArgs::perm_ysign).
// POULPY_DBG_AUTO_SPECTRAL: 0 never; 2 not for the plain form (mode 0: key switch + signed permutation pass instead); 3 only p = 1 mod 4.
struct SpectralPerm { bool on = false; unsigned mul = 0, add = 0; bool conj = false; };
static SpectralPerm spectral_perm(const GlweCall& c) {
    static const int au_spec = exp_knob("POULPY_DBG_AUTO_SPECTRAL", 1);
    SpectralPerm sp;
    sp.on = au_spec && c.au && (c.au_big || au_spec == 1 || au_spec == 3) && ((c.au_p & 3u) == 1u || au_spec != 3) && c.M->plan.m2 == 128 &&
            c.M->dbg_stages == 7;
    if (!sp.on) return sp;
    const unsigned mm = (unsigned)c.M->m;
    if ((c.au_p & 3u) == 1u) {
        sp.mul = c.au_g & (mm - 1u);
        const unsigned long long c0 = (unsigned long long)(((c.au_p - 1u) >> 2) & (mm - 1u));
        sp.add = (unsigned)((mm - (unsigned)(((unsigned long long)sp.mul * c0) & (unsigned long long)(mm - 1u))) & (mm - 1u));
    } else {
        sp.conj = true;
        sp.mul = (mm - (c.au_g & (mm - 1u))) & (mm - 1u);                                            // (-p)^-1 mod m
        const unsigned long long c0 = (unsigned long long)((((unsigned long long)c.au_p + 1ull) >> 2) & (unsigned long long)(mm - 1u));
        sp.add = (unsigned)(((unsigned long long)sp.mul * c0) & (unsigned long long)(mm - 1u));      // (-p)^-1 (p + 1)/4
    }
    return sp;
}
// The tail of the spectral form adds ONE operand stream per column at the natural index: +-a[col] and, on the body column, the stream
// prepared here by one k_automorphism pass over that column into the (cache-resident) workspace - phi(body) for the plain form,
// +-phi(body) + a0 for add / sub / sub_negate.  No permutation pass over the result, no gathers in the tail, in-place forms safe.
// whether the spectral tail takes its body-column operand as 16-bit copies (see wave_spectral_tail)
static bool spectral_body16(const GlweCall& c) {
    pz_module* M = c.M;
    const long long n = c.n;
    static const int fold_knob = exp_knob("POULPY_DBG_AUTO_FOLD", 0);
    static const int b16_knob = exp_knob("POULPY_DBG_AUTO_BODY16", 1);
    const int64_t* a_end = c.a + (long long)c.batch * c.a_bs;
    const int64_t* r_end = c.res + (long long)c.batch * c.res_bs;
    const bool fold = fold_knob != 0 && (c.res >= a_end || c.a >= r_end) && !c.want_rsh;
    return b16_knob != 0 && !fold && !M->probe && (!c.want_rsh || (c.au_big && (int)c.p->key_base2k <= 14 && (int)c.p->res_base2k <= 29)) &&
           n >= 4096 && n <= 65536 && (int)c.p->key_base2k <= (c.au_big ? 15 : 16) && (int)c.p->res_base2k <= 31 && tail_rsh_supported(M) &&
           !(c.au_big && exp_knob("POULPY_DBG_AUTO_BODYADD", 0));
}
static int wave_spectral_tail(const GlweCall& c, const FusedBufs& f, size_t b0, int nb, const DV& av) {
    pz_module* M = c.M;
    const long long n = c.n;
    const int bl = std::min(av.size, c.ksz);   // the tail reads operand limbs j < min(key_size, a_size) only: the pre-pass covers exactly those
    PolyMap bsm{bl, 1, av.bs, (long long)av.cols * n, 0, 0}, bdm{bl, 1, (long long)bl * n, n, 0, 0};
    TailCall t = wave_tail(c, nb, f.T2, b0);
    tail_operand(t, av, true);
    // phi(body): prepared by a pre-pass in the workspace.  POULPY_DBG_AUTO_FOLD=1: gathered by the tail itself instead (round 4, VERDICT r03
    // item 3) - bit-exact and one kernel and 8.6 GB of traffic less per 1024 ciphertexts, but the tail goes from 5.05 to 8.3 - 8.8 ms
    // (the pre-pass costs 1.8 - 3.0): 16 dependent 8-byte gathers per thread and limb in front of the carry chain, for every Galois
    // element tried, conjugation included (profiles/r04_ab_auto_fold.txt).  A copy-rate pre-pass is the cheaper form.
    static const int fold_knob = exp_knob("POULPY_DBG_AUTO_FOLD", 0);
    // (never in place: other workgroups would gather from a body that this launch is already overwriting)
    const int64_t* a_end = c.a + (long long)c.batch * c.a_bs;
    const int64_t* r_end = c.res + (long long)c.batch * c.res_bs;
    const bool fold = fold_knob != 0 && (c.res >= a_end || c.a >= r_end) && !c.want_rsh;   // (not for glwe_trace: the shifted-store variant has no gathered form)
    // Round 6, plain form: the pre-pass leaves phi(body) as 16-bit values in the tail's own tile order (2 B written and 2 B read per coefficient
    // instead of 8) where the digits are expected to fit - a key base of at most 16 bits - and the body column then rides on the f64 chain of the
    // sign-only tail with that operand (k_inv_tail<.., NZF = 7, SGN>) instead of the operand variant's integer chain.  A value that does not fit
    // (un-normalized input) raises a device flag, and the wave then runs exactly the i64 scheme: a second, CONDITIONAL pre-pass writes the i64 operand
    // over the copies (its blocks return at once while the flag is down), the 16-bit form of the tail returns at once and the operand variant, launched
    // beside it and returning at once while the flag is down, does the column.  In-place calls included (both pre-passes read the input before any
    // tail writes); with the shifted store of glwe_trace too (k_inv_tail<..,RSH,7,SGN>: the one-bit shift behind the f64 chain).
    static const int b16_knob = exp_knob("POULPY_DBG_AUTO_BODY16", 1);
    // (not under the rounding-margin probe: its instantiation of the tail keeps the i64 operand - the values that are rounded are the same)
    // (add / sub forms: the operand phi(body) +- a0 is a sum of two digits - a key base of at most 15 bits; the other columns keep their 8-byte operand)
    // (glwe_trace's steps, want_rsh: their input is the previous step's - or the initial shift's - normalized output, so with a base of at most 14 bits
    //  the operand always fits and the flag-up launches of the shifted-store forms stay what they are there: never taken)
    const bool body16 = spectral_body16(c);   // (glwe_fused zeroed the flag word in front of pass 1: that kernel may raise it too, f.side16)
    (void)b16_knob;
    short* b16 = body16 ? (short*)f.res_tmp : nullptr;
    if (body16) {
        t.body16 = b16; t.body16_limbs = bl; t.body16_wide = M->wide16();
        if (f.side16 && c.au_big) t.other16 = f.side16;
    }
    if (fold) { t.body_gather = true; t.gather_mul = c.au_g; }
    else { t.body_src = (const long long*)f.res_tmp; t.body_bs = (long long)bl * n; t.body_ls = n; }
    const int cond = body16 ? 32 : 0;   // (launch_automorphism: the i64 pre-pass only if the flag is up)
    if (!c.au_big) {
        // plain form, res = phi(normalize(big)) (glwe_ct.rs:65-71): the inverse transform is phi(big) with phi's signs; the tail undoes
        // them in front of the carry chain (auto_mul) and puts them back on the digits (post_neg); only the body column has an operand
        if (body16) PZ_TRY(launch_automorphism(M, nb * bl, (const long long*)av.p, bsm, nullptr, bdm, c.au_g, 1, nullptr, PolyMap{1, 1, 0, 0, 0, 0}, b16));
        if (!fold) PZ_TRY(launch_automorphism(M, nb * bl, (const long long*)av.p, bsm, (long long*)f.res_tmp, bdm, c.au_g, 1 | cond));
        t.auto_mul = c.au_g; t.post_neg = true; t.body_only = true;
        return launch_inv_tail(M, t);
    }
    // operand of the body column, one stream: phi(body) + a0 (add) or -phi(body) + a0 (sub forms: the tail negates every operand)
    // POULPY_DBG_AUTO_BODYADD=1: the pre-pass only permutes (+-phi(body)) and the tail adds a0 from the ciphertext itself (a second operand
    // stream on the body column) instead of a pre-pass with an add operand; not with the shifted stores of glwe_trace (registers)
    static const int bodyadd_knob = exp_knob("POULPY_DBG_AUTO_BODYADD", 0);
    const bool rsh = c.want_rsh && tail_rsh_supported(M) && !c.cross_out && c.p->res_base2k <= 29;   // (32-bit shift steps: device_fft.hpp)
    // (16-bit scheme: the pre-pass writes the operand the chain adds - phi(body) + a0 (add), phi(body) - a0 (sub forms; the i64 scheme stores
    //  -phi(body) + a0 and lets the tail negate it))
    if (body16) PZ_TRY(launch_automorphism(M, nb * bl, (const long long*)av.p, bsm, nullptr, bdm, c.au_g, c.au->mode == 1 ? 1 : (1 | 16), (const long long*)av.p, bsm, b16));
    if (fold) t.gather_neg = c.au->mode != 1;
    else if (bodyadd_knob && !rsh) {
        PZ_TRY(launch_automorphism(M, nb * bl, (const long long*)av.p, bsm, (long long*)f.res_tmp, bdm, c.au_g, c.au->mode == 1 ? 1 : 3));
        t.body_add = true;
    } else PZ_TRY(launch_automorphism(M, nb * bl, (const long long*)av.p, bsm, (long long*)f.res_tmp, bdm, c.au_g, (c.au->mode == 1 ? 1 : 3) | cond,
                                      (const long long*)av.p, bsm));
    if (c.au->mode == 3) { t.auto_mul = 2u * (unsigned)n; t.auto_neg = true; }   // a - phi(big): every sign flipped
    t.small_neg = c.au->mode != 1;
    t.post_rsh = rsh;
    PZ_TRY(launch_inv_tail(M, t));
    if (rsh) *c.post_rsh = true;
    return PZ_OK;
}
// res_base2k != key_base2k: vec_znx_big_normalize(res_base2k <- key_base2k) in two exact steps - the tail's carry chain writes balanced
// key-base digits (all key_size limbs: nothing is dropped), the cross-base kernel converts them.  Both steps are functions of the torus
// value only, so the result equals the reference's single cross-base pass over the big value (checked on the oracle over thousands of
// random shapes / edge digits, and by the parity tests).
static int wave_cross_base_tail(const GlweCall& c, const FusedBufs& f, size_t b0, int nb, const DV& av) {
    const long long tmp_ct = c.n * c.s.cols_out * (long long)c.ksz;
    TailCall t = wave_tail(c, nb, f.T2, b0);
    t.res = (long long*)f.key_digits; t.res_bs = tmp_ct; t.res_size = c.ksz; t.base2k = (int)c.p->key_base2k;
    if (c.ks) tail_operand(t, av, c.tensor);
    PZ_TRY(launch_inv_tail(c.M, t));
    DV tv{f.key_digits, tmp_ct, c.s.cols_out, c.ksz}, rv{c.res_at(b0), c.res_bs, c.s.cols_out, (int)c.p->res_size};
    for (int col = 0; col < c.s.cols_out; ++col)
        PZ_TRY(dev_normalize(c.M, nb, rv, (int)c.p->res_base2k, 0, col, tv, (int)c.p->key_base2k, col));
    return PZ_OK;
}
// plain product / key switch / relinearization, and the automorphism family where the spectral form does not apply (m2 = 256 plan,
// POULPY_DBG_AUTO_SPECTRAL): the tail gathers -+phi^-1(a) (+ body) itself, writes into res_tmp, one permutation pass follows
static int wave_plain_tail(const GlweCall& c, const FusedBufs& f, size_t b0, int nb, const DV& av) {
    pz_module* M = c.M;
    if (M->dbg_stages & 4) {
        TailCall t = wave_tail(c, nb, f.T2, b0);
        if (c.au) { t.res = (long long*)f.res_tmp; t.res_bs = c.res_ct; }
        if (c.ks) tail_operand(t, av, c.au_big || c.tensor);
        if (c.a16) { t.acc32 = 4; t.small16 = c.a16 + (long long)b0 * av.size * c.n; t.small16_cs = c.a16_cs; t.small_bs = 0; }
        if (c.au_big) { t.auto_mul = c.au_p; t.gather_mul = c.au_p; t.gather_neg = c.au->mode != 1; }
        t.auto_neg = c.au && c.au->mode == 3;
        PZ_TRY(launch_inv_tail(M, t));
    }
    if (c.au) {
        PolyMap tm{(int)c.p->res_size, c.s.cols_out, c.res_ct, (long long)c.s.cols_out * c.n, c.n, 0};
        PZ_TRY(launch_automorphism(M, nb * (int)c.p->res_size * c.s.cols_out, (const long long*)f.res_tmp, tm, (long long*)c.res_at(b0), tm, c.au_g,
                                   c.au->mode == 0 ? 1 : 0));
    }
    return PZ_OK;
}

static int glwe_fused(const GlweCall& c) {
    pz_module* M = c.M;
    FusedBufs f;
    PZ_TRY(fused_carve(c, &f));
    PZ_TRY(wave_key(c, f.key_scratch, false, &f.Pp));
    const MidDigits dg = c.digits ? digit_terms(c) : MidDigits{};
    const SpectralPerm sp = spectral_perm(c);
    const bool two_kernel = n4096_two_kernel(c);
    for (size_t b0 = 0; b0 < c.batch; b0 += c.chunk) {
        const int nb = (int)std::min(c.chunk, c.batch - b0);
        DV av;
        PZ_TRY(wave_input(c, b0, nb, f.a_conv, &av));
        PolyMap sm{av.size, c.s.cols_in, av.bs, (long long)av.cols * c.n, c.n, c.n * c.s.a_col0};
        if (two_kernel) { PZ_TRY(wave_n4096_two_kernel(c, f, b0, nb, av, sm)); continue; }
        f.side16 = nullptr;
        if (sp.on && spectral_body16(c)) {
            PZ_TRY(launch_zero_bytes(M, M->margin + 1, 8));   // the wide flag (module.hpp: wide16), in front of everything that may raise it
            // add / sub forms at rank 1 (one mask column = the key switch's input): pass 1 also leaves its input as 16-bit values for the tail's other
            // column, behind the body operand's segment of res_tmp when there is room (16 res_size - 8 bl >= 2 a_size limbs' worth per ciphertext)
            static const int side_knob = exp_knob("POULPY_DBG_AUTO_SIDE16", 1);
            const int bl_ = std::min(av.size, c.ksz);
            if (side_knob && c.au_big && c.s.cols_in == 1 && c.s.cols_out == 2 && !c.digits && !c.a16 && (M->dbg_stages & 1) &&
                16 * (long long)c.p->res_size - 8 * (long long)bl_ >= 2 * (long long)av.size)
                f.side16 = (short*)((char*)f.res_tmp + (size_t)nb * bl_ * c.n * 8);
        }
        if (c.a16) {
            PolyMap s16{av.size, c.s.cols_in, (long long)av.size * c.n, c.n, c.a16_cs, c.a16_cs * c.s.a_col0};
            PZ_TRY(launch_fwd_pass1_t16(M, nb * c.npi, c.a16 + (long long)b0 * av.size * c.n, s16, f.T));
        } else if (f.side16) {
            PZ_TRY(launch_fwd_pass1_w16(M, nb * c.npi, (const long long*)av.p, sm, f.T, f.side16));
        } else if (M->dbg_stages & 1) {
            PZ_TRY(launch_fwd_pass1(M, nb * c.npi, (const long long*)av.p, sm, f.T, true));
        }
        if (c.digits && dg.n == 0) {   // nothing reaches the product (e.g. dsize > a.size): the big value is the body alone
            PZ_TRY(launch_zero_bytes(M, f.T2, (size_t)nb * c.npo * M->m * sizeof(cplx)));
        } else if (M->dbg_stages & 2) {
            PZ_TRY(launch_mid(M, nb, f.T, f.T2, f.Pp, c.npi, c.npo, c.nrows, c.ncols, f.mid_dummy, sp.mul, sp.add, c.digits ? &dg : nullptr, nullptr,
                              sp.conj));
        }
        if (sp.on) PZ_TRY(wave_spectral_tail(c, f, b0, nb, av));
        else if (c.cross_out) PZ_TRY(wave_cross_base_tail(c, f, b0, nb, av));
        else PZ_TRY(wave_plain_tail(c, f, b0, nb, av));
    }
    return PZ_OK;
}

// ------------------------------------------------------------------------------
// N = 1024 / 2048: no pipeline plan (their per-op split is 16 x 32 / 32 x 32), but whole polynomials fit LDS: the two-kernel pipeline of
// device_small.hpp with its own m = M1 x 128 tables.  Mixed bases as in the fused pipeline: `a` re-expressed in the key's base first; a
// result in another base = balanced key-base digits from the inverse kernel (all key limbs), then one cross-base pass.  The automorphism
// family rides along: phi is an index / sign map inside the inverse kernel's carry-chain stage.
// ------------------------------------------------------------------------------
static int glwe_small_ring(const GlweCall& c) {
    pz_module* M = c.M;
    const SmallWs w = small_ws(M, c.p, c.s, c.chunk);
    PZ_TRY(ws_reserve(M, w.total));
    char* base = (char*)M->ws;
    cplx* key_scratch; cplx* S; int64_t* a_conv; int64_t* key_digits;
    PZ_TRY(ws_take(M, base, w.key, &key_scratch));
    PZ_TRY(ws_take(M, base, w.spectra, &S));
    PZ_TRY(ws_take(M, base, w.conv, &a_conv));
    PZ_TRY(ws_take(M, base, w.digits, &key_digits));
    const cplx* Pp;
    PZ_TRY(wave_key(c, key_scratch, true, &Pp));
    const bool rsh = c.want_rsh && c.au && c.au->mode != 0 && !c.cross_out && c.p->res_base2k <= 29;
    for (size_t b0 = 0; b0 < c.batch; b0 += c.chunk) {
        const int nb = (int)std::min(c.chunk, c.batch - b0);
        DV av;
        PZ_TRY(wave_input(c, b0, nb, a_conv, &av));
        PolyMap sm{av.size, c.s.cols_in, av.bs, (long long)av.cols * c.n, c.n, c.n * c.s.a_col0};
        const long long* body = c.ks ? (const long long*)av.p : nullptr;
        // plain product / key switch of a rank-1 ciphertext: one kernel, the spectra never leave the CU (round 6, device_small_one.hpp)
        if (!c.au && !c.cross_out && small_one_supported(M, c.npi, c.nrows, c.ncols, c.s.cols_out, c.ksz, nb)) {
            PZ_TRY(launch_small_one(M, nb, (const long long*)av.p, sm, Pp, c.npi, c.nrows, c.ncols, c.ksz, (long long*)c.res_at(b0), c.res_bs, c.s.cols_out,
                                    (int)c.p->res_size, body, av.bs, c.s.cols_a, av.size, (int)c.p->res_base2k, c.body_col));
            continue;
        }
        PZ_TRY(launch_small_fwd(M, nb * c.npi, (const long long*)av.p, sm, S));
        if (c.cross_out) {
            const long long tmp_ct = c.n * c.s.cols_out * (long long)c.ksz;
            PZ_TRY(launch_small_inv(M, nb, S, Pp, c.npi, c.nrows, c.ncols, c.s.cols_out, c.ksz, (long long*)key_digits, tmp_ct, c.s.cols_out, c.ksz,
                                    body, av.bs, c.s.cols_a, av.size, (int)c.p->key_base2k, c.body_col));
            DV tv{key_digits, tmp_ct, c.s.cols_out, c.ksz}, rv{c.res_at(b0), c.res_bs, c.s.cols_out, (int)c.p->res_size};
            for (int col = 0; col < c.s.cols_out; ++col)
                PZ_TRY(dev_normalize(M, nb, rv, (int)c.p->res_base2k, 0, col, tv, (int)c.p->key_base2k, col));
            continue;
        }
        PZ_TRY(launch_small_inv(M, nb, S, Pp, c.npi, c.nrows, c.ncols, c.s.cols_out, c.ksz, (long long*)c.res_at(b0), c.res_bs, c.s.cols_out,
                                (int)c.p->res_size, body, av.bs, c.s.cols_a, av.size, (int)c.p->res_base2k, c.body_col, false, nullptr, 0,
                                c.au != nullptr, c.au_p, c.au ? c.au->mode : 0, rsh));
    }
    if (rsh) *c.post_rsh = true;
    return PZ_OK;
}

// ------------------------------------------------------------------------------
// five-kernel path: the reference's op sequence, one batched kernel per HAL op (every shape; what the fused pipelines are tested against)
// ------------------------------------------------------------------------------
struct UnfusedBufs {
    int64_t* a_conv; double* a_dft; double* res_dft; double* tmp_dft; cplx* T; int64_t* res_tmp;
};
// a_dft <- DFT of the input limbs, res_dft <- VMP; returns the limbs of res_dft that carry the result
static int wave_unfused_product(const GlweCall& c, const UnfusedBufs& u, int nb, const DV& av, DV rd, int* res_dft_size) {
    pz_module* M = c.M;
    const long long n = c.n;
    const int a_size = av.size, a_col0 = c.s.a_col0;   // key-switch transforms the mask columns only (keyswitching/glwe.rs:231-234)
    *res_dft_size = c.ksz;
    if (c.dsize == 1) {
        DV ad{u.a_dft, n * c.s.cols_in * a_size, c.s.cols_in, a_size};
        PZ_TRY(dev_dft_apply(M, nb, 1, 0, ad, 0, av, a_col0, c.s.cols_in, nullptr, u.T));
        return dev_vmp(M, nb, rd, ad, c.pmat, c.dnum, c.s.cols_in, c.s.cols_out, c.ksz, 0);
    }
    // external_product/glwe.rs:235-267 ; keyswitching/glwe.rs:332-379
    // res_dft starts zeroed (glwe.rs:122): limbs skipped by the first iterations are only ever added to
    PZ_TRY(launch_zero_bytes(M, u.res_dft, (size_t)nb * rd.bs * 8));
    DV td{u.tmp_dft, n * c.s.cols_out * c.ksz, c.s.cols_out, c.ksz};
    for (int di = 0; di < c.dsize; ++di) {
        int a_sz = (a_size + di) / c.dsize;
        if (c.ks) a_sz = std::min(a_sz, c.dnum);
        const int drop = std::max(c.dsize - di - 2, 0);
        *res_dft_size = c.ksz - drop;
        DV ad{u.a_dft, n * c.s.cols_in * a_sz, c.s.cols_in, a_sz};
        PZ_TRY(dev_dft_apply(M, nb, c.dsize, c.dsize - 1 - di, ad, 0, av, a_col0, c.s.cols_in, nullptr, u.T));
        DV rdi{u.res_dft, rd.bs, c.s.cols_out, *res_dft_size};
        if (di == 0) {
            PZ_TRY(dev_vmp(M, nb, rdi, ad, c.pmat, c.dnum, c.s.cols_in, c.s.cols_out, c.ksz, 0));
        } else {
            DV tdi{u.tmp_dft, td.bs, c.s.cols_out, *res_dft_size};
            PZ_TRY(dev_vmp(M, nb, tdi, ad, c.pmat, c.dnum, c.s.cols_in, c.s.cols_out, c.ksz, di));
            PZ_TRY(launch_ew(M, EW_ADD, u.res_dft, rd.bs, n, u.res_dft, rd.bs, n, u.tmp_dft, td.bs, n, c.s.cols_out * *res_dft_size, nb));
        }
    }
    // keyswitching/glwe.rs:378 res.set_size(res.max_size()): limbs dropped by the last iterations keep the value of the earlier ones
    if (c.ks) *res_dft_size = c.ksz;
    return PZ_OK;
}
// automorphism family, op by op as the reference: big value, body, [automorphism of the big value, +- a], normalize, [automorphism]
static int wave_unfused_auto(const GlweCall& c, const UnfusedBufs& u, int nb, const DV& av, DV rb, DV rv) {
    pz_module* M = c.M;
    const long long n = c.n;
    const int a_size = av.size, L = rb.size;
    PZ_TRY(dev_idft(M, nb, rb, 0, rb, 0, c.s.cols_out, L, u.T));
    const long long big_ls = (long long)c.s.cols_out * n, a_ls = (long long)av.cols * n;
    PZ_TRY(launch_ew(M, EW_ADD_I64, u.res_dft, rb.bs, big_ls, u.res_dft, rb.bs, big_ls, av.p, av.bs, a_ls, std::min(L, a_size), nb));
    DV nsrc = rb;
    if (c.au_big) {
        int64_t* big2 = (int64_t*)u.T;  // free again: same bytes as the big value
        PolyMap bm{L, c.s.cols_out, rb.bs, big_ls, n, 0};
        PZ_TRY(launch_automorphism(M, nb * L * c.s.cols_out, (const long long*)u.res_dft, bm, (long long*)big2, bm, c.au_g, 1));
        const int sum = std::min(L, a_size);
        for (int col = 0; col < c.s.cols_out; ++col) {
            int64_t* bc = big2 + (long long)col * n;
            const int64_t* ac = (const int64_t*)av.p + (long long)col * n;
            if (c.au->mode == 1) PZ_TRY(launch_ew(M, EW_ADD_I64, bc, rb.bs, big_ls, bc, rb.bs, big_ls, ac, av.bs, a_ls, sum, nb));
            else if (c.au->mode == 2) PZ_TRY(launch_ew(M, EW_SUB_I64, bc, rb.bs, big_ls, bc, rb.bs, big_ls, ac, av.bs, a_ls, sum, nb));
            else {  // a - big, and -big where a has no limb (vec_znx/sub.rs:84-110)
                PZ_TRY(launch_ew(M, EW_SUB_I64, bc, rb.bs, big_ls, ac, av.bs, a_ls, bc, rb.bs, big_ls, sum, nb));
                PZ_TRY(launch_ew(M, EW_NEG_I64, bc + (long long)sum * big_ls, rb.bs, big_ls, bc + (long long)sum * big_ls, rb.bs, big_ls,
                                 nullptr, 0, 0, L - sum, nb));
            }
        }
        nsrc = DV{big2, rb.bs, c.s.cols_out, L};
    }
    DV nd = c.au->mode == 0 ? DV{u.res_tmp, c.res_ct, c.s.cols_out, (int)c.p->res_size} : rv;
    for (int col = 0; col < c.s.cols_out; ++col)
        PZ_TRY(dev_normalize(M, nb, nd, (int)c.p->res_base2k, 0, col, nsrc, (int)c.p->key_base2k, col));
    if (c.au->mode == 0) {
        PolyMap tm{(int)c.p->res_size, c.s.cols_out, c.res_ct, (long long)c.s.cols_out * n, n, 0};
        PZ_TRY(launch_automorphism(M, nb * (int)c.p->res_size * c.s.cols_out, (const long long*)u.res_tmp, tm, (long long*)rv.p, tm, c.au_g, 1));
    }
    return PZ_OK;
}
static int glwe_unfused(const GlweCall& c) {
    pz_module* M = c.M;
    const long long n = c.n;
    const OpWs w = op_ws(M, c.p, c.s, c.chunk, c.ks, c.au != nullptr);
    PZ_TRY(ws_reserve(M, w.total));
    char* base = (char*)M->ws;
    UnfusedBufs u;
    PZ_TRY(ws_take(M, base, w.a_conv, &u.a_conv));
    PZ_TRY(ws_take(M, base, w.a_dft, &u.a_dft));
    PZ_TRY(ws_take(M, base, w.res_dft, &u.res_dft));
    PZ_TRY(ws_take(M, base, w.tmp_dft, &u.tmp_dft));
    PZ_TRY(ws_take(M, base, w.T, &u.T));
    PZ_TRY(ws_take(M, base, w.res_tmp, &u.res_tmp));
    for (size_t b0 = 0; b0 < c.batch; b0 += c.chunk) {
        const int nb = (int)std::min(c.chunk, c.batch - b0);
        const DV raw_av{(void*)(c.a + (long long)b0 * c.a_bs), c.a_bs, c.s.cols_a, (int)c.p->a_size};
        DV av;
        PZ_TRY(wave_input(c, b0, nb, u.a_conv, &av));
        DV rd{u.res_dft, n * c.s.cols_out * c.ksz, c.s.cols_out, c.ksz};
        int res_dft_size = c.ksz;
        PZ_TRY(wave_unfused_product(c, u, nb, av, rd, &res_dft_size));
        DV rb{u.res_dft, rd.bs, c.s.cols_out, res_dft_size};
        DV rv{(void*)c.res_at(b0), c.res_bs, c.s.cols_out, (int)c.p->res_size};
        if (c.au) {
            PZ_TRY(wave_unfused_auto(c, u, nb, av, rb, rv));
        } else if (c.p->res_base2k == c.p->key_base2k && M->fuse_tail && tail_supported(M)) {
            // inverse pass 2, then the fused tail: inverse pass 1 + body add + carry chain, no VecZnxBig in HBM
            PolyMap sm{res_dft_size, c.s.cols_out, rb.bs, (long long)c.s.cols_out * n, n, 0};
            PZ_TRY(launch_inv_pass2(M, nb * res_dft_size * c.s.cols_out, u.res_dft, sm, u.T));
            // (tensor: every column receives its operand; with a conversion the reference still adds the UN-normalized a when
            //  res_base2k == key_base2k, operations/glwe.rs:588-592)
            TailCall t;
            t.batch = nb; t.T = u.T; t.rowmajor = false; t.nlimbs = res_dft_size; t.ncols = c.s.cols_out;
            t.res = (long long*)rv.p; t.res_bs = rv.bs; t.res_cols = rv.cols; t.res_size = rv.size; t.base2k = (int)c.p->res_base2k;
            t.body_col = c.body_col;
            if (c.ks) tail_operand(t, c.tensor ? raw_av : av, c.tensor);
            PZ_TRY(launch_inv_tail(M, t));
        } else {
            PZ_TRY(dev_idft(M, nb, rb, 0, rb, 0, c.s.cols_out, res_dft_size, u.T));
            const long long big_ls = (long long)c.s.cols_out * n;
            if (c.tensor) {  // operations/glwe.rs:588-598: + a[col] on every column (raw a when res_base2k == key_base2k, else the converted one)
                const DV& sv = c.p->res_base2k == c.p->key_base2k ? raw_av : av;
                for (int col = 0; col < c.s.cols_out; ++col)
                    PZ_TRY(launch_ew(M, EW_ADD_I64, u.res_dft + (long long)col * n, rb.bs, big_ls, u.res_dft + (long long)col * n, rb.bs, big_ls,
                                     (const int64_t*)sv.p + (long long)col * n, sv.bs, (long long)sv.cols * n, std::min(res_dft_size, sv.size), nb));
            } else if (c.ks) {  // body column added after the inverse transform (keyswitching/glwe.rs:237)
                PZ_TRY(launch_ew(M, EW_ADD_I64, u.res_dft + (long long)c.body_col * n, rb.bs, big_ls, u.res_dft + (long long)c.body_col * n, rb.bs,
                                 big_ls, av.p, av.bs, (long long)av.cols * n, std::min(res_dft_size, av.size), nb));
            }
            for (int col = 0; col < c.s.cols_out; ++col)
                PZ_TRY(dev_normalize(M, nb, rv, (int)c.p->res_base2k, 0, col, rb, (int)c.p->key_base2k, col));
        }
    }
    return PZ_OK;
}

// ------------------------------------------------------------------------------
// glwe_op
// ------------------------------------------------------------------------------
// Automorphism family on top of the key switch (poulpy-core automorphism/glwe_ct.rs:51-275).  With phi = X -> X^p:
//   mode 0  res = phi(normalize(big))                       (:65-71)
//   mode 1  res = normalize(phi(big) + a)   (add, :133-138)   2: phi(big) - a (:222-227)   3: a - phi(big) (:268-273)
// where big is the key-switch value including the body (keyswitching/glwe.rs:236-237).  Normalization acts per
// coefficient, so modes 1-3 are computed as  phi(normalize'(s .* (big + small)))  with small = -+phi^-1(a) (+ body), s(n) the sign phi
// gives coefficient n (applied inside the tail before the carry chain; flipped for mode 3); mode 0 is the plain key switch followed by
// the signed permutation - or, in the spectral form, all of it inside the three kernels (spectral_perm above).
// (Round 2 experiment, removed — git history has it: a CU-partitioned, overlapped form of the fused pipeline.  With a CU mask spread
//  over the 8 XCDs pass 1 and the tail keep their full rate down to 64 CUs while the middle kernel scales with its CU count
//  (profiles/r02_cu_mask_scaling.txt), so chunk c+1's pass 1, chunk c's middle kernel and chunk c-1's tail were run concurrently on
//  disjoint CU sets, chained by events.  Bit-exact, but slower in every split tried (best 73 500/s against 88 700/s back to back,
//  profiles/r02_overlap_sweep.txt): under concurrency the three kernels share HBM at ~4.7 TB/s aggregate.)
static int glwe_call_init(GlweCall& c, pz_module* M, bool ks, int64_t* res, const int64_t* a, const double* pmat, const pz_glwe_op_params* p,
                          size_t batch, const AutoSpec* au, const OpLayout* lay, bool tensor, bool* post_rsh) {
    c.want_rsh = post_rsh && *post_rsh;
    c.post_rsh = post_rsh;
    if (post_rsh) *post_rsh = false;
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(p->dsize >= 1 && p->dnum >= 1 && p->key_size >= 1 && p->a_size >= 1 && p->res_size >= 1, "glwe op: empty shape");
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(a) && is_device_ptr(pmat), "batched entry points take device pointers");
    PZ_REQUIRE(!(tensor && (au || lay)), "glwe_tensor_relinearize: packed tensors, no automorphism");
    if (tensor) ks = true;   // the product is gglwe_product_dft, as for a key switch
    c.M = M; c.p = p; c.ks = ks; c.tensor = tensor; c.au = au; c.lay = lay; c.res = res; c.a = a; c.pmat = pmat; c.batch = batch;
    c.s = op_shape(p, ks, tensor);
    c.chunk = pick_chunk(M, p, c.s, std::max<size_t>(batch, 1));
    c.n = (long long)M->n;
    c.dsize = (int)p->dsize; c.dnum = (int)p->dnum; c.ksz = (int)p->key_size;
    c.a_ct = c.n * c.s.cols_a * (long long)p->a_size;
    c.res_ct = c.n * c.s.cols_out * (long long)p->res_size;
    c.npi = c.s.cols_in * c.s.a_size_eff; c.npo = c.s.cols_out * c.ksz;
    c.nrows = c.dnum * c.s.cols_in; c.ncols = c.s.cols_out * c.ksz;
    c.au_big = au && au->mode != 0;
    c.au_p = au ? (unsigned)((unsigned long long)au->p & (2ull * (unsigned long long)c.n - 1ull)) : 0u;
    c.au_g = au ? inv_mod_2n(au->p, c.n) : 0u;
    c.a_bs = lay ? lay->a_stride : c.a_ct; c.res_bs = lay ? lay->res_stride : c.res_ct;
    c.body_col = lay ? lay->body_col : 0;
    c.digits = c.dsize > 1; c.cross_out = p->res_base2k != p->key_base2k;
    PZ_REQUIRE(!(au && lay), "glwe_automorphism: packed ciphertexts only");
    PZ_REQUIRE(c.body_col >= 0 && c.body_col < c.s.cols_out, "body column out of range");
    if (au) {
        PZ_REQUIRE(ks && c.s.cols_a == c.s.cols_out, "glwe_automorphism: the key must map rank -> rank");
        PZ_REQUIRE((au->p & 1) != 0, "glwe_automorphism: the Galois element must be odd");
        PZ_REQUIRE(au->mode >= 0 && au->mode <= 3, "glwe_automorphism: unknown mode");
    }
    return PZ_OK;
}

extern "C" {

int glwe_op(pz_module* M, bool ks, int64_t* res, const int64_t* a, const double* pmat, const pz_glwe_op_params* p, size_t batch,
            const AutoSpec* au, const OpLayout* lay, bool tensor, bool* post_rsh) {
    GlweCall c;
    PZ_TRY(glwe_call_init(c, M, ks, res, a, pmat, p, batch, au, lay, tensor, post_rsh));
    if (batch == 0) return PZ_OK;
    // dsize > 1 (digit-selected product inside the middle kernel) and res_base2k != key_base2k (the tail normalizes into the key's base,
    // one cross-base pass follows) ride on the fused pipeline too; both need the 128-point-row plans and no automorphism (fused_applies)
    if (fused_applies(M, p, c.s, c.tensor, au != nullptr)) return glwe_fused(c);
    if (small_ring_applies(M, p, c.s, c.ks, c.tensor, au != nullptr, lay == nullptr)) return glwe_small_ring(c);
    return glwe_unfused(c);
}

// glwe_tensor_relinearize on a GLWETensor kept as 16-bit digits (the fused multiply + relinearize, api_cnv.hip): forward pass 1 reads the pair
// columns, the tail adds the first rank + 1 columns - both from a16 (GlweCall::a16)
bool glwe_relin_t16_supported(const pz_module* M, const pz_glwe_op_params* p) {
    if (!p || p->dsize != 1 || p->a_base2k != p->key_base2k || p->res_base2k != p->key_base2k || p->rank_out != p->rank || p->rank < 1) return false;
    const OpShape s = op_shape(p, true, true);
    return fused_applies(M, p, s, true, false) && tail_d16_only_supported(M) && M->dbg_stages == 7;
}
size_t glwe_relin_chunk(const pz_module* M, const pz_glwe_op_params* p, size_t batch) { return pick_chunk(M, p, op_shape(p, true, true), std::max<size_t>(batch, 1)); }
int glwe_relin_t16(pz_module* M, int64_t* res, const short* a16, long long a16_cs, const double* pmat, const pz_glwe_op_params* p, size_t batch) {
    GlweCall c;
    PZ_TRY(glwe_call_init(c, M, true, res, reinterpret_cast<const int64_t*>(a16), pmat, p, batch, nullptr, nullptr, true, nullptr));
    PZ_REQUIRE(glwe_relin_t16_supported(M, p) && !n4096_two_kernel(c), "glwe_relin_t16: the pipeline path only");
    if (batch == 0) return PZ_OK;
    c.a16 = a16; c.a16_cs = a16_cs;
    return glwe_fused(c);
}

// The four GLWE-level entry points accept device pointers (batched, device-resident: the measured path) or HOST containers
// (what a CoreImpl override of the Rust shim passes): host ciphertexts are staged, a host-resident prepared key is mirrored on
// the device (resolve_key); the call is then logically synchronous like every host-pointer call.
// Pinned (page-locked: pz_alloc_bytes, hipHostMalloc, hipHostRegister) host memory - the only kind a copy engine reads and writes on its own
static bool is_pinned_host(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}
// Host containers, several ciphertexts per call, pinned memory (round 5): the call as waves of ciphertexts on three streams - wave k + 1 travels to
// the device (stream2) while wave k runs on the module stream and wave k - 1 travels back (stream_out).  PCIe is full duplex: the serial form
// (everything up, kernels, everything down) used one direction at a time - 3 200 external products/s at 16 ciphertexts per call at the metric
// shape, 53 GB/s summed over both directions.  Logically synchronous like every host-pointer call: returns when the last wave is back.
static int glwe_entry_duplex(pz_module* M, bool ks, bool tensor, int64_t* res, const int64_t* a, const double* key, const pz_glwe_op_params* p,
                             size_t batch, const AutoSpec* au, size_t res_ct_bytes, size_t a_ct_bytes) {
    if (!M->stream2) {
        SideStream probe(M);   // (creates the side stream and its events)
        PZ_TRY(probe.fork());
    }
    if (!M->stream_out && hipStreamCreateWithFlags(&M->stream_out, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return fail(PZ_ERR_HIP, "stream create failed"); }
    const bool in_place = (const void*)res == (const void*)a;
    void *a_dev = nullptr, *r_dev = nullptr;
    PZ_TRY(arena_alloc(M, batch * a_ct_bytes, &a_dev));
    if (in_place) r_dev = a_dev; else PZ_TRY(arena_alloc(M, batch * res_ct_bytes, &r_dev));
    // 8 equal waves (measured at 16 ciphertexts per call at the metric shape: 4 waves 4 700/s, 8 waves 4 960/s, 16 waves 4 850/s; a half-size first
    // and last wave - shorter fill and drain, one wave more - 4 610/s; profiles/r05_host_duplex.txt)
    const size_t per = (batch + 7) / 8;
    std::vector<size_t> waves;
    for (size_t left = batch; left > 0; left -= waves.back()) waves.push_back(std::min(per, left));
    std::vector<hipEvent_t> up, done;
    auto ev = [&](std::vector<hipEvent_t>& v) -> hipEvent_t {
        hipEvent_t e = KTimer::get(M);
        if (e) v.push_back(e);
        return e;
    };
    int rc = PZ_OK;
    // (no HIP-graph capture of the waves' launch sequences: one graph per wave - keyed by its pointers - would fill the module's 16-entry cache,
    //  and instantiating them costs more than the three launches per wave they would save)
    const bool graphs_were_on = M->graphs_on;
    M->graphs_on = false;
    // everything issued on the module stream so far (key mirror upload, earlier calls) precedes the first copies
    hipEvent_t e0 = ev(up);
    if (!e0 || hipEventRecord(e0, M->stream) != hipSuccess || hipStreamWaitEvent(M->stream2, e0, 0) != hipSuccess ||
        hipStreamWaitEvent(M->stream_out, e0, 0) != hipSuccess) rc = fail(PZ_ERR_HIP, "duplex host path: event set-up failed");
    size_t b0 = 0;
    for (size_t wi = 0; rc == PZ_OK && wi < waves.size(); b0 += waves[wi], ++wi) {
        const size_t nb = waves[wi];
        hipEvent_t eu = ev(up), ed = ev(done);
        if (!eu || !ed) { rc = fail(PZ_ERR_HIP, "duplex host path: no event"); break; }
        char* ad = (char*)a_dev + b0 * a_ct_bytes;
        char* rd = (char*)r_dev + b0 * res_ct_bytes;
        if (hipMemcpyAsync(ad, (const char*)a + b0 * a_ct_bytes, nb * a_ct_bytes, hipMemcpyHostToDevice, M->stream2) != hipSuccess ||
            hipEventRecord(eu, M->stream2) != hipSuccess || hipStreamWaitEvent(M->stream, eu, 0) != hipSuccess) { rc = fail(PZ_ERR_HIP, "duplex host path: upload failed"); break; }
        rc = glwe_op(M, ks, (int64_t*)rd, (const int64_t*)ad, key, p, nb, au, nullptr, tensor);
        if (rc != PZ_OK) break;
        if (hipEventRecord(ed, M->stream) != hipSuccess || hipStreamWaitEvent(M->stream_out, ed, 0) != hipSuccess ||
            hipMemcpyAsync((char*)res + b0 * res_ct_bytes, rd, nb * res_ct_bytes, hipMemcpyDeviceToHost, M->stream_out) != hipSuccess) { rc = fail(PZ_ERR_HIP, "duplex host path: download failed"); break; }
    }
    M->graphs_on = graphs_were_on;
    // the call returns when every wave is back (and nothing of it is still in flight on an error path)
    const hipError_t s1 = hipStreamSynchronize(M->stream2), s2 = hipStreamSynchronize(M->stream), s3 = hipStreamSynchronize(M->stream_out);
    for (hipEvent_t e : up) M->event_pool.push_back(e);
    for (hipEvent_t e : done) M->event_pool.push_back(e);
    if (rc == PZ_OK && (s1 != hipSuccess || s2 != hipSuccess || s3 != hipSuccess)) { (void)hipGetLastError(); rc = fail(PZ_ERR_HIP, "duplex host path: a stream failed"); }
    return rc;
}
static int glwe_entry(pz_module* M, bool ks, bool tensor, int64_t* res, const int64_t* a, const double* pmat, const pz_glwe_op_params* p,
                      size_t batch, const AutoSpec* au) {
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(p->dsize >= 1 && p->dnum >= 1 && p->key_size >= 1 && p->a_size >= 1 && p->res_size >= 1, "glwe op: empty shape");
    const OpShape s = op_shape(p, ks || tensor, tensor);
    const size_t n8 = (size_t)M->n * 8;
    // in place (*_assign forms): one layout for a and res - checked HERE, in front of both host paths (the duplex path below does not go through
    // glwe_args_in: with res == a and a larger res every wave's kernels would write past the a-sized arena block).  Host ranges that overlap
    // without being equal take the serial path: there the whole input is on the device before the first result travels back
    const size_t res_ct_bytes = n8 * s.cols_out * p->res_size, a_ct_bytes = n8 * s.cols_a * p->a_size;
    if (res != nullptr && (const void*)res == (const void*)a) PZ_REQUIRE(res_ct_bytes == a_ct_bytes, "in-place call with different layouts for a and res");
    const bool partial_overlap = res != nullptr && a != nullptr && (const void*)res != (const void*)a &&
                                 (const char*)res < (const char*)a + batch * a_ct_bytes && (const char*)a < (const char*)res + batch * res_ct_bytes;
    if (batch >= 2 && res != nullptr && a != nullptr && pmat != nullptr && !partial_overlap && !M->timing && !canary_mode() && is_pinned_host(a) && is_pinned_host(res)) {
        const double* key = nullptr;
        PZ_TRY(resolve_key(M, pmat, n8 * p->dnum * s.cols_in * s.cols_out * p->key_size, &key));
        return glwe_entry_duplex(M, ks, tensor, res, a, key, p, batch, au, res_ct_bytes, a_ct_bytes);
    }
    GlweArgs g;
    PZ_TRY(glwe_args_in(M, g, res, a, pmat, batch * res_ct_bytes, batch * a_ct_bytes, n8 * p->dnum * s.cols_in * s.cols_out * p->key_size));
    PZ_TRY(glwe_op(M, ks, g.res, g.a, g.key, p, batch, au, nullptr, tensor));
    return glwe_args_out(M, g);
}
int pz_glwe_external_product_batched(pz_module* M, int64_t* res, const int64_t* a, const double* ggsw_pmat,
                                     const pz_glwe_op_params* p, size_t batch) {
    PZ_ENTER(M);
    return glwe_entry(M, false, false, res, a, ggsw_pmat, p, batch, nullptr);
}
int pz_glwe_keyswitch_batched(pz_module* M, int64_t* res, const int64_t* a, const double* key_pmat, const pz_glwe_op_params* p,
                              size_t batch) {
    PZ_ENTER(M);
    return glwe_entry(M, true, false, res, a, key_pmat, p, batch, nullptr);
}
int pz_glwe_automorphism_batched(pz_module* M, int64_t* res, const int64_t* a, const double* key_pmat, const pz_glwe_op_params* p,
                                 int64_t gal, int mode, size_t batch) {
    PZ_ENTER(M);
    AutoSpec au{(long long)gal, mode};
    return glwe_entry(M, true, false, res, a, key_pmat, p, batch, &au);
}
// glwe_tensor_relinearize (poulpy-core/src/operations/glwe.rs:541-607) on `batch` GLWETensors sharing one prepared tensor key
int pz_glwe_tensor_relinearize_batched(pz_module* M, int64_t* res, const int64_t* a, const double* tsk_pmat, const pz_glwe_op_params* p,
                                       size_t batch) {
    PZ_ENTER(M);
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(p->rank >= 1 && p->rank_out == p->rank, "glwe_tensor_relinearize: the tensor key maps rank (rank + 1) / 2 -> rank");
    return glwe_entry(M, true, true, res, a, tsk_pmat, p, batch, nullptr);
}
// ggsw_external_product (external_product/ggsw.rs:54-58): every (row, column) entry of the GGSW `a` is a GLWE and the entries
// are contiguous in the MatZnx layout, so the operation is one batched external product over dnum_a * (rank+1) ciphertexts
int pz_ggsw_external_product(pz_module* M, int64_t* res, const int64_t* a, size_t a_dnum, const double* ggsw_pmat,
                             const pz_glwe_op_params* p) {
    PZ_ENTER(M);
    PZ_REQUIRE(p != nullptr, "null params");
    return glwe_op(M, false, res, a, ggsw_pmat, p, a_dnum * (p->rank + 1));
}

// ggsw_expand_row (conversion/gglwe_to_ggsw.rs:116-268): column `col` >= 1 of every row is the key switch of the mask of
// res.at(row, 0) by tsk.at(col - 1), with the body of res.at(row, 0) added to column `col` of the big value before the
// normalization.  The entries (row, 0) of `count` contiguous GGSWs are `count * dnum` ciphertexts at a fixed stride, so
// each column is one batched key switch; column 0 is left untouched.
int ggsw_expand_row(pz_module* M, int64_t* ggsw, size_t dnum, const double* const* tsk_pmat, const pz_glwe_op_params* p, size_t count) {
    PZ_REQUIRE(p != nullptr && tsk_pmat != nullptr, "null params");
    PZ_REQUIRE(p->a_size == p->res_size && p->a_base2k == p->res_base2k, "ggsw_expand_row: a and res describe the same GGSW");
    PZ_REQUIRE(dnum >= 1, "ggsw_expand_row: empty GGSW");
    const size_t cols = p->rank + 1;
    const long long ct = (long long)M->n * (long long)cols * (long long)p->res_size;
    for (size_t col = 1; col < cols; ++col) {
        PZ_REQUIRE(tsk_pmat[col - 1] != nullptr, "ggsw_expand_row: null tensor key");
        OpLayout lay{ct * (long long)cols, ct * (long long)cols, (int)col};
        PZ_TRY(glwe_op(M, true, ggsw + (long long)col * ct, ggsw, tsk_pmat[col - 1], p, count * dnum, nullptr, &lay));
    }
    return PZ_OK;
}
int pz_ggsw_expand_row_batched(pz_module* M, int64_t* ggsw, size_t dnum, const double* const* tsk_pmat, const pz_glwe_op_params* p,
                               size_t count) {
    PZ_ENTER(M);
    return ggsw_expand_row(M, ggsw, dnum, tsk_pmat, p, count);
}

// ggsw_from_gglwe (conversion/gglwe_to_ggsw.rs:32-61): entries (row, 0) of the GGSW are copies of the entries (row, 0) of
// the GGLWE `a` (glwe_copy), then ggsw_expand_row.  `count` contiguous GGLWEs -> `count` contiguous GGSWs, one strided copy.
int pz_ggsw_from_gglwe_batched(pz_module* M, int64_t* ggsw, const int64_t* a, size_t a_cols_in, size_t dnum,
                               const double* const* tsk_pmat, const pz_glwe_op_params* p, size_t count) {
    PZ_ENTER(M);
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(is_device_ptr(ggsw) && is_device_ptr(a), "batched entry points take device pointers");
    PZ_REQUIRE(a_cols_in >= 1 && dnum >= 1, "ggsw_from_gglwe: empty GGLWE");
    PZ_REQUIRE((const void*)ggsw != (const void*)a, "ggsw_from_gglwe: res must not alias a");
    const size_t cols = p->rank + 1;
    const long long n = (long long)M->n, ct = n * (long long)cols * (long long)p->res_size;
    PZ_TRY(launch_ew(M, EW_COPY, ggsw, (long long)cols * ct, n, a, (long long)a_cols_in * ct, n, nullptr, 0, 0, (int)(cols * p->res_size),
                     (int)(count * dnum)));
    return ggsw_expand_row(M, ggsw, dnum, tsk_pmat, p, count);  // (the module lock is not recursive)
}

// glwe_trace_assign (poulpy-core/src/glwe_trace.rs:129-176) on `batch` ciphertexts:
//   for every step s:  res = rsh(res, 1 bit) on every column (operations/glwe.rs:1096-1112);  res = glwe_automorphism_add_assign(res, key_s)
// res in another base than the keys (:153-163; test_suite/trace.rs:36-39): (a_size, a_base2k = key_base2k) describe res re-expressed in
// the keys' base (a_size = ceil(res.max_k / key_base2k)); normalize into a temporary of that layout, trace there, normalize back.
int glwe_trace(pz_module* M, int64_t* res, size_t nsteps, const int64_t* gals, const double* const* key_pmats,
                      const pz_glwe_op_params* p, size_t batch) {
    PZ_REQUIRE(p != nullptr && (nsteps == 0 || (gals != nullptr && key_pmats != nullptr)), "glwe_trace: null argument");
    PZ_REQUIRE(p->rank_out == p->rank, "glwe_trace: rank_out != rank");
    PZ_REQUIRE(is_device_ptr(res), "batched entry points take device pointers");
    if (p->res_base2k != p->key_base2k) {
        PZ_REQUIRE(p->a_base2k == p->key_base2k && p->a_size >= 1 && p->res_size >= 1,
                   "glwe_trace: with res in another base than the keys, (a_size, a_base2k) is its layout in the keys' base");
        if (batch == 0) return PZ_OK;
        const long long n = (long long)M->n;
        const int cols = (int)p->rank + 1, B = (int)batch;
        const long long ct_c = n * cols * (long long)p->a_size, ct_r = n * cols * (long long)p->res_size;
        PZ_TRY(ws2_reserve(M, (size_t)B * ct_c * 8));
        int64_t* conv = (int64_t*)M->ws2;
        DV cv{conv, ct_c, cols, (int)p->a_size}, rv{res, ct_r, cols, (int)p->res_size};
        for (int c = 0; c < cols; ++c) PZ_TRY(dev_normalize(M, B, cv, (int)p->key_base2k, 0, c, rv, (int)p->res_base2k, c));
        pz_glwe_op_params q = *p;
        q.res_size = p->a_size; q.res_base2k = p->key_base2k;
        PZ_TRY(glwe_trace(M, conv, nsteps, gals, key_pmats, &q, batch));
        for (int c = 0; c < cols; ++c) PZ_TRY(dev_normalize(M, B, rv, (int)p->res_base2k, 0, c, cv, (int)p->key_base2k, c));
        return PZ_OK;
    }
    PZ_REQUIRE(p->a_size == p->res_size && p->a_base2k == p->res_base2k,
               "glwe_trace: a and res describe the same ciphertexts when res is in the keys' base");
    if (batch == 0) return PZ_OK;
    const long long n = (long long)M->n;
    const int cols = (int)p->rank + 1;
    const long long ct = n * cols * (long long)p->res_size;
    // the one-bit shift in front of step s + 1 rides on the tail of step s where that path has the shifted-store variant
    // (POULPY_DBG_TRACE_RSH=0: always the separate pass)
    static const int fuse_rsh = exp_knob("POULPY_DBG_TRACE_RSH", 1);
    bool shifted = false;
    for (size_t s = 0; s < nsteps; ++s) {
        PZ_REQUIRE((gals[s] & 1) != 0, "glwe_trace: Galois elements must be odd");
        if (!shifted) PZ_TRY(launch_rsh(M, (int)batch, (long long*)res, ct, cols, (int)p->res_size, 0, cols, (int)p->res_base2k, 1));
        AutoSpec au{(long long)gals[s], 1};
        bool rsh = fuse_rsh && s + 1 < nsteps;
        PZ_TRY(glwe_op(M, true, res, res, key_pmats[s], p, batch, &au, nullptr, false, &rsh));
        shifted = rsh;
    }
    return PZ_OK;
}

int pz_glwe_trace_batched(pz_module* M, int64_t* res, size_t nsteps, const int64_t* gals, const double* const* key_pmats,
                          const pz_glwe_op_params* p, size_t batch) {
    PZ_ENTER(M);
    KeyHash k;
    k.add((int)1); k.add(res); k.add(nsteps); k.add(batch);
    if (p) k.add(*p);
    for (size_t s = 0; s < nsteps && gals && key_pmats; ++s) { k.add(gals[s]); k.add(key_pmats[s]); }
    graph_key_module(M, k);
    return with_graph(M, k.h, [&]() { return glwe_trace(M, res, nsteps, gals, key_pmats, p, batch); });
}
}  // extern "C"

// api_lwe.hip composes the LWE <-> GLWE conversions around the batched key switch while holding the module lock
namespace pz {
int glwe_keyswitch_nolock(pz_module* M, int64_t* res, const int64_t* a, const double* key_pmat, const pz_glwe_op_params* p, size_t batch) {
    return glwe_entry(M, true, false, res, a, key_pmat, p, batch, nullptr);
}
}  // namespace pz
