// tail_forms.hpp — the instantiations of the fused tail (k_inv_tail, device_fft.hpp) and the choice among them.  Included by two translation
// units: launch_tail.hip instantiates the product forms (PROBE = false), launch_tail_probe.hip the same forms with the rounding-margin block
// compiled in (PROBE = true) - they compile side by side.
#pragma once
#include "internal.hpp"

namespace pz {

#define PZ_P1F_CASES(X) X(4, 1, 4) X(8, 1, 4) X(8, 1, 16) X(16, 1, 16) X(4, 4, 16) X(4, 8, 16) X(8, 8, 16) X(8, 16, 16) X(16, 16, 16) X(8, 16, 8)
// the forms beyond the plain one - digits leaving through a one-bit vec_znx_rsh (glwe_trace), the sign-only tail of a spectral automorphism,
// the tensoring tails - are instantiated for the 128-point-row plans (N = 2^13 .. 2^16)
#define PZ_RSH_CASES(X) X(4, 8, 16) X(8, 8, 16) X(8, 16, 16) X(16, 16, 16)
// the 32-bit-accumulator form (blind rotation on the pipeline): every 128-point-row plan it runs on, N = 2^12 .. 2^16
#define PZ_ACC32_CASES(X) X(4, 4, 16) PZ_RSH_CASES(X)

struct TailForm {
    enum Kind { PLAIN, RSH, SGN, NZ1, NZ2, ACC32, NZ1W, NZ2R, NZ1O, NZ2O, SGN16, SGN16R } kind = PLAIN;   // SGN16: the sign-only form with a 16-bit operand; SGN16R: + the shifted store   // NZ1W / NZ2R: the tensoring tails with 16-bit side copies (write / read); NZ1O / NZ2O: 16-bit digits ONLY   // ACC32: 32-bit accumulator digits (blind rotation's pipeline path)
    bool rowmajor = false, has_small = false;   // PLAIN: one instantiation per (row-major, body add) combination
};

template <bool PROBE>
static int tail_launch_form(pz_module* M, const TailArgs& g, int blocks, const TailForm& f) {
    const FftPlan& pl = M->plan;
#define PZ_TAIL_GO(A, B, C, R_, S_, RSH_, NZ_, SGN_) PZ_TAIL_GO2(A, B, C, R_, S_, RSH_, NZ_, SGN_, false)
#define PZ_TAIL_GO2(A, B, C, R_, S_, RSH_, NZ_, SGN_, A32_)                                                     \
    {                                                                                                           \
        const size_t lds = ((size_t)2 * (A + 1) * C * B + 2 * A * B) * sizeof(cplx);                            \
        PZ_TRY(set_lds((k_inv_tail<A, B, C, R_, S_, RSH_, NZ_, SGN_, PROBE, A32_>), lds));                      \
        hipLaunchKernelGGL((k_inv_tail<A, B, C, R_, S_, RSH_, NZ_, SGN_, PROBE, A32_>), dim3(blocks), dim3(TailShape<A, B, C>::NT), lds, M->stream, g); \
        PZ_HIP(hipGetLastError());                                                                              \
        return PZ_OK;                                                                                           \
    }
    if (f.kind == TailForm::ACC32) {
#define X(A, B, C) if (pl.f1a == A && pl.f1b == B && pl.cb == C) PZ_TAIL_GO2(A, B, C, true, true, false, 0, false, true)
        PZ_ACC32_CASES(X)
#undef X
        return fail(PZ_ERR_UNSUPPORTED, "fused tail: the 32-bit-accumulator form is not instantiated for this plan");
    }
    if (f.kind != TailForm::PLAIN) {
#define X(A, B, C)                                                                                              \
    if (pl.f1a == A && pl.f1b == B && pl.cb == C) {                                                             \
        if (f.kind == TailForm::RSH) PZ_TAIL_GO(A, B, C, true, true, true, 0, false)                            \
        if (f.kind == TailForm::SGN) PZ_TAIL_GO(A, B, C, true, false, false, 0, true)                           \
        if (f.kind == TailForm::SGN16) PZ_TAIL_GO(A, B, C, true, false, false, 7, true)                         \
        if (f.kind == TailForm::SGN16R) PZ_TAIL_GO(A, B, C, true, false, true, 7, true)                         \
        if (f.kind == TailForm::NZ2) PZ_TAIL_GO(A, B, C, true, false, false, 2, false)                          \
        if (f.kind == TailForm::NZ1) PZ_TAIL_GO(A, B, C, true, false, false, 1, false)                          \
        if (f.kind == TailForm::NZ1W) PZ_TAIL_GO(A, B, C, true, false, false, 3, false)                         \
        if (f.kind == TailForm::NZ2R) PZ_TAIL_GO(A, B, C, true, false, false, 4, false)                         \
        if (f.kind == TailForm::NZ1O) PZ_TAIL_GO(A, B, C, true, false, false, 5, false)                         \
        if (f.kind == TailForm::NZ2O) PZ_TAIL_GO(A, B, C, true, false, false, 6, false)                         \
    }
        PZ_RSH_CASES(X)
#undef X
        return fail(PZ_ERR_UNSUPPORTED, "fused tail: form %d is not instantiated for this plan", (int)f.kind);
    }
#define X(A, B, C)                                                                                              \
    if (pl.f1a == A && pl.f1b == B && pl.cb == C) {                                                             \
        if (!f.rowmajor && !f.has_small) PZ_TAIL_GO(A, B, C, false, false, false, 0, false)                     \
        if (!f.rowmajor && f.has_small) PZ_TAIL_GO(A, B, C, false, true, false, 0, false)                       \
        if (f.rowmajor && !f.has_small) PZ_TAIL_GO(A, B, C, true, false, false, 0, false)                       \
        if (f.rowmajor && f.has_small) PZ_TAIL_GO(A, B, C, true, true, false, 0, false)                         \
    }
    PZ_P1F_CASES(X)
#undef X
#undef PZ_TAIL_GO
#undef PZ_TAIL_GO2
    return fail(PZ_ERR_UNSUPPORTED, "no fused tail kernel for m1=%d", pl.m1);
}
// launch_tail_probe.hip
int tail_launch_form_probe(pz_module* M, const TailArgs& g, int blocks, const TailForm& f);

}  // namespace pz
