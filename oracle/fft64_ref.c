/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see fft64_ref.h for the full header note).
 * "parity unpinned" by reference fixtures: the Rust reference cannot be built in
 * this image and ships no golden vectors; pinned by exact-arithmetic properties.
 *
 * Plain-C restatement of poulpy-cpu-ref's FFT64 family.  Same operation order
 * as the reference so that f64 spectra come out bit-identical to the Rust code
 * when built with -ffp-contract=off against the same libm.
 */
#include "fft64_ref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* twiddle tables: reim/table_fft.rs, reim/table_ifft.rs, reim/mod.rs:50-64  */
/* ------------------------------------------------------------------------ */

struct pzr_tables {
    size_t m;
    double* fwd; /* 2m */
    double* inv; /* 2m */
};

static const double PZR_PI = 3.14159265358979323846264338327950288;

/* reim/mod.rs:50-64 */
static double frac_rev_bits(size_t x) {
    if (x == 0) return 0.0;
    if (x == 1) return 0.5;
    if ((x & 1) == 0) return frac_rev_bits(x >> 1) * 0.5;
    return frac_rev_bits(x >> 1) * 0.5 + 0.5;
}

static size_t log2_ceil(size_t m) { /* usize::BITS - (m-1).leading_zeros() */
    size_t l = 0;
    size_t v = m - 1;
    while (v) { ++l; v >>= 1; }
    return l;
}

/* table_fft.rs:85-93 */
static size_t fill_fft2(double j, double* omg, size_t pos) {
    double* o = omg + pos;
    double angle = j / 2.0;
    double two_pi = 2.0 * PZR_PI;
    o[0] = cos(two_pi * angle);
    o[1] = sin(two_pi * angle);
    return pos + 2;
}

/* table_fft.rs:96-107 */
static size_t fill_fft4(double j, double* omg, size_t pos) {
    double* o = omg + pos;
    double a1 = j / 2.0, a2 = j / 4.0;
    double two_pi = 2.0 * PZR_PI;
    o[0] = cos(two_pi * a1);
    o[1] = sin(two_pi * a1);
    o[2] = cos(two_pi * a2);
    o[3] = sin(two_pi * a2);
    return pos + 4;
}

/* table_fft.rs:110-127 */
static size_t fill_fft8(double j, double* omg, size_t pos) {
    double* o = omg + pos;
    double e8 = 1. / 8.;
    double a1 = j / 2.0, a2 = j / 4.0, a4 = j / 8.0;
    double two_pi = 2.0 * PZR_PI;
    o[0] = cos(two_pi * a1);
    o[1] = sin(two_pi * a1);
    o[2] = cos(two_pi * a2);
    o[3] = sin(two_pi * a2);
    o[4] = cos(two_pi * a4);
    o[5] = cos(two_pi * (a4 + e8));
    o[6] = sin(two_pi * a4);
    o[7] = sin(two_pi * (a4 + e8));
    return pos + 8;
}

/* table_fft.rs:130-157 */
static size_t fill_fft16(double j, double* omg, size_t pos) {
    double* o = omg + pos;
    double e8 = 1. / 8., e16 = 1. / 16.;
    double a1 = j / 2.0, a2 = j / 4.0, a4 = j / 8.0, a8 = j / 16.0;
    double two_pi = 2.0 * PZR_PI;
    o[0] = cos(two_pi * a1);
    o[1] = sin(two_pi * a1);
    o[2] = cos(two_pi * a2);
    o[3] = sin(two_pi * a2);
    o[4] = cos(two_pi * a4);
    o[5] = sin(two_pi * a4);
    o[6] = cos(two_pi * (a4 + e8));
    o[7] = sin(two_pi * (a4 + e8));
    o[8] = cos(two_pi * a8);
    o[9] = cos(two_pi * (a8 + e8));
    o[10] = cos(two_pi * (a8 + e16));
    o[11] = cos(two_pi * (a8 + e8 + e16));
    o[12] = sin(two_pi * a8);
    o[13] = sin(two_pi * (a8 + e8));
    o[14] = sin(two_pi * (a8 + e16));
    o[15] = sin(two_pi * (a8 + e8 + e16));
    return pos + 16;
}

/* table_fft.rs:160-200 */
static size_t fill_fft_bfs_16(size_t m, double j, double* omg, size_t pos) {
    size_t log_m = log2_ceil(m);
    size_t mm = m;
    double jj = j;
    double two_pi = 2.0 * PZR_PI;
    if (log_m & 1) {
        size_t h = mm >> 1;
        double j2 = jj * 0.5;
        omg[pos] = cos(two_pi * j2);
        omg[pos + 1] = sin(two_pi * j2);
        pos += 2;
        mm = h;
        jj = j2;
    }
    while (mm > 16) {
        size_t h = mm >> 2;
        double j4 = jj * (1. / 4.);
        for (size_t i = 0; i < m; i += mm) {
            double rs_0 = j4 + frac_rev_bits(i / mm) * (1. / 4.);
            double rs_1 = 2.0 * rs_0;
            omg[pos] = cos(two_pi * rs_1);
            omg[pos + 1] = sin(two_pi * rs_1);
            omg[pos + 2] = cos(two_pi * rs_0);
            omg[pos + 3] = sin(two_pi * rs_0);
            pos += 4;
        }
        mm = h;
        jj = j4;
    }
    for (size_t i = 0; i < m; i += 16) {
        double jb = jj + frac_rev_bits(i >> 4);
        fill_fft16(jb, omg, pos);
        pos += 16;
    }
    return pos;
}

/* table_fft.rs:203-217 */
static size_t fill_fft_rec_16(size_t m, double j, double* omg, size_t pos) {
    if (m <= 2048) return fill_fft_bfs_16(m, j, omg, pos);
    size_t h = m >> 1;
    double s = j * 0.5;
    double two_pi = 2.0 * PZR_PI;
    omg[pos] = cos(two_pi * s);
    omg[pos + 1] = sin(two_pi * s);
    pos += 2;
    pos = fill_fft_rec_16(h, s, omg, pos);
    pos = fill_fft_rec_16(h, s + 0.5, omg, pos);
    return pos;
}

/* table_ifft.rs:85-93 (note the exp2(2) forms there) */
static size_t fill_ifft2(double j, double* omg, size_t pos) {
    double* o = omg + pos;
    double angle = j / exp2(2.0);
    double two_pi = exp2(2.0) * PZR_PI;
    o[0] = cos(two_pi * angle);
    o[1] = -sin(two_pi * angle);
    return pos + 2;
}

/* table_ifft.rs:96-107 */
static size_t fill_ifft4(double j, double* omg, size_t pos) {
    double* o = omg + pos;
    double a1 = j / 2.0, a2 = j / 4.0;
    double two_pi = 2.0 * PZR_PI;
    o[0] = cos(two_pi * a2);
    o[1] = -sin(two_pi * a2);
    o[2] = cos(two_pi * a1);
    o[3] = -sin(two_pi * a1);
    return pos + 4;
}

/* table_ifft.rs:110-127 */
static size_t fill_ifft8(double j, double* omg, size_t pos) {
    double* o = omg + pos;
    double e8 = 1. / 8.;
    double a1 = j / 2.0, a2 = j / 4.0, a4 = j / 8.0;
    double two_pi = 2.0 * PZR_PI;
    o[0] = cos(two_pi * a4);
    o[1] = cos(two_pi * (a4 + e8));
    o[2] = -sin(two_pi * a4);
    o[3] = -sin(two_pi * (a4 + e8));
    o[4] = cos(two_pi * a2);
    o[5] = -sin(two_pi * a2);
    o[6] = cos(two_pi * a1);
    o[7] = -sin(two_pi * a1);
    return pos + 8;
}

/* table_ifft.rs:130-157 */
static size_t fill_ifft16(double j, double* omg, size_t pos) {
    double* o = omg + pos;
    double e8 = 1. / 8., e16 = 1. / 16.;
    double a1 = j / 2.0, a2 = j / 4.0, a4 = j / 8.0, a8 = j / 16.0;
    double two_pi = 2.0 * PZR_PI;
    o[0] = cos(two_pi * a8);
    o[1] = cos(two_pi * (a8 + e8));
    o[2] = cos(two_pi * (a8 + e16));
    o[3] = cos(two_pi * (a8 + e8 + e16));
    o[4] = -sin(two_pi * a8);
    o[5] = -sin(two_pi * (a8 + e8));
    o[6] = -sin(two_pi * (a8 + e16));
    o[7] = -sin(two_pi * (a8 + e8 + e16));
    o[8] = cos(two_pi * a4);
    o[9] = -sin(two_pi * a4);
    o[10] = cos(two_pi * (a4 + e8));
    o[11] = -sin(two_pi * (a4 + e8));
    o[12] = cos(two_pi * a2);
    o[13] = -sin(two_pi * a2);
    o[14] = cos(two_pi * a1);
    o[15] = -sin(two_pi * a1);
    return pos + 16;
}

/* table_ifft.rs:160-200 */
static size_t fill_ifft_bfs_16(size_t m, double j, double* omg, size_t pos) {
    size_t log_m = log2_ceil(m);
    double jj = j * 16.0 / (double)m;
    for (size_t i = 0; i < m; i += 16) {
        double jb = jj + frac_rev_bits(i >> 4);
        fill_ifft16(jb, omg, pos);
        pos += 16;
    }
    size_t h = 16;
    size_t m_half = m >> 1;
    double two_pi = 2.0 * PZR_PI;
    while (h < m_half) {
        size_t mm = h << 2;
        for (size_t i = 0; i < m; i += mm) {
            double rs_0 = jj + frac_rev_bits(i / mm) / 4.0;
            double rs_1 = 2.0 * rs_0;
            omg[pos] = cos(two_pi * rs_0);
            omg[pos + 1] = -sin(two_pi * rs_0);
            omg[pos + 2] = cos(two_pi * rs_1);
            omg[pos + 3] = -sin(two_pi * rs_1);
            pos += 4;
        }
        h = mm;
        jj = jj * 4.0;
    }
    if (log_m & 1) {
        omg[pos] = cos(two_pi * jj);
        omg[pos + 1] = -sin(two_pi * jj);
        pos += 2;
        jj = jj * 2.0;
    }
    /* table_ifft.rs:197 asserts jj == j here */
    if (jj != j) abort();
    return pos;
}

/* table_ifft.rs:203-216 */
static size_t fill_ifft_rec_16(size_t m, double j, double* omg, size_t pos) {
    if (m <= 2048) return fill_ifft_bfs_16(m, j, omg, pos);
    size_t h = m >> 1;
    double s = j / 2.0;
    pos = fill_ifft_rec_16(h, s, omg, pos);
    pos = fill_ifft_rec_16(h, s + 0.5, omg, pos);
    double two_pi = 2.0 * PZR_PI;
    omg[pos] = cos(two_pi * s);
    omg[pos + 1] = -sin(two_pi * s);
    pos += 2;
    return pos;
}

/* ReimFFTTable::new / ReimIFFTTable::new  (table_fft.rs:39-69, table_ifft.rs:39-69) */
pzr_tables* pzr_tables_new(uint64_t n) {
    if (n < 2 || (n & (n - 1))) return NULL;
    size_t m = (size_t)(n >> 1);
    pzr_tables* t = (pzr_tables*)calloc(1, sizeof(*t));
    t->m = m;
    t->fwd = (double*)calloc(2 * m + 16, sizeof(double));
    t->inv = (double*)calloc(2 * m + 16, sizeof(double));
    double quarter = 1. / 4.;
    double quarter_inv = exp2(-2.0);
    switch (m) {
        case 1: break;
        case 2: fill_fft2(quarter, t->fwd, 0); fill_ifft2(quarter_inv, t->inv, 0); break;
        case 4: fill_fft4(quarter, t->fwd, 0); fill_ifft4(quarter_inv, t->inv, 0); break;
        case 8: fill_fft8(quarter, t->fwd, 0); fill_ifft8(quarter_inv, t->inv, 0); break;
        case 16: fill_fft16(quarter, t->fwd, 0); fill_ifft16(quarter_inv, t->inv, 0); break;
        default:
            if (m <= 2048) {
                fill_fft_bfs_16(m, quarter, t->fwd, 0);
                fill_ifft_bfs_16(m, quarter_inv, t->inv, 0);
            } else {
                fill_fft_rec_16(m, quarter, t->fwd, 0);
                fill_ifft_rec_16(m, quarter_inv, t->inv, 0);
            }
    }
    return t;
}

void pzr_tables_free(pzr_tables* t) {
    if (!t) return;
    free(t->fwd);
    free(t->inv);
    free(t);
}
uint64_t pzr_tables_m(const pzr_tables* t) { return t->m; }
const double* pzr_tables_omg_fft(const pzr_tables* t) { return t->fwd; }
const double* pzr_tables_omg_ifft(const pzr_tables* t) { return t->inv; }

/* ------------------------------------------------------------------------ */
/* forward FFT: reim/fft_ref.rs                                              */
/* ------------------------------------------------------------------------ */

/* fft_ref.rs:60-67 */
static inline void ctw(double* ra, double* ia, double* rb, double* ib, double wr, double wi) {
    double dr = *rb * wr - *ib * wi;
    double di = *rb * wi + *ib * wr;
    *rb = *ra - dr;
    *ib = *ia - di;
    *ra = *ra + dr;
    *ia = *ia + di;
}

/* fft_ref.rs:70-77 */
static inline void citw(double* ra, double* ia, double* rb, double* ib, double wr, double wi) {
    double dr = *rb * wi + *ib * wr;
    double di = *rb * wr - *ib * wi;
    *rb = *ra + dr;
    *ib = *ia - di;
    *ra = *ra - dr;
    *ia = *ia + di;
}

#define TW(a, b, wr, wi) ctw(&re[a], &im[a], &re[b], &im[b], (wr), (wi))
#define ITW(a, b, wr, wi) citw(&re[a], &im[a], &re[b], &im[b], (wr), (wi))

/* fft_ref.rs:80-85 */
static void fft2(double* re, double* im, const double* o) { TW(0, 1, o[0], o[1]); }

/* fft_ref.rs:88-105 */
static void fft4(double* re, double* im, const double* o) {
    TW(0, 2, o[0], o[1]);
    TW(1, 3, o[0], o[1]);
    TW(0, 1, o[2], o[3]);
    ITW(2, 3, o[2], o[3]);
}

/* fft_ref.rs:108-140 */
static void fft8(double* re, double* im, const double* o) {
    for (int i = 0; i < 4; ++i) TW(i, i + 4, o[0], o[1]);
    for (int i = 0; i < 2; ++i) TW(i, i + 2, o[2], o[3]);
    for (int i = 4; i < 6; ++i) ITW(i, i + 2, o[2], o[3]);
    TW(0, 1, o[4], o[6]);
    ITW(2, 3, o[4], o[6]);
    TW(4, 5, o[5], o[7]);
    ITW(6, 7, o[5], o[7]);
}

/* fft_ref.rs:143-244 */
static void fft16(double* re, double* im, const double* o) {
    for (int i = 0; i < 8; ++i) TW(i, i + 8, o[0], o[1]);
    for (int i = 0; i < 4; ++i) TW(i, i + 4, o[2], o[3]);
    for (int i = 8; i < 12; ++i) ITW(i, i + 4, o[2], o[3]);
    TW(0, 2, o[4], o[5]);
    TW(1, 3, o[4], o[5]);
    TW(8, 10, o[6], o[7]);
    TW(9, 11, o[6], o[7]);
    ITW(4, 6, o[4], o[5]);
    ITW(5, 7, o[4], o[5]);
    ITW(12, 14, o[6], o[7]);
    ITW(13, 15, o[6], o[7]);
    TW(0, 1, o[8], o[12]);
    TW(4, 5, o[9], o[13]);
    TW(8, 9, o[10], o[14]);
    TW(12, 13, o[11], o[15]);
    ITW(2, 3, o[8], o[12]);
    ITW(6, 7, o[9], o[13]);
    ITW(10, 11, o[10], o[14]);
    ITW(14, 15, o[11], o[15]);
}

/* fft_ref.rs:280-290 */
static void twiddle_fft(size_t h, double* re, double* im, const double* o) {
    for (size_t i = 0; i < h; ++i) ctw(&re[i], &im[i], &re[h + i], &im[h + i], o[0], o[1]);
}

/* fft_ref.rs:293-316 */
static void bitwiddle_fft(size_t h, double* re, double* im, const double* o) {
    double *r0 = re, *r1 = re + h, *r2 = re + 2 * h, *r3 = re + 3 * h;
    double *i0 = im, *i1 = im + h, *i2 = im + 2 * h, *i3 = im + 3 * h;
    for (size_t i = 0; i < h; ++i) {
        ctw(&r0[i], &i0[i], &r2[i], &i2[i], o[0], o[1]);
        ctw(&r1[i], &i1[i], &r3[i], &i3[i], o[0], o[1]);
    }
    for (size_t i = 0; i < h; ++i) {
        ctw(&r0[i], &i0[i], &r1[i], &i1[i], o[2], o[3]);
        citw(&r2[i], &i2[i], &r3[i], &i3[i], o[2], o[3]);
    }
}

/* fft_ref.rs:247-277 */
static size_t fft_bfs_16(size_t m, double* re, double* im, const double* omg, size_t pos) {
    size_t log_m = log2_ceil(m);
    size_t mm = m;
    if (log_m & 1) {
        size_t h = mm >> 1;
        twiddle_fft(h, re, im, omg + pos);
        pos += 2;
        mm = h;
    }
    while (mm > 16) {
        size_t h = mm >> 2;
        for (size_t off = 0; off < m; off += mm) {
            bitwiddle_fft(h, re + off, im + off, omg + pos);
            pos += 4;
        }
        mm = h;
    }
    for (size_t off = 0; off < m; off += 16) {
        fft16(re + off, im + off, omg + pos);
        pos += 16;
    }
    return pos;
}

/* fft_ref.rs:46-57 */
static size_t fft_rec_16(size_t m, double* re, double* im, const double* omg, size_t pos) {
    if (m <= 2048) return fft_bfs_16(m, re, im, omg, pos);
    size_t h = m >> 1;
    twiddle_fft(h, re, im, omg + pos);
    pos += 2;
    pos = fft_rec_16(h, re, im, omg, pos);
    pos = fft_rec_16(h, re + h, im + h, omg, pos);
    return pos;
}

/* fft_ref.rs:25-43 */
void pzr_fft(const pzr_tables* t, double* data) {
    size_t m = t->m;
    double* re = data;
    double* im = data + m;
    const double* omg = t->fwd;
    if (m <= 16) {
        switch (m) {
            case 2: fft2(re, im, omg); break;
            case 4: fft4(re, im, omg); break;
            case 8: fft8(re, im, omg); break;
            case 16: fft16(re, im, omg); break;
            default: break;
        }
    } else if (m <= 2048) {
        fft_bfs_16(m, re, im, omg, 0);
    } else {
        fft_rec_16(m, re, im, omg, 0);
    }
}

/* ------------------------------------------------------------------------ */
/* inverse FFT: reim/ifft_ref.rs                                             */
/* ------------------------------------------------------------------------ */

/* ifft_ref.rs:91-98 */
static inline void invtw(double* ra, double* ia, double* rb, double* ib, double wr, double wi) {
    double rd = *ra - *rb;
    double id = *ia - *ib;
    *ra = *ra + *rb;
    *ia = *ia + *ib;
    *rb = rd * wr - id * wi;
    *ib = rd * wi + id * wr;
}

/* ifft_ref.rs:101-108 */
static inline void invitw(double* ra, double* ia, double* rb, double* ib, double wr, double wi) {
    double rd = *ra - *rb;
    double id = *ia - *ib;
    *ra = *ra + *rb;
    *ia = *ia + *ib;
    *rb = rd * wi + id * wr;
    *ib = -rd * wr + id * wi;
}

#define IW(a, b, wr, wi) invtw(&re[a], &im[a], &re[b], &im[b], (wr), (wi))
#define IIW(a, b, wr, wi) invitw(&re[a], &im[a], &re[b], &im[b], (wr), (wi))

/* ifft_ref.rs:111-116 */
static void ifft2(double* re, double* im, const double* o) { IW(0, 1, o[0], o[1]); }

/* ifft_ref.rs:119-137 */
static void ifft4(double* re, double* im, const double* o) {
    IW(0, 1, o[0], o[1]);
    IIW(2, 3, o[0], o[1]);
    IW(0, 2, o[2], o[3]);
    IW(1, 3, o[2], o[3]);
}

/* ifft_ref.rs:140-172 */
static void ifft8(double* re, double* im, const double* o) {
    IW(0, 1, o[0], o[2]);
    IIW(2, 3, o[0], o[2]);
    IW(4, 5, o[1], o[3]);
    IIW(6, 7, o[1], o[3]);
    IW(0, 2, o[4], o[5]);
    IW(1, 3, o[4], o[5]);
    IIW(4, 6, o[4], o[5]);
    IIW(5, 7, o[4], o[5]);
    for (int i = 0; i < 4; ++i) IW(i, i + 4, o[6], o[7]);
}

/* ifft_ref.rs:175-272 */
static void ifft16(double* re, double* im, const double* o) {
    IW(0, 1, o[0], o[4]);
    IIW(2, 3, o[0], o[4]);
    IW(4, 5, o[1], o[5]);
    IIW(6, 7, o[1], o[5]);
    IW(8, 9, o[2], o[6]);
    IIW(10, 11, o[2], o[6]);
    IW(12, 13, o[3], o[7]);
    IIW(14, 15, o[3], o[7]);

    IW(0, 2, o[8], o[9]);
    IW(1, 3, o[8], o[9]);
    IIW(4, 6, o[8], o[9]);
    IIW(5, 7, o[8], o[9]);
    IW(8, 10, o[10], o[11]);
    IW(9, 11, o[10], o[11]);
    IIW(12, 14, o[10], o[11]);
    IIW(13, 15, o[10], o[11]);

    for (int i = 0; i < 4; ++i) IW(i, i + 4, o[12], o[13]);
    for (int i = 8; i < 12; ++i) IIW(i, i + 4, o[12], o[13]);

    for (int i = 0; i < 8; ++i) IW(i, i + 8, o[14], o[15]);
}

/* ifft_ref.rs:275-285 */
static void inv_twiddle_ifft(size_t h, double* re, double* im, const double* o) {
    for (size_t i = 0; i < h; ++i) invtw(&re[i], &im[i], &re[h + i], &im[h + i], o[0], o[1]);
}

/* ifft_ref.rs:288-311 */
static void inv_bitwiddle_ifft(size_t h, double* re, double* im, const double* o) {
    double *r0 = re, *r1 = re + h, *r2 = re + 2 * h, *r3 = re + 3 * h;
    double *i0 = im, *i1 = im + h, *i2 = im + 2 * h, *i3 = im + 3 * h;
    for (size_t i = 0; i < h; ++i) {
        invtw(&r0[i], &i0[i], &r1[i], &i1[i], o[0], o[1]);
        invitw(&r2[i], &i2[i], &r3[i], &i3[i], o[0], o[1]);
    }
    for (size_t i = 0; i < h; ++i) {
        invtw(&r0[i], &i0[i], &r2[i], &i2[i], o[2], o[3]);
        invtw(&r1[i], &i1[i], &r3[i], &i3[i], o[2], o[3]);
    }
}

/* ifft_ref.rs:58-88 */
static size_t ifft_bfs_16(size_t m, double* re, double* im, const double* omg, size_t pos) {
    size_t log_m = log2_ceil(m);
    for (size_t off = 0; off < m; off += 16) {
        ifft16(re + off, im + off, omg + pos);
        pos += 16;
    }
    size_t h = 16;
    size_t m_half = m >> 1;
    while (h < m_half) {
        size_t mm = h << 2;
        for (size_t off = 0; off < m; off += mm) {
            inv_bitwiddle_ifft(h, re + off, im + off, omg + pos);
            pos += 4;
        }
        h = mm;
    }
    if (log_m & 1) {
        inv_twiddle_ifft(h, re, im, omg + pos);
        pos += 2;
    }
    return pos;
}

/* ifft_ref.rs:45-55 */
static size_t ifft_rec_16(size_t m, double* re, double* im, const double* omg, size_t pos) {
    if (m <= 2048) return ifft_bfs_16(m, re, im, omg, pos);
    size_t h = m >> 1;
    pos = ifft_rec_16(h, re, im, omg, pos);
    pos = ifft_rec_16(h, re + h, im + h, omg, pos);
    inv_twiddle_ifft(h, re, im, omg + pos);
    pos += 2;
    return pos;
}

/* ifft_ref.rs:24-42 */
void pzr_ifft(const pzr_tables* t, double* data) {
    size_t m = t->m;
    double* re = data;
    double* im = data + m;
    const double* omg = t->inv;
    if (m <= 16) {
        switch (m) {
            case 2: ifft2(re, im, omg); break;
            case 4: ifft4(re, im, omg); break;
            case 8: ifft8(re, im, omg); break;
            case 16: ifft16(re, im, omg); break;
            default: break;
        }
    } else if (m <= 2048) {
        ifft_bfs_16(m, re, im, omg, 0);
    } else {
        ifft_rec_16(m, re, im, omg, 0);
    }
}

/* ------------------------------------------------------------------------ */
/* conversions: reim/conversion.rs                                           */
/* ------------------------------------------------------------------------ */

/* conversion.rs:19-28 */
void pzr_reim_from_znx_i64(double* res, const int64_t* a, size_t len) {
    for (size_t i = 0; i < len; ++i) res[i] = (double)a[i];
}

/* Rust `f64 as i64`: saturating, NaN -> 0 */
static inline int64_t f64_to_i64_sat(double x) {
    if (x != x) return 0;
    if (x >= 9223372036854775808.0) return INT64_MAX;
    if (x <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)x;
}

/* Rounding-margin probe of the ORACLE (round 6; not in the reference): max |x - round(x)| over every value the two conversions below
 * round while the probe is on - how far cpu-ref's own f64 transforms stay from a wrong limb on the inputs of a test
 * (poulpy-hal/docs/backend_safety_contract.md:25-27 asks for a documented tolerance; tools/margin.py, tests/test_gpu_structured.py).
 * A plain global: single-threaded diagnostic use only; off by default (the timed cpu_baseline leg never turns it on). */
static int pzr_margin_on = 0;
static double pzr_margin_max = 0.0;
void pzr_margin_probe_set(int on) { pzr_margin_on = on; pzr_margin_max = 0.0; }
double pzr_margin_probe_get(void) { return pzr_margin_max; }
static inline void pzr_margin_note(double x) {
    if (x == x && fabs(x) < 4503599627370496.0) {   /* |x| < 2^52: the value has a fractional part at all */
        double d = fabs(x - round(x));
        if (d > pzr_margin_max) pzr_margin_max = d;
    }
}

/* conversion.rs:43-52 ; f64::round = half away from zero = C round() */
void pzr_reim_to_znx_i64(int64_t* res, double divisor, const double* a, size_t len) {
    double inv_div = 1. / divisor;
    if (pzr_margin_on) for (size_t i = 0; i < len; ++i) pzr_margin_note(a[i] * inv_div);
    for (size_t i = 0; i < len; ++i) res[i] = f64_to_i64_sat(round(a[i] * inv_div));
}

/* conversion.rs:55-60 */
void pzr_reim_to_znx_i64_assign(double* res, double divisor, size_t len) {
    double inv_div = 1. / divisor;
    if (pzr_margin_on) for (size_t i = 0; i < len; ++i) pzr_margin_note(res[i] * inv_div);
    for (size_t i = 0; i < len; ++i) {
        int64_t v = f64_to_i64_sat(round(res[i] * inv_div));
        memcpy(&res[i], &v, sizeof(v));
    }
}

/* ------------------------------------------------------------------------ */
/* pointwise helpers: reim/fft_vec.rs                                        */
/* ------------------------------------------------------------------------ */

static inline double* at_f64(double* p, size_t n, size_t cols, size_t col, size_t limb) { return p + n * (limb * cols + col); }
static inline const double* at_cf64(const double* p, size_t n, size_t cols, size_t col, size_t limb) { return p + n * (limb * cols + col); }
static inline int64_t* at_i64(int64_t* p, size_t n, size_t cols, size_t col, size_t limb) { return p + n * (limb * cols + col); }
static inline const int64_t* at_ci64(const int64_t* p, size_t n, size_t cols, size_t col, size_t limb) { return p + n * (limb * cols + col); }
static inline size_t zmin(size_t a, size_t b) { return a < b ? a : b; }

static void reim_zero(double* r, size_t len) { memset(r, 0, len * sizeof(double)); }
static void reim_copy(double* r, const double* a, size_t len) { memmove(r, a, len * sizeof(double)); }

/* fft_vec.rs:150-173 (res = a*b) ; :124-147 (res = a*res) */
static void reim_mul(double* res, const double* a, const double* b, size_t n) {
    size_t m = n >> 1;
    for (size_t i = 0; i < m; ++i) {
        double ar = a[i], ai = a[m + i], br = b[i], bi = b[m + i];
        double rr = ar * br - ai * bi;
        double ri = ar * bi + ai * br;
        res[i] = rr;
        res[m + i] = ri;
    }
}

/* ------------------------------------------------------------------------ */
/* reference/fft64/vec_znx_dft.rs                                            */
/* ------------------------------------------------------------------------ */

/* vec_znx_dft.rs:160-200 */
void pzr_vec_znx_dft_apply(const pzr_tables* t, size_t step, size_t offset,
                           double* res, size_t res_cols, size_t res_size, size_t res_col,
                           const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t n = t->m << 1;
    size_t steps = (a_size + step - 1) / step;
    size_t min_steps = zmin(res_size, steps);
    for (size_t j = 0; j < min_steps; ++j) {
        size_t limb = offset + j * step;
        if (limb < a_size) {
            double* out = at_f64(res, n, res_cols, res_col, j);
            pzr_reim_from_znx_i64(out, at_ci64(a, n, a_cols, a_col, limb), n);
            pzr_fft(t, out);
        } /* else: left untouched (vec_znx_dft.rs:191-194) */
    }
    for (size_t j = min_steps; j < res_size; ++j) reim_zero(at_f64(res, n, res_cols, res_col, j), n);
}

/* vec_znx_dft.rs:202-232 */
void pzr_vec_znx_idft_apply(const pzr_tables* t,
                            int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                            const double* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t n = t->m << 1;
    size_t min_size = zmin(res_size, a_size);
    double divisor = (double)t->m;
    for (size_t j = 0; j < min_size; ++j) {
        double* slot = (double*)at_i64(res, n, res_cols, res_col, j);
        reim_copy(slot, at_cf64(a, n, a_cols, a_col, j), n);
        pzr_ifft(t, slot);
        pzr_reim_to_znx_i64_assign(slot, divisor, n);
    }
    for (size_t j = min_size; j < res_size; ++j) memset(at_i64(res, n, res_cols, res_col, j), 0, n * sizeof(int64_t));
}

/* vec_znx_dft.rs:234-262 */
void pzr_vec_znx_idft_apply_tmpa(const pzr_tables* t,
                                 int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                 double* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t n = t->m << 1;
    size_t min_size = zmin(res_size, a_size);
    double divisor = (double)t->m;
    for (size_t j = 0; j < min_size; ++j) {
        double* aj = at_f64(a, n, a_cols, a_col, j);
        pzr_ifft(t, aj);
        pzr_reim_to_znx_i64(at_i64(res, n, res_cols, res_col, j), divisor, aj, n);
    }
    for (size_t j = min_size; j < res_size; ++j) memset(at_i64(res, n, res_cols, res_col, j), 0, n * sizeof(int64_t));
}

/* vec_znx_dft.rs:264-288 */
void pzr_vec_znx_idft_apply_consume(const pzr_tables* t, double* data, size_t cols, size_t size) {
    size_t n = t->m << 1;
    double divisor = (double)t->m;
    for (size_t i = 0; i < cols; ++i)
        for (size_t j = 0; j < size; ++j) {
            double* p = at_f64(data, n, cols, i, j);
            pzr_ifft(t, p);
            pzr_reim_to_znx_i64_assign(p, divisor, n);
        }
}

/* vec_znx_dft.rs:14-66 (SUB=0) and :290-342 (SUB=1) */
static void dft_add_sub(int sub, size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col,
                        const double* a, size_t a_cols, size_t a_size, size_t a_col,
                        const double* b, size_t b_cols, size_t b_size, size_t b_col) {
    int a_le_b = a_size <= b_size;
    size_t sum_size = zmin(a_le_b ? a_size : b_size, res_size);
    size_t cpy_size = zmin(a_le_b ? b_size : a_size, res_size);
    for (size_t j = 0; j < sum_size; ++j) {
        double* r = at_f64(res, n, res_cols, res_col, j);
        const double* x = at_cf64(a, n, a_cols, a_col, j);
        const double* y = at_cf64(b, n, b_cols, b_col, j);
        for (size_t i = 0; i < n; ++i) r[i] = sub ? x[i] - y[i] : x[i] + y[i];
    }
    for (size_t j = sum_size; j < cpy_size; ++j) {
        double* r = at_f64(res, n, res_cols, res_col, j);
        if (a_le_b) {
            const double* y = at_cf64(b, n, b_cols, b_col, j);
            if (sub) { for (size_t i = 0; i < n; ++i) r[i] = -y[i]; }
            else reim_copy(r, y, n);
        } else {
            reim_copy(r, at_cf64(a, n, a_cols, a_col, j), n);
        }
    }
    for (size_t j = cpy_size; j < res_size; ++j) reim_zero(at_f64(res, n, res_cols, res_col, j), n);
}

void pzr_vec_znx_dft_add_into(size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col,
                              const double* a, size_t a_cols, size_t a_size, size_t a_col,
                              const double* b, size_t b_cols, size_t b_size, size_t b_col) {
    dft_add_sub(0, n, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, b, b_cols, b_size, b_col);
}
void pzr_vec_znx_dft_sub(size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col,
                         const double* a, size_t a_cols, size_t a_size, size_t a_col,
                         const double* b, size_t b_cols, size_t b_size, size_t b_col) {
    dft_add_sub(1, n, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, b, b_cols, b_size, b_col);
}

/* vec_znx_dft.rs:68-91 */
void pzr_vec_znx_dft_add_assign(size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col,
                                const double* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t sum_size = zmin(a_size, res_size);
    for (size_t j = 0; j < sum_size; ++j) {
        double* r = at_f64(res, n, res_cols, res_col, j);
        const double* x = at_cf64(a, n, a_cols, a_col, j);
        for (size_t i = 0; i < n; ++i) r[i] += x[i];
    }
}

/* vec_znx_dft.rs:93-128 */
void pzr_vec_znx_dft_add_scaled_assign(size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col,
                                       const double* a, size_t a_cols, size_t a_size, size_t a_col, int64_t a_scale) {
    size_t res_shift = 0, a_shift = 0, sum_size;
    if (a_scale > 0) {
        size_t shift = zmin((size_t)a_scale, a_size);
        size_t mn = zmin(a_size, res_size);
        sum_size = mn > shift ? mn - shift : 0;
        a_shift = shift;
    } else if (a_scale < 0) {
        size_t shift = zmin((size_t)(-a_scale), res_size);
        sum_size = zmin(a_size, res_size - shift);
        res_shift = shift;
    } else {
        sum_size = zmin(a_size, res_size);
    }
    for (size_t j = 0; j < sum_size; ++j) {
        double* r = at_f64(res, n, res_cols, res_col, j + res_shift);
        const double* x = at_cf64(a, n, a_cols, a_col, j + a_shift);
        for (size_t i = 0; i < n; ++i) r[i] += x[i];
    }
}

/* vec_znx_dft.rs:344-366 */
void pzr_vec_znx_dft_sub_assign(size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col,
                                const double* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t sum_size = zmin(a_size, res_size);
    for (size_t j = 0; j < sum_size; ++j) {
        double* r = at_f64(res, n, res_cols, res_col, j);
        const double* x = at_cf64(a, n, a_cols, a_col, j);
        for (size_t i = 0; i < n; ++i) r[i] -= x[i];
    }
}

/* vec_znx_dft.rs:368-394 */
void pzr_vec_znx_dft_sub_negate_assign(size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col,
                                       const double* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t sum_size = zmin(a_size, res_size);
    for (size_t j = 0; j < sum_size; ++j) {
        double* r = at_f64(res, n, res_cols, res_col, j);
        const double* x = at_cf64(a, n, a_cols, a_col, j);
        for (size_t i = 0; i < n; ++i) r[i] = x[i] - r[i];
    }
    for (size_t j = sum_size; j < res_size; ++j) {
        double* r = at_f64(res, n, res_cols, res_col, j);
        for (size_t i = 0; i < n; ++i) r[i] = -r[i];
    }
}

/* vec_znx_dft.rs:130-158 */
void pzr_vec_znx_dft_copy(size_t n, size_t step, size_t offset,
                          double* res, size_t res_cols, size_t res_size, size_t res_col,
                          const double* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t steps = (a_size + step - 1) / step;
    size_t min_steps = zmin(res_size, steps);
    for (size_t j = 0; j < min_steps; ++j) {
        size_t limb = offset + j * step;
        if (limb < a_size) reim_copy(at_f64(res, n, res_cols, res_col, j), at_cf64(a, n, a_cols, a_col, limb), n);
        else reim_zero(at_f64(res, n, res_cols, res_col, j), n);
    }
    for (size_t j = min_steps; j < res_size; ++j) reim_zero(at_f64(res, n, res_cols, res_col, j), n);
}

/* vec_znx_dft.rs:396-405 */
void pzr_vec_znx_dft_zero(size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col) {
    for (size_t j = 0; j < res_size; ++j) reim_zero(at_f64(res, n, res_cols, res_col, j), n);
}

/* ------------------------------------------------------------------------ */
/* reference/fft64/svp.rs                                                    */
/* ------------------------------------------------------------------------ */

/* svp.rs:9-19 ; SvpPPol(n, cols) has size 1, ScalarZnx(n, cols) likewise */
void pzr_svp_prepare(const pzr_tables* t, double* res, size_t res_cols, size_t res_col,
                     const int64_t* a, size_t a_cols, size_t a_col) {
    size_t n = t->m << 1;
    (void)res_cols; (void)a_cols;
    double* out = res + n * res_col;
    pzr_reim_from_znx_i64(out, a + n * a_col, n);
    pzr_fft(t, out);
}

/* svp.rs:21-54 */
void pzr_svp_apply_dft(const pzr_tables* t,
                       double* res, size_t res_cols, size_t res_size, size_t res_col,
                       const double* ppol, size_t a_cols, size_t a_col,
                       const int64_t* b, size_t b_cols, size_t b_size, size_t b_col) {
    size_t n = t->m << 1;
    (void)a_cols;
    size_t min_size = zmin(res_size, b_size);
    const double* pp = ppol + n * a_col;
    for (size_t j = 0; j < min_size; ++j) {
        double* out = at_f64(res, n, res_cols, res_col, j);
        pzr_reim_from_znx_i64(out, at_ci64(b, n, b_cols, b_col, j), n);
        pzr_fft(t, out);
        reim_mul(out, pp, out, n); /* reim_mul_assign(out, ppol): a=ppol, b=out (fft_vec.rs:124-147) */
    }
    for (size_t j = min_size; j < res_size; ++j) reim_zero(at_f64(res, n, res_cols, res_col, j), n);
}

/* svp.rs:56-79 */
void pzr_svp_apply_dft_to_dft(size_t n,
                              double* res, size_t res_cols, size_t res_size, size_t res_col,
                              const double* ppol, size_t a_cols, size_t a_col,
                              const double* b, size_t b_cols, size_t b_size, size_t b_col) {
    (void)a_cols;
    size_t min_size = zmin(res_size, b_size);
    const double* pp = ppol + n * a_col;
    for (size_t j = 0; j < min_size; ++j) reim_mul(at_f64(res, n, res_cols, res_col, j), pp, at_cf64(b, n, b_cols, b_col, j), n);
    for (size_t j = min_size; j < res_size; ++j) reim_zero(at_f64(res, n, res_cols, res_col, j), n);
}

/* svp.rs:81-94 */
void pzr_svp_apply_dft_to_dft_assign(size_t n,
                                     double* res, size_t res_cols, size_t res_size, size_t res_col,
                                     const double* ppol, size_t a_cols, size_t a_col) {
    (void)a_cols;
    const double* pp = ppol + n * a_col;
    for (size_t j = 0; j < res_size; ++j) {
        double* r = at_f64(res, n, res_cols, res_col, j);
        reim_mul(r, pp, r, n);
    }
}

/* ------------------------------------------------------------------------ */
/* reference/fft64/vmp.rs + reim4/arithmetic_ref.rs                          */
/* ------------------------------------------------------------------------ */

size_t pzr_vmp_prepare_tmp_bytes(size_t n) { return n * sizeof(int64_t); } /* vmp.rs:13-15 */
size_t pzr_vmp_apply_dft_to_dft_tmp_bytes(size_t a_size, size_t prows, size_t pcols_in) { /* vmp.rs:132-135 */
    size_t row_max = zmin(a_size, prows);
    return (16 + 8 * row_max * pcols_in) * sizeof(double);
}
size_t pzr_vmp_apply_dft_tmp_bytes(size_t n, size_t a_size, size_t prows, size_t pcols_in) { /* vmp.rs:95-98 */
    size_t row_max = zmin(a_size, prows);
    return (16 + (n + 8) * row_max * pcols_in) * sizeof(double);
}

/* arithmetic_ref.rs:21-35 */
static void reim4_extract_1blk(size_t m, size_t rows, size_t blk, double* dst, const double* src) {
    size_t off = blk << 2;
    for (size_t r = 0; r < 2 * rows; ++r) memcpy(dst + 4 * r, src + r * m + off, 4 * sizeof(double));
}

/* arithmetic_ref.rs:223-232 */
static inline void reim4_add_mul(double* dst, const double* a, const double* b) {
    for (int k = 0; k < 4; ++k) {
        double ar = a[k], br = b[k], ai = a[k + 4], bi = b[k + 4];
        dst[k] += ar * br - ai * bi;
        dst[k + 4] += ar * bi + ai * br;
    }
}

/* arithmetic_ref.rs:138-158 */
static void reim4_mat1col(size_t nrows, double* dst, const double* u, const double* v) {
    double acc[8] = {0};
    for (size_t i = 0; i < nrows; ++i) reim4_add_mul(acc, u + 8 * i, v + 8 * i);
    memcpy(dst, acc, sizeof(acc));
}

/* arithmetic_ref.rs:161-186 */
static void reim4_mat2cols(size_t nrows, double* dst, const double* u, const double* v) {
    double acc0[8] = {0}, acc1[8] = {0};
    for (size_t i = 0; i < nrows; ++i) {
        reim4_add_mul(acc0, u + 8 * i, v + 16 * i);
        reim4_add_mul(acc1, u + 8 * i, v + 16 * i + 8);
    }
    memcpy(dst, acc0, sizeof(acc0));
    memcpy(dst + 8, acc1, sizeof(acc1));
}

/* arithmetic_ref.rs:189-220 */
static void reim4_mat2cols_2ndcol(size_t nrows, double* dst, const double* u, const double* v) {
    double acc[8] = {0};
    for (size_t i = 0; i < nrows; ++i) reim4_add_mul(acc, u + 8 * i, v + 16 * i + 8);
    memcpy(dst, acc, sizeof(acc));
}

/* arithmetic_ref.rs:53-82 (OVERWRITE = true) */
static void reim4_save_1blk(size_t m, size_t blk, double* dst, const double* src) {
    size_t off = blk << 2;
    memcpy(dst + off, src, 4 * sizeof(double));
    memcpy(dst + off + m, src + 4, 4 * sizeof(double));
}

/* arithmetic_ref.rs:85-135 (OVERWRITE = true) */
static void reim4_save_2blk(size_t m, size_t blk, double* dst, const double* src) {
    size_t off = blk << 2;
    memcpy(dst + off, src, 4 * sizeof(double));
    memcpy(dst + off + m, src + 4, 4 * sizeof(double));
    memcpy(dst + off + 2 * m, src + 8, 4 * sizeof(double));
    memcpy(dst + off + 3 * m, src + 12, 4 * sizeof(double));
}

/* vmp.rs:17-93 */
void pzr_vmp_prepare(const pzr_tables* t, double* pmat, const int64_t* mat,
                     size_t rows, size_t cols_in, size_t cols_out, size_t size) {
    size_t m = t->m, n = m << 1;
    size_t nrows = cols_in * rows;
    size_t ncols = cols_out * size;
    size_t offset = nrows * ncols * 8;
    double* tmp = (double*)malloc(n * sizeof(double));
    for (size_t row_i = 0; row_i < nrows; ++row_i) {
        for (size_t col_i = 0; col_i < ncols; ++col_i) {
            size_t pos = n * (row_i * ncols + col_i);
            pzr_reim_from_znx_i64(tmp, mat + pos, n);
            pzr_fft(t, tmp);
            double* dst;
            if (col_i == ncols - 1 && (ncols & 1)) dst = pmat + col_i * nrows * 8 + row_i * 8;
            else dst = pmat + (col_i / 2) * (nrows * 16) + row_i * 16 + (col_i % 2) * 8;
            for (size_t blk = 0; blk < (m >> 2); ++blk) reim4_extract_1blk(m, 1, blk, dst + blk * offset, tmp);
        }
    }
    free(tmp);
}

/* vmp.rs:186-264 (OVERWRITE = true).  NOTE: for limb_offset > 0 the reference
 * leaves res columns [col_max - limb_offset, col_max) unwritten;
 * that is restated literally here; pzr_vmp_apply_dft_to_dft builds the NTT120
 * semantics the build follows on top of it. */
static void vmp_apply_core(size_t n, double* res, size_t res_polys, const double* a, size_t a_polys,
                           const double* pmat, size_t limb_offset, size_t nrows, size_t ncols) {
    size_t m = n >> 1;
    size_t row_max = zmin(nrows, a_polys);
    size_t col_max = zmin(ncols, res_polys);
    if (limb_offset >= col_max) {
        reim_zero(res, res_polys * n);
        return;
    }
    double out[16];
    double* ext = (double*)malloc((8 * row_max + 8) * sizeof(double));
    for (size_t blk = 0; blk < (m >> 2); ++blk) {
        const double* mat_blk = pmat + blk * (8 * nrows * ncols);
        reim4_extract_1blk(m, row_max, blk, ext, a);
        if ((limb_offset & 1) == 0) {
            size_t col_res = 0;
            for (size_t col_pmat = limb_offset; col_pmat + 1 < col_max; col_pmat += 2, col_res += 2) {
                reim4_mat2cols(row_max, out, ext, mat_blk + col_pmat * (8 * nrows));
                reim4_save_2blk(m, blk, res + col_res * n, out);
            }
        } else {
            reim4_mat2cols_2ndcol(row_max, out, ext, mat_blk + (limb_offset - 1) * (8 * nrows));
            reim4_save_1blk(m, blk, res, out);
            size_t col_res = 1;
            for (size_t col_pmat = limb_offset + 1; col_pmat + 1 < col_max; col_pmat += 2, col_res += 2) {
                reim4_mat2cols(row_max, out, ext, mat_blk + col_pmat * (8 * nrows));
                reim4_save_2blk(m, blk, res + col_res * n, out);
            }
        }
        if (col_max & 1) {
            size_t last_col = col_max - 1;
            if (last_col >= limb_offset) {
                if (ncols == col_max) reim4_mat1col(row_max, out, ext, mat_blk + last_col * (8 * nrows));
                else reim4_mat2cols(row_max, out, ext, mat_blk + last_col * (8 * nrows));
                reim4_save_1blk(m, blk, res + (last_col - limb_offset) * n, out);
            }
        }
    }
    free(ext);
    reim_zero(res + col_max * n, (res_polys - col_max) * n);
}

/* vmp.rs:144-183 */
void pzr_vmp_apply_dft_to_dft(size_t n,
                              double* res, size_t res_cols, size_t res_size,
                              const double* a, size_t a_cols, size_t a_size,
                              const double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size,
                              size_t limb_offset) {
    size_t nrows = cols_in * rows;
    size_t ncols = cols_out * size;
    size_t res_polys = res_cols * res_size, off = limb_offset * cols_out;
    if (off == 0) {
        vmp_apply_core(n, res, res_polys, a, a_cols * a_size, pmat, 0, nrows, ncols);
        return;
    }
    /* limb_offset > 0: the FFT64 core clamps the key columns it reads to res_polys and leaves the last `off` of them
     * unwritten, while its NTT120 sibling (reference/ntt120/vmp.rs:190,281-287) reads key columns up to res_polys + off and
     * zeroes the rest.  SURVEY.md A.2 resolves this in favour of the NTT120 (evidently intended, deterministic) semantics:
     *   res[c] = sum_r a[r] * P[r][c + off] for c < min(res_polys, ncols - off), zero beyond.
     * Obtained from the literal core by running it on a zeroed result that is `off` polynomials longer. */
    double* tmp = (double*)calloc((res_polys + off) * n, sizeof(double));
    vmp_apply_core(n, tmp, res_polys + off, a, a_cols * a_size, pmat, off, nrows, ncols);
    memcpy(res, tmp, res_polys * n * sizeof(double));
    free(tmp);
}

/* vmp.rs:100-130 (hal_impl/family_common.rs:17-54 is the same glue) */
void pzr_vmp_apply_dft(const pzr_tables* t,
                       double* res, size_t res_cols, size_t res_size,
                       const int64_t* a, size_t a_cols, size_t a_size,
                       const double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size) {
    size_t n = t->m << 1;
    size_t cols = cols_in;
    size_t sz = zmin(a_size, rows);
    double* a_dft = (double*)malloc(n * cols * sz * sizeof(double) + 8);
    size_t offset = cols - a_cols;
    /* NOTE vmp.rs:124-126 calls vec_znx_dft_apply(.., &mut a_dft, j, &a, offset + j): literal restatement.
     * With a_cols == cols (every caller) offset is 0. */
    for (size_t j = 0; j < cols; ++j) {
        if (offset + j < a_cols) pzr_vec_znx_dft_apply(t, 1, 0, a_dft, cols, sz, j, a, a_cols, a_size, offset + j);
        else pzr_vec_znx_dft_zero(n, a_dft, cols, sz, j);
    }
    pzr_vmp_apply_dft_to_dft(n, res, res_cols, res_size, a_dft, cols, sz, pmat, rows, cols_in, cols_out, size, 0);
    free(a_dft);
}

/* ------------------------------------------------------------------------ */
/* reference/znx/normalization.rs                                            */
/* ------------------------------------------------------------------------ */

static inline int64_t wadd(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
static inline int64_t wshl(int64_t a, size_t s) { return (int64_t)((uint64_t)a << s); }
/* normalization.rs:4-6 */
static inline int64_t get_digit(size_t k, int64_t x) { return (int64_t)((uint64_t)x << (64 - k)) >> (64 - k); }
/* normalization.rs:9-11 */
static inline int64_t get_carry(size_t k, int64_t x, int64_t d) { return (int64_t)((uint64_t)x - (uint64_t)d) >> k; }

/* normalization.rs:24-41 */
static void nz_first_step_carry_only(size_t k, size_t lsh, const int64_t* x, int64_t* c, size_t n) {
    size_t kk = lsh == 0 ? k : k - lsh;
    for (size_t i = 0; i < n; ++i) c[i] = get_carry(kk, x[i], get_digit(kk, x[i]));
}

/* normalization.rs:107-129 */
static void nz_middle_step_carry_only(size_t k, size_t lsh, const int64_t* x, int64_t* c, size_t n) {
    size_t kk = lsh == 0 ? k : k - lsh;
    for (size_t i = 0; i < n; ++i) {
        int64_t d = get_digit(kk, x[i]);
        int64_t cr = get_carry(kk, x[i], d);
        int64_t dpc = wadd(wshl(d, lsh), c[i]);
        c[i] = wadd(cr, get_carry(k, dpc, get_digit(k, dpc)));
    }
}

/* normalization.rs:132-157 */
static void nz_middle_step_assign(size_t k, size_t lsh, int64_t* x, int64_t* c, size_t n) {
    size_t kk = lsh == 0 ? k : k - lsh;
    for (size_t i = 0; i < n; ++i) {
        int64_t d = get_digit(kk, x[i]);
        int64_t cr = get_carry(kk, x[i], d);
        int64_t dpc = wadd(wshl(d, lsh), c[i]);
        x[i] = get_digit(k, dpc);
        c[i] = wadd(cr, get_carry(k, dpc, x[i]));
    }
}

/* normalization.rs:160-166 */
static void nz_extract_digit_addmul(size_t k, size_t lsh, int64_t* res, int64_t* src, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        int64_t d = get_digit(k, src[i]);
        src[i] = get_carry(k, src[i], d);
        res[i] = wadd(res[i], wshl(d, lsh));
    }
}

/* normalization.rs:179-221 (OVERWRITE = true) */
static void nz_middle_step(size_t k, size_t lsh, int64_t* x, const int64_t* a, int64_t* c, size_t n) {
    size_t kk = lsh == 0 ? k : k - lsh;
    for (size_t i = 0; i < n; ++i) {
        int64_t d = get_digit(kk, a[i]);
        int64_t cr = get_carry(kk, a[i], d);
        int64_t dpc = wadd(wshl(d, lsh), c[i]);
        int64_t x1 = get_digit(k, dpc);
        x[i] = x1;
        c[i] = wadd(cr, get_carry(k, dpc, x1));
    }
}

/* normalization.rs:254-272 */
static void nz_final_step_assign(size_t k, size_t lsh, int64_t* x, int64_t* c, size_t n) {
    size_t kk = lsh == 0 ? k : k - lsh;
    for (size_t i = 0; i < n; ++i) x[i] = get_digit(k, wadd(wshl(get_digit(kk, x[i]), lsh), c[i]));
}

/* znx/mul.rs:30-49 */
static void znx_mul_power_of_two_assign(int64_t kp, int64_t* x, size_t n) {
    if (kp == 0) return;
    if (kp > 0) {
        for (size_t i = 0; i < n; ++i) x[i] = wshl(x[i], (size_t)kp);
        return;
    }
    size_t k = (size_t)(-kp);
    for (size_t i = 0; i < n; ++i) {
        int64_t sign_bit = (x[i] >> 63) & 1;
        int64_t bias = ((int64_t)1 << (k - 1)) - sign_bit;
        x[i] = wadd(x[i], bias) >> k;
    }
}

static void znx_add_assign(int64_t* r, const int64_t* a, size_t n) {
    for (size_t i = 0; i < n; ++i) r[i] = wadd(r[i], a[i]);
}

static inline int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------ */
/* reference/vec_znx/normalize.rs                                            */
/* ------------------------------------------------------------------------ */

size_t pzr_vec_znx_normalize_tmp_bytes(size_t n) { return 3 * n * sizeof(int64_t); } /* normalize.rs:13-15 */

/* normalize.rs:50-144 */
static void normalize_inter(size_t n, size_t base2k,
                            int64_t* res, size_t res_cols, size_t res_size, int64_t res_offset, size_t res_col,
                            const int64_t* a, size_t a_cols, size_t a_size, size_t a_col, int64_t* carry) {
    int64_t lsh = res_offset % (int64_t)base2k;
    int64_t limbs_offset = res_offset / (int64_t)base2k;
    if (res_offset < 0 && lsh != 0) {
        lsh = (lsh + (int64_t)base2k) % (int64_t)base2k;
        limbs_offset -= 1;
    }
    size_t lsh_pos = (size_t)lsh;
    size_t res_end = (size_t)clampi(-limbs_offset, 0, (int64_t)res_size);
    size_t res_start = (size_t)clampi((int64_t)a_size - limbs_offset, 0, (int64_t)res_size);
    size_t a_end = (size_t)clampi(limbs_offset, 0, (int64_t)a_size);
    size_t a_start = (size_t)clampi((int64_t)res_size + limbs_offset, 0, (int64_t)a_size);
    size_t a_out_range = a_size > a_start ? a_size - a_start : 0;

    for (size_t j = 0; j < a_out_range; ++j) {
        const int64_t* aj = at_ci64(a, n, a_cols, a_col, a_size - j - 1);
        if (j == 0) nz_first_step_carry_only(base2k, lsh_pos, aj, carry, n);
        else nz_middle_step_carry_only(base2k, lsh_pos, aj, carry, n);
    }
    if (a_out_range == 0) memset(carry, 0, n * sizeof(int64_t));
    for (size_t j = res_start; j < res_size; ++j) memset(at_i64(res, n, res_cols, res_col, j), 0, n * sizeof(int64_t));
    size_t mid_range = a_start > a_end ? a_start - a_end : 0;
    for (size_t j = 0; j < mid_range; ++j)
        nz_middle_step(base2k, lsh_pos, at_i64(res, n, res_cols, res_col, res_start - j - 1),
                       at_ci64(a, n, a_cols, a_col, a_start - j - 1), carry, n);
    for (size_t j = 0; j < res_end; ++j) {
        int64_t* r = at_i64(res, n, res_cols, res_col, res_end - j - 1);
        memset(r, 0, n * sizeof(int64_t));
        if (j == res_end - 1) nz_final_step_assign(base2k, lsh_pos, r, carry, n);
        else nz_middle_step_assign(base2k, lsh_pos, r, carry, n);
    }
}

/* normalize.rs:147-401 */
static void normalize_cross(size_t n,
                            int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k, int64_t res_offset, size_t res_col,
                            const int64_t* a, size_t a_cols, size_t a_size, size_t a_base2k, size_t a_col, int64_t* scratch) {
    int64_t* a_norm = scratch;
    int64_t* res_carry = scratch + n;
    int64_t* a_carry = scratch + 2 * n;
    memset(res_carry, 0, n * sizeof(int64_t));

    size_t a_tot_bits = a_size * a_base2k;
    size_t res_tot_bits = res_size * res_base2k;

    int64_t lsh = res_offset % (int64_t)a_base2k;
    int64_t limbs_offset = res_offset / (int64_t)a_base2k;
    if (res_offset < 0 && lsh != 0) {
        lsh = (lsh + (int64_t)a_base2k) % (int64_t)a_base2k;
        limbs_offset -= 1;
    }
    size_t lsh_pos = (size_t)lsh;

    size_t res_end_bit = (size_t)clampi(-limbs_offset * (int64_t)a_base2k, 0, (int64_t)res_tot_bits);
    size_t res_start_bit = (size_t)clampi((int64_t)a_tot_bits - limbs_offset * (int64_t)a_base2k, 0, (int64_t)res_tot_bits);
    size_t a_end_bit = (size_t)clampi(limbs_offset * (int64_t)a_base2k, 0, (int64_t)a_tot_bits);
    size_t a_start_bit = (size_t)clampi((int64_t)res_tot_bits + limbs_offset * (int64_t)a_base2k, 0, (int64_t)a_tot_bits);

    size_t res_end = res_end_bit / res_base2k;
    size_t res_start = (res_start_bit + res_base2k - 1) / res_base2k;
    size_t a_end = a_end_bit / a_base2k;
    size_t a_start = (a_start_bit + a_base2k - 1) / a_base2k;

    for (size_t j = 0; j < res_size; ++j) memset(at_i64(res, n, res_cols, res_col, j), 0, n * sizeof(int64_t));
    if (res_start == 0) return;

    size_t a_out_range = a_size > a_start ? a_size - a_start : 0;
    for (size_t j = 0; j < a_out_range; ++j) {
        const int64_t* aj = at_ci64(a, n, a_cols, a_col, a_size - j - 1);
        if (j == 0) nz_first_step_carry_only(a_base2k, lsh_pos, aj, a_carry, n);
        else nz_middle_step_carry_only(a_base2k, lsh_pos, aj, a_carry, n);
    }
    if (a_out_range == 0) memset(a_carry, 0, n * sizeof(int64_t));

    size_t res_acc_left = res_base2k;
    size_t res_limb = res_start - 1;
    size_t mid_range = a_start > a_end ? a_start - a_end : 0;

    for (size_t j = 0; j < mid_range; ++j) {
        size_t a_limb = a_start - j - 1;
        const int64_t* a_slice = at_ci64(a, n, a_cols, a_col, a_limb);
        size_t a_take_left = a_base2k;
        nz_middle_step(a_base2k, lsh_pos, a_norm, a_slice, a_carry, n);
        if (j == 0) {
            if ((a_tot_bits - a_start_bit) % a_base2k != 0) {
                size_t take = (a_tot_bits - a_start_bit) % a_base2k;
                znx_mul_power_of_two_assign(-(int64_t)take, a_norm, n);
                a_take_left -= take;
            } else if ((res_tot_bits - res_start_bit) % res_base2k != 0) {
                res_acc_left -= (res_tot_bits - res_start_bit) % res_base2k;
            }
        }
        int done = 0;
        for (;;) { /* 'inner: normalize.rs:299-370 */
            int64_t* res_slice = at_i64(res, n, res_cols, res_col, res_limb);
            size_t a_take = zmin(zmin(a_base2k, a_take_left), res_acc_left);
            if (a_take != 0) {
                size_t scale = res_base2k - res_acc_left;
                nz_extract_digit_addmul(a_take, scale, res_slice, a_norm, n);
                a_take_left -= a_take;
                res_acc_left -= a_take;
            }
            if (res_acc_left == 0 || a_limb == 0) {
                if (a_limb == 0 && a_take_left == 0) { /* normalize.rs:326-355 */
                    znx_add_assign(a_carry, a_norm, n);
                    if (res_acc_left != 0) {
                        size_t scale = res_base2k - res_acc_left;
                        nz_extract_digit_addmul(res_acc_left, scale, res_slice, a_carry, n);
                    }
                    nz_middle_step_assign(res_base2k, 0, res_slice, res_carry, n);
                    znx_add_assign(res_carry, a_carry, n);
                    done = 1; /* break 'outer */
                    break;
                }
                if (res_limb == 0) { /* normalize.rs:358-360 */
                    done = 1;
                    break;
                }
                res_acc_left += res_base2k;
                res_limb -= 1;
            }
            if (a_take_left == 0) { /* normalize.rs:366-369 */
                znx_add_assign(a_carry, a_norm, n);
                break;
            }
        }
        if (done) break;
    }

    if (res_end != 0) {
        int64_t* carry_to_use = (a_start == a_end) ? a_carry : res_carry;
        for (size_t j = 0; j < res_end; ++j) {
            int64_t* r = at_i64(res, n, res_cols, res_col, res_end - j - 1);
            if (j == res_end - 1) nz_final_step_assign(res_base2k, 0, r, carry_to_use, n);
            else nz_middle_step_assign(res_base2k, 0, r, carry_to_use, n);
        }
    }
}

/* normalize.rs:18-48 */
void pzr_vec_znx_normalize(size_t n,
                           int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k, int64_t res_offset, size_t res_col,
                           const int64_t* a, size_t a_cols, size_t a_size, size_t a_base2k, size_t a_col) {
    int64_t* scratch = (int64_t*)malloc(3 * n * sizeof(int64_t));
    if (res_base2k == a_base2k)
        normalize_inter(n, res_base2k, res, res_cols, res_size, res_offset, res_col, a, a_cols, a_size, a_col, scratch);
    else
        normalize_cross(n, res, res_cols, res_size, res_base2k, res_offset, res_col, a, a_cols, a_size, a_base2k, a_col, scratch);
    free(scratch);
}

/* vec_znx_big.rs:122-138 -> vec_znx/add.rs:60-82 */
void pzr_vec_znx_big_add_small_assign(size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                      const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t sum_size = zmin(a_size, res_size);
    for (size_t j = 0; j < sum_size; ++j) znx_add_assign(at_i64(res, n, res_cols, res_col, j), at_ci64(a, n, a_cols, a_col, j), n);
}

/* reference/znx/automorphism.rs:1-17: res[(i*p) mod 2n] = a[i], negated when the index wraps past n.
 * Sequential scatter exactly as the reference (so an even p overwrites the same way). */
static void znx_automorphism(int64_t p, int64_t* res, const int64_t* a, size_t n) {
    size_t mask = 2 * n - 1;
    size_t p_2n = (size_t)(p & (int64_t)mask);
    size_t k = 0;
    res[0] = a[0];
    for (size_t i = 1; i < n; ++i) {
        k = (k + p_2n) & mask;
        if (k < n) res[k] = a[i];
        else res[k - n] = (int64_t)(0 - (uint64_t)a[i]);
    }
}

/* reference/vec_znx/automorphism.rs:10-35 (also vec_znx_big_automorphism, fft64/vec_znx_big.rs:144-170: same
 * function on the i64 big container) */
void pzr_vec_znx_automorphism(size_t n, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                              const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t min_size = zmin(res_size, a_size);
    for (size_t j = 0; j < min_size; ++j) znx_automorphism(p, at_i64(res, n, res_cols, res_col, j), at_ci64(a, n, a_cols, a_col, j), n);
    for (size_t j = min_size; j < res_size; ++j) memset(at_i64(res, n, res_cols, res_col, j), 0, n * sizeof(int64_t));
}

/* reference/vec_znx/automorphism.rs:37-51 (tmp = one polynomial) */
void pzr_vec_znx_automorphism_assign(size_t n, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col) {
    int64_t* tmp = (int64_t*)malloc(n * sizeof(int64_t));
    for (size_t j = 0; j < res_size; ++j) {
        int64_t* r = at_i64(res, n, res_cols, res_col, j);
        znx_automorphism(p, tmp, r, n);
        memcpy(r, tmp, n * sizeof(int64_t));
    }
    free(tmp);
}

/* fft64/vec_znx_big.rs:499-516 -> vec_znx/sub.rs:60-82 : res -= a over min sizes */
void pzr_vec_znx_big_sub_small_assign(size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                      const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t sum_size = zmin(a_size, res_size);
    for (size_t j = 0; j < sum_size; ++j) {
        int64_t* r = at_i64(res, n, res_cols, res_col, j);
        const int64_t* x = at_ci64(a, n, a_cols, a_col, j);
        for (size_t i = 0; i < n; ++i) r[i] = (int64_t)((uint64_t)r[i] - (uint64_t)x[i]);
    }
}

/* fft64/vec_znx_big.rs:519-536 -> vec_znx/sub.rs:84-110 : res = a - res over min sizes, res = -res beyond */
void pzr_vec_znx_big_sub_small_negate_assign(size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                             const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t sum_size = zmin(a_size, res_size);
    for (size_t j = 0; j < res_size; ++j) {
        int64_t* r = at_i64(res, n, res_cols, res_col, j);
        if (j < sum_size) {
            const int64_t* x = at_ci64(a, n, a_cols, a_col, j);
            for (size_t i = 0; i < n; ++i) r[i] = (int64_t)((uint64_t)x[i] - (uint64_t)r[i]);
        } else {
            for (size_t i = 0; i < n; ++i) r[i] = (int64_t)(0 - (uint64_t)r[i]);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* i64 VecZnx limb-wise family                           */
/* reference/vec_znx/add.rs:6-109, sub.rs:6-112, negate.rs:6-44, copy.rs, zero.rs; wrapping i64 arithmetic */
/* ------------------------------------------------------------------------ */
static void znx_add_sub(int sub, size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                        const int64_t* a, size_t a_cols, size_t a_size, size_t a_col,
                        const int64_t* b, size_t b_cols, size_t b_size, size_t b_col) {
    int a_le_b = a_size <= b_size;
    size_t sum_size = zmin(a_le_b ? a_size : b_size, res_size);
    size_t cpy_size = zmin(a_le_b ? b_size : a_size, res_size);
    for (size_t j = 0; j < res_size; ++j) {
        int64_t* r = at_i64(res, n, res_cols, res_col, j);
        if (j < sum_size) {
            const int64_t* x = at_ci64(a, n, a_cols, a_col, j);
            const int64_t* y = at_ci64(b, n, b_cols, b_col, j);
            for (size_t i = 0; i < n; ++i)
                r[i] = sub ? (int64_t)((uint64_t)x[i] - (uint64_t)y[i]) : (int64_t)((uint64_t)x[i] + (uint64_t)y[i]);
        } else if (j < cpy_size) {
            if (a_le_b) { /* add.rs:35-37 copy b ; sub.rs:37-39 negate b */
                const int64_t* y = at_ci64(b, n, b_cols, b_col, j);
                for (size_t i = 0; i < n; ++i) r[i] = sub ? (int64_t)(0 - (uint64_t)y[i]) : y[i];
            } else {
                const int64_t* x = at_ci64(a, n, a_cols, a_col, j);
                for (size_t i = 0; i < n; ++i) r[i] = x[i];
            }
        } else {
            memset(r, 0, n * sizeof(int64_t));
        }
    }
}
void pzr_vec_znx_add_into(size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                          const int64_t* a, size_t a_cols, size_t a_size, size_t a_col,
                          const int64_t* b, size_t b_cols, size_t b_size, size_t b_col) {
    znx_add_sub(0, n, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, b, b_cols, b_size, b_col);
}
void pzr_vec_znx_sub(size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                     const int64_t* a, size_t a_cols, size_t a_size, size_t a_col,
                     const int64_t* b, size_t b_cols, size_t b_size, size_t b_col) {
    znx_add_sub(1, n, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, b, b_cols, b_size, b_col);
}
/* add.rs:88-109, sub.rs:60-82 (mode 0 res += a, 1 res -= a over the common limbs), sub.rs:84-112 (mode 2: res = a - res, -res beyond) */
void pzr_vec_znx_assign_op(int mode, size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                           const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t sum_size = zmin(a_size, res_size);
    for (size_t j = 0; j < res_size; ++j) {
        int64_t* r = at_i64(res, n, res_cols, res_col, j);
        if (j < sum_size) {
            const int64_t* x = at_ci64(a, n, a_cols, a_col, j);
            for (size_t i = 0; i < n; ++i) {
                uint64_t rv = (uint64_t)r[i], xv = (uint64_t)x[i];
                r[i] = (int64_t)(mode == 0 ? rv + xv : mode == 1 ? rv - xv : xv - rv);
            }
        } else if (mode == 2) {
            for (size_t i = 0; i < n; ++i) r[i] = (int64_t)(0 - (uint64_t)r[i]);
        }
    }
}
/* negate.rs:6-29 (res = -a, zero tail), :31-44 (in place, a == NULL) */
void pzr_vec_znx_negate(size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                        const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    for (size_t j = 0; j < res_size; ++j) {
        int64_t* r = at_i64(res, n, res_cols, res_col, j);
        if (!a) {
            for (size_t i = 0; i < n; ++i) r[i] = (int64_t)(0 - (uint64_t)r[i]);
        } else if (j < zmin(res_size, a_size)) {
            const int64_t* x = at_ci64(a, n, a_cols, a_col, j);
            for (size_t i = 0; i < n; ++i) r[i] = (int64_t)(0 - (uint64_t)x[i]);
        } else {
            memset(r, 0, n * sizeof(int64_t));
        }
    }
}
/* copy.rs (common limbs, zero tail) */
void pzr_vec_znx_copy(size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                      const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t min_size = zmin(res_size, a_size);
    for (size_t j = 0; j < res_size; ++j) {
        int64_t* r = at_i64(res, n, res_cols, res_col, j);
        if (j < min_size) memcpy(r, at_ci64(a, n, a_cols, a_col, j), n * sizeof(int64_t));
        else memset(r, 0, n * sizeof(int64_t));
    }
}

/* reference/vec_znx/shift.rs:68-135 (vec_znx_lsh, OVERWRITE = true), :16-66 (vec_znx_lsh_assign), :245-342 (vec_znx_rsh,
 * OVERWRITE = true), restated literally with the step functions of znx/normalization.rs.  (They are the same limb walk as
 * vec_znx_normalize with res_offset = +k / -k at equal bases: tests/test_oracle_exact.py P10 pins that equality, and the
 * device exposes them through its normalize kernels.) */
static void nz_final_step(size_t k, size_t lsh, int64_t* x, const int64_t* a, const int64_t* c, size_t n) { /* normalization.rs:274-300 */
    size_t kk = lsh == 0 ? k : k - lsh;
    for (size_t i = 0; i < n; ++i) x[i] = get_digit(k, wadd(wshl(get_digit(kk, a[i]), lsh), c[i]));
}
static void nz_first_step_assign(size_t k, size_t lsh, int64_t* x, int64_t* c, size_t n) { /* normalization.rs:44-65 */
    size_t kk = lsh == 0 ? k : k - lsh;
    for (size_t i = 0; i < n; ++i) {
        int64_t d = get_digit(kk, x[i]);
        c[i] = get_carry(kk, x[i], d);
        x[i] = wshl(d, lsh);
    }
}
void pzr_vec_znx_lsh(size_t n, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                     const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t steps = k / base2k, k_rem = k % base2k;
    if (steps >= (res_size > a_size ? res_size : a_size)) {
        for (size_t j = 0; j < res_size; ++j) memset(at_i64(res, n, res_cols, res_col, j), 0, n * sizeof(int64_t));
        return;
    }
    int64_t* carry = (int64_t*)calloc(n, sizeof(int64_t));
    size_t min_size = zmin(res_size, a_size > steps ? a_size - steps : 0);
    size_t carry_only_start = zmin(steps + min_size, a_size);
    for (size_t j = a_size; j-- > carry_only_start;) {
        if (j == a_size - 1) nz_first_step_carry_only(base2k, k_rem, at_ci64(a, n, a_cols, a_col, j), carry, n);
        else nz_middle_step_carry_only(base2k, k_rem, at_ci64(a, n, a_cols, a_col, j), carry, n);
    }
    if (carry_only_start == a_size) memset(carry, 0, n * sizeof(int64_t));
    for (size_t j = min_size; j-- > 0;) {
        if (j == 0) nz_final_step(base2k, k_rem, at_i64(res, n, res_cols, res_col, j), at_ci64(a, n, a_cols, a_col, j + steps), carry, n);
        else nz_middle_step(base2k, k_rem, at_i64(res, n, res_cols, res_col, j), at_ci64(a, n, a_cols, a_col, j + steps), carry, n);
    }
    for (size_t j = min_size; j < res_size; ++j) memset(at_i64(res, n, res_cols, res_col, j), 0, n * sizeof(int64_t));
    free(carry);
}
void pzr_vec_znx_lsh_assign(size_t n, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col) {
    size_t steps = k / base2k, k_rem = k % base2k;
    if (steps >= res_size) {
        for (size_t j = 0; j < res_size; ++j) memset(at_i64(res, n, res_cols, res_col, j), 0, n * sizeof(int64_t));
        return;
    }
    if (steps > 0) {
        for (size_t j = 0; j < res_size - steps; ++j)
            memcpy(at_i64(res, n, res_cols, res_col, j), at_i64(res, n, res_cols, res_col, j + steps), n * sizeof(int64_t));
        for (size_t j = res_size - steps; j < res_size; ++j) memset(at_i64(res, n, res_cols, res_col, j), 0, n * sizeof(int64_t));
    }
    int64_t* carry = (int64_t*)calloc(n, sizeof(int64_t));
    for (size_t j = res_size - steps; j-- > 0;) {
        int64_t* x = at_i64(res, n, res_cols, res_col, j);
        if (j == res_size - steps - 1) nz_first_step_assign(base2k, k_rem, x, carry, n);
        else if (j == 0) nz_final_step_assign(base2k, k_rem, x, carry, n);
        else nz_middle_step_assign(base2k, k_rem, x, carry, n);
    }
    free(carry);
}
void pzr_vec_znx_rsh(size_t n, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                     const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t steps = k / base2k, k_rem = k % base2k;
    if (k_rem != 0) steps += 1;
    size_t lsh = (base2k - k_rem) % base2k;
    size_t res_end = zmin(res_size, steps);
    size_t res_start = zmin(res_size, a_size + steps);
    size_t a_start = zmin(a_size, res_size > steps ? res_size - steps : 0);
    size_t a_out_range = a_size - a_start;
    int64_t* carry = (int64_t*)calloc(n, sizeof(int64_t));
    for (size_t j = 0; j < a_out_range; ++j) {
        if (j == 0) nz_first_step_carry_only(base2k, lsh, at_ci64(a, n, a_cols, a_col, a_size - j - 1), carry, n);
        else nz_middle_step_carry_only(base2k, lsh, at_ci64(a, n, a_cols, a_col, a_size - j - 1), carry, n);
    }
    if (a_out_range == 0) memset(carry, 0, n * sizeof(int64_t));
    for (size_t j = 0; j < res_size; ++j) memset(at_i64(res, n, res_cols, res_col, j), 0, n * sizeof(int64_t));
    size_t mid_range = res_start > res_end ? res_start - res_end : 0;
    for (size_t j = 0; j < mid_range; ++j)
        nz_middle_step(base2k, lsh, at_i64(res, n, res_cols, res_col, res_start - j - 1), at_ci64(a, n, a_cols, a_col, a_start - j - 1), carry, n);
    for (size_t j = 0; j < res_end; ++j) {
        int64_t* r = at_i64(res, n, res_cols, res_col, res_end - j - 1);
        if (j == res_end - 1) nz_final_step_assign(base2k, lsh, r, carry, n);
        else nz_middle_step_assign(base2k, lsh, r, carry, n);
    }
    free(carry);
}

/* ------------------------------------------------------------------------ */
/* poulpy-core callers                                                       */
/* ------------------------------------------------------------------------ */

static size_t div_ceil(size_t a, size_t b) { return (a + b - 1) / b; }

/* external_product/glwe.rs:99-141 (default) + :197-271 (internal) */
void pzr_glwe_external_product(const pzr_tables* t, size_t rank,
                               int64_t* res, size_t res_size, size_t res_base2k,
                               const int64_t* a, size_t a_size, size_t a_base2k,
                               const double* ggsw_pmat, size_t dnum, size_t ggsw_size, size_t dsize, size_t ggsw_base2k) {
    size_t n = t->m << 1;
    size_t cols = rank + 1;
    int64_t* a_conv = NULL;
    if (a_base2k != ggsw_base2k) { /* glwe.rs:124-132 ; operations/glwe.rs:1286-1310 */
        size_t conv_size = div_ceil(a_size * a_base2k, ggsw_base2k);
        a_conv = (int64_t*)calloc(n * cols * conv_size, sizeof(int64_t));
        for (size_t i = 0; i < cols; ++i)
            pzr_vec_znx_normalize(n, a_conv, cols, conv_size, ggsw_base2k, 0, i, a, cols, a_size, a_base2k, i);
        a = a_conv;
        a_size = conv_size;
    }
    double* res_dft = (double*)calloc(n * cols * ggsw_size, sizeof(double)); /* glwe.rs:121-122 */
    size_t a_dft_max = div_ceil(a_size, dsize);
    double* a_dft = (double*)calloc(n * cols * a_dft_max + 8, sizeof(double)); /* glwe.rs:225-226 */
    size_t res_dft_size = ggsw_size;
    if (dsize == 1) {
        for (size_t j = 0; j < cols; ++j) pzr_vec_znx_dft_apply(t, 1, 0, a_dft, cols, a_size, j, a, cols, a_size, j);
        pzr_vmp_apply_dft_to_dft(n, res_dft, cols, ggsw_size, a_dft, cols, a_size, ggsw_pmat, dnum, cols, cols, ggsw_size, 0);
    } else {
        /* glwe.rs:235-267.  The reference takes res_dft_tmp from scratch un-zeroed and its FFT64
         * vmp core leaves the last limb_offset limbs of the output unwritten, so
         * cpu-ref's dsize > 1 result depends on stale scratch.  The restatement takes the NTT120
         * sibling's (evidently intended) zero-tail semantics: tmp is cleared before every product.
         * Byte parity with cpu-ref is therefore claimed for dsize = 1 only; dsize > 1 is validated
         * against the exact bivariate product (tests/test_oracle_exact.py). */
        double* tmp = (double*)calloc(n * cols * ggsw_size, sizeof(double));
        for (size_t di = 0; di < dsize; ++di) {
            size_t a_sz = (a_size + di) / dsize;
            long drop = (long)(dsize - di) - 2;
            res_dft_size = ggsw_size - (size_t)(drop > 0 ? drop : 0);
            for (size_t j = 0; j < cols; ++j)
                pzr_vec_znx_dft_apply(t, dsize, dsize - 1 - di, a_dft, cols, a_sz, j, a, cols, a_size, j);
            if (di == 0) {
                pzr_vmp_apply_dft_to_dft(n, res_dft, cols, res_dft_size, a_dft, cols, a_sz, ggsw_pmat, dnum, cols, cols, ggsw_size, 0);
            } else {
                memset(tmp, 0, n * cols * ggsw_size * sizeof(double));
                pzr_vmp_apply_dft_to_dft(n, tmp, cols, res_dft_size, a_dft, cols, a_sz, ggsw_pmat, dnum, cols, cols, ggsw_size, di);
                for (size_t c = 0; c < cols; ++c)
                    pzr_vec_znx_dft_add_assign(n, res_dft, cols, res_dft_size, c, tmp, cols, res_dft_size, c);
            }
        }
        free(tmp);
    }
    /* glwe.rs:270: consume with the size set by the last iteration */
    pzr_vec_znx_idft_apply_consume(t, res_dft, cols, res_dft_size);
    const int64_t* res_big = (const int64_t*)res_dft;
    for (size_t j = 0; j < cols; ++j) /* glwe.rs:138-140 */
        pzr_vec_znx_normalize(n, res, cols, res_size, res_base2k, 0, j, res_big, cols, res_dft_size, ggsw_base2k, j);
    free(a_dft);
    free(res_dft);
    free(a_conv);
}

/* keyswitching/glwe.rs:298-380 (gglwe_product_dft): res_dft(cols_out, key_size) = a_dft(cin, a_size) x key, digit by digit for
 * dsize > 1.  res_dft arrives zeroed. */
static void gglwe_product_dft(size_t n, double* res_dft, size_t cols_out, size_t key_size, const double* a_dft, size_t cin, size_t a_size,
                              const double* key_pmat, size_t dnum, size_t dsize) {
    if (dsize == 1) {
        pzr_vmp_apply_dft_to_dft(n, res_dft, cols_out, key_size, a_dft, cin, a_size, key_pmat, dnum, cin, cols_out, key_size, 0);
    } else {
        size_t ai_max = zmin(div_ceil(a_size, dsize), dnum);
        double* ai = (double*)calloc(n * cin * ai_max + 8, sizeof(double));
        double* tmp = (double*)calloc(n * cols_out * key_size, sizeof(double));
        for (size_t di = 0; di < dsize; ++di) {
            size_t ai_sz = zmin((a_size + di) / dsize, dnum);
            long drop = (long)(dsize - di) - 2;
            size_t r_sz = key_size - (size_t)(drop > 0 ? drop : 0);
            for (size_t j = 0; j < cin; ++j)
                pzr_vec_znx_dft_copy(n, dsize, dsize - di - 1, ai, cin, ai_sz, j, a_dft, cin, a_size, j);
            if (di == 0) {
                pzr_vmp_apply_dft_to_dft(n, res_dft, cols_out, r_sz, ai, cin, ai_sz, key_pmat, dnum, cin, cols_out, key_size, 0);
            } else {
                memset(tmp, 0, n * cols_out * key_size * sizeof(double)); /* zero-tail semantics, see above */
                pzr_vmp_apply_dft_to_dft(n, tmp, cols_out, r_sz, ai, cin, ai_sz, key_pmat, dnum, cin, cols_out, key_size, di);
                for (size_t c = 0; c < cols_out; ++c)
                    pzr_vec_znx_dft_add_assign(n, res_dft, cols_out, r_sz, c, tmp, cols_out, r_sz, c);
            }
        }
        free(ai);
        free(tmp);
        /* glwe.rs:378: res.set_size(res.max_size()) */
    }
}

/* keyswitching/glwe.rs:53-109 (default), :207-239 (internal), :298-380 (gglwe_product_dft);
 * mode != PZR_KS_PLAIN: the automorphism family on top of it, automorphism/glwe_ct.rs:51-72 (AUTO: key switch, then the
 * automorphism of the normalized result), :96-140 (ADD), :185-229 (SUB), :231-275 (SUB_NEGATE): automorphism of the big
 * value, +/- a, then normalize. */
static void glwe_keyswitch_core(const pzr_tables* t, size_t rank_in, size_t rank_out,
                                int64_t* res, size_t res_size, size_t res_base2k,
                                const int64_t* a, size_t a_size, size_t a_base2k,
                                const double* key_pmat, size_t dnum, size_t key_size, size_t dsize, size_t key_base2k,
                                int mode, int64_t p, size_t body_col) {
    size_t n = t->m << 1;
    size_t cols_a = rank_in + 1;
    size_t cols_out = rank_out + 1;
    int64_t* a_conv = NULL;
    if (a_base2k != key_base2k) {
        size_t conv_size = div_ceil(a_size * a_base2k, key_base2k);
        a_conv = (int64_t*)calloc(n * cols_a * conv_size, sizeof(int64_t));
        for (size_t i = 0; i < cols_a; ++i)
            pzr_vec_znx_normalize(n, a_conv, cols_a, conv_size, key_base2k, 0, i, a, cols_a, a_size, a_base2k, i);
        a = a_conv;
        a_size = conv_size;
    }
    double* res_dft = (double*)calloc(n * cols_out * key_size, sizeof(double));
    size_t cin = cols_a - 1;
    double* a_dft = (double*)calloc(n * cin * a_size + 8, sizeof(double));
    for (size_t c = 0; c < cin; ++c) /* mask columns only: glwe.rs:231-234 */
        pzr_vec_znx_dft_apply(t, 1, 0, a_dft, cin, a_size, c, a, cols_a, a_size, c + 1);
    gglwe_product_dft(n, res_dft, cols_out, key_size, a_dft, cin, a_size, key_pmat, dnum, dsize);
    pzr_vec_znx_idft_apply_consume(t, res_dft, cols_out, key_size);
    int64_t* res_big = (int64_t*)res_dft;
    /* glwe.rs:237 (body_col = 0); conversion/gglwe_to_ggsw.rs:251 adds it to column `col` instead */
    pzr_vec_znx_big_add_small_assign(n, res_big, cols_out, key_size, body_col, a, cols_a, a_size, 0);
    for (size_t i = 0; i < cols_out; ++i) {
        if (mode == PZR_KS_AUTO_ADD || mode == PZR_KS_AUTO_SUB || mode == PZR_KS_AUTO_SUB_NEGATE) {
            /* glwe_ct.rs:134-137 / :223-226 / :269-272 (a is a_conv when the bases differ: :126-130) */
            pzr_vec_znx_automorphism_assign(n, p, res_big, cols_out, key_size, i);
            if (mode == PZR_KS_AUTO_ADD) pzr_vec_znx_big_add_small_assign(n, res_big, cols_out, key_size, i, a, cols_a, a_size, i);
            else if (mode == PZR_KS_AUTO_SUB) pzr_vec_znx_big_sub_small_assign(n, res_big, cols_out, key_size, i, a, cols_a, a_size, i);
            else pzr_vec_znx_big_sub_small_negate_assign(n, res_big, cols_out, key_size, i, a, cols_a, a_size, i);
        }
        /* glwe.rs:105-108 */
        pzr_vec_znx_normalize(n, res, cols_out, res_size, res_base2k, 0, i, res_big, cols_out, key_size, key_base2k, i);
    }
    if (mode == PZR_KS_AUTO) /* glwe_ct.rs:69-71 */
        for (size_t i = 0; i < cols_out; ++i) pzr_vec_znx_automorphism_assign(n, p, res, cols_out, res_size, i);
    free(a_dft);
    free(res_dft);
    free(a_conv);
}

void pzr_glwe_keyswitch(const pzr_tables* t, size_t rank_in, size_t rank_out,
                        int64_t* res, size_t res_size, size_t res_base2k,
                        const int64_t* a, size_t a_size, size_t a_base2k,
                        const double* key_pmat, size_t dnum, size_t key_size, size_t dsize, size_t key_base2k) {
    glwe_keyswitch_core(t, rank_in, rank_out, res, res_size, res_base2k, a, a_size, a_base2k, key_pmat, dnum, key_size, dsize,
                        key_base2k, PZR_KS_PLAIN, 0, 0);
}

/* automorphism/glwe_ct.rs:51-275: `mode` selects glwe_automorphism / _add / _sub / _sub_negate; the key is the prepared
 * automorphism key (a GGLWE with rank_in = rank_out = rank) and p its Galois element (key.p()). */
void pzr_glwe_automorphism(const pzr_tables* t, size_t rank, int mode, int64_t p,
                           int64_t* res, size_t res_size, size_t res_base2k,
                           const int64_t* a, size_t a_size, size_t a_base2k,
                           const double* key_pmat, size_t dnum, size_t key_size, size_t dsize, size_t key_base2k) {
    glwe_keyswitch_core(t, rank, rank, res, res_size, res_base2k, a, a_size, a_base2k, key_pmat, dnum, key_size, dsize,
                        key_base2k, mode, p, 0);
}

/* conversion/gglwe_to_ggsw.rs:116-162 (ggsw_expand_row_default) + :182-268 (ggsw_expand_rows_internal): for every row
 * of the GGSW, column `col` >= 1 is the gglwe product of the mask of res.at(row, 0) with tsk.at(col - 1), plus the body
 * of res.at(row, 0) added to column `col` of the big value, normalized into res.at(row, col).  That is the key-switch
 * core with the body landing in column `col`; a_dft / a_0 (:140-158) are its a_conv / DFT of the mask, including the
 * cross-base case (:152-158).  ggsw: MatZnx(rows = dnum, cols_in = cols, cols_out = cols, size); keys[c]: the prepared
 * GGLWE tsk.at(c) (rank -> rank), c in [0, rank). */
void pzr_ggsw_expand_row(const pzr_tables* t, size_t rank, int64_t* ggsw, size_t dnum, size_t size, size_t base2k,
                         const double* const* keys, size_t key_dnum, size_t key_size, size_t dsize, size_t key_base2k) {
    size_t n = t->m << 1;
    size_t cols = rank + 1;
    size_t ct = n * cols * size;
    for (size_t row = 0; row < dnum; ++row) {
        const int64_t* a = ggsw + (row * cols) * ct;
        for (size_t col = 1; col < cols; ++col)
            glwe_keyswitch_core(t, rank, rank, ggsw + (row * cols + col) * ct, size, base2k, a, size, base2k, keys[col - 1],
                                key_dnum, key_size, dsize, key_base2k, PZR_KS_PLAIN, 0, col);
    }
}

/* conversion/gglwe_to_ggsw.rs:32-61 (ggsw_from_gglwe_default): res.at(row, 0) <- a.at(row, 0) (glwe_copy,
 * api/operations.rs:557-577: vec_znx_copy per column = the common limbs, zero tail), then ggsw_expand_row.
 * a: the GGLWE, MatZnx(rows = dnum, cols_in = a_cols_in, cols_out = rank+1, a_size). */
void pzr_ggsw_from_gglwe(const pzr_tables* t, size_t rank, int64_t* ggsw, size_t dnum, size_t size, size_t base2k,
                         const int64_t* a, size_t a_cols_in, size_t a_size,
                         const double* const* keys, size_t key_dnum, size_t key_size, size_t dsize, size_t key_base2k) {
    size_t n = t->m << 1;
    size_t cols = rank + 1;
    size_t min_size = zmin(size, a_size);
    for (size_t row = 0; row < dnum; ++row) {
        int64_t* r = ggsw + (row * cols) * (n * cols * size);
        const int64_t* src = a + (row * a_cols_in) * (n * cols * a_size);
        memcpy(r, src, n * cols * min_size * sizeof(int64_t));
        memset(r + n * cols * min_size, 0, n * cols * (size - min_size) * sizeof(int64_t));
    }
    pzr_ggsw_expand_row(t, rank, ggsw, dnum, size, base2k, keys, key_dnum, key_size, dsize, key_base2k);
}

/* ------------------------------------------------------------------------ */
/* poulpy-bin-fhe blind rotation (CGGI), SURVEY.md 8f rank 2                  */
/* ------------------------------------------------------------------------ */

/* reference/znx/rotate.rs:3-27: res = X^p * src in Z[X]/(X^n+1) */
static void znx_rotate(int64_t p, int64_t* res, const int64_t* src, size_t n) {
    size_t mp_2n = (size_t)(p & (int64_t)(2 * n - 1));
    size_t mp_1n = mp_2n & (n - 1);
    size_t mp_1n_neg = n - mp_1n;
    int neg_first = mp_2n < n;
    /* dst1 = res[0..mp_1n] <- src2 = src[mp_1n_neg..n] ; dst2 = res[mp_1n..n] <- src1 = src[0..mp_1n_neg] */
    for (size_t i = 0; i < mp_1n; ++i) {
        int64_t v = src[mp_1n_neg + i];
        res[i] = neg_first ? (int64_t)(0 - (uint64_t)v) : v;
    }
    for (size_t i = 0; i < mp_1n_neg; ++i) {
        int64_t v = src[i];
        res[mp_1n + i] = neg_first ? v : (int64_t)(0 - (uint64_t)v);
    }
}

/* reference/vec_znx/rotate.rs:10-36 */
void pzr_vec_znx_rotate(size_t n, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                        const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    size_t min_size = zmin(res_size, a_size);
    for (size_t j = 0; j < min_size; ++j) znx_rotate(p, at_i64(res, n, res_cols, res_col, j), at_ci64(a, n, a_cols, a_col, j), n);
    for (size_t j = min_size; j < res_size; ++j) memset(at_i64(res, n, res_cols, res_col, j), 0, n * sizeof(int64_t));
}

/* reference/vec_znx/mul_xp_minus_one.rs:23-37: res = (X^p - 1) * res, limb by limb through one polynomial of scratch */
void pzr_vec_znx_mul_xp_minus_one_assign(size_t n, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col) {
    int64_t* tmp = (int64_t*)malloc(n * sizeof(int64_t));
    for (size_t j = 0; j < res_size; ++j) {
        int64_t* r = at_i64(res, n, res_cols, res_col, j);
        znx_rotate(p, tmp, r, n);
        for (size_t i = 0; i < n; ++i) r[i] = (int64_t)((uint64_t)tmp[i] - (uint64_t)r[i]); /* znx_sub_negate_assign */
    }
    free(tmp);
}

/* reference/vec_znx/normalize.rs:403-425 (in place, same base) ; first step: znx/normalization.rs:44-65 */
void pzr_vec_znx_normalize_assign(size_t n, size_t base2k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col) {
    int64_t* carry = (int64_t*)calloc(n, sizeof(int64_t));
    for (size_t jj = res_size; jj-- > 0;) {
        int64_t* x = at_i64(res, n, res_cols, res_col, jj);
        if (jj == res_size - 1) {
            for (size_t i = 0; i < n; ++i) {
                int64_t d = get_digit(base2k, x[i]);
                carry[i] = get_carry(base2k, x[i], d);
                x[i] = d;
            }
        } else if (jj == 0) {
            nz_final_step_assign(base2k, 0, x, carry, n);
        } else {
            nz_middle_step_assign(base2k, 0, x, carry, n);
        }
    }
    free(carry);
}

/* blind_rotation/utils.rs:6-40 + algorithms/cggi/key_prepared.rs:66-74: x_pow_a[i] = svp_prepare(X^i), i in [0, 2n),
 * with X^i = -X^(i-n) for i >= n.  out: 2n SvpPPol(cols = 1), n doubles each. */
void pzr_blind_rotation_x_pow_a(const pzr_tables* t, double* out) {
    size_t n = t->m << 1;
    int64_t* buf = (int64_t*)calloc(n, sizeof(int64_t));
    for (size_t ai = 0; ai < 2 * n; ++ai) {
        if (ai < n) buf[ai] = 1;
        else buf[(ai - n) & (n - 1)] = -1;
        pzr_svp_prepare(t, out + ai * n, 1, 0, buf, 1, 0);
        if (ai < n) buf[ai] = 0;
        else buf[(ai - n) & (n - 1)] = 0;
    }
    free(buf);
}

/* blind_rotation/algorithms/cggi/algorithm.rs: execute_block_binary (:265-368) when block_size > 1, execute_standard
 * (:370-440) when block_size == 1 (extension_factor = 1).  lwe_2n = the n_lwe+1 values produced by mod_switch_2n
 * (algorithms/mod.rs:136-171: [b, a_1..a_n_lwe]); lut = VecZnx(1, lut_size); brk = n_lwe prepared GGSWs
 * (VmpPMat rows = dnum, cols_in = cols_out = rank+1, size = brk_size), contiguous; res = GLWE(rank+1, res_size). */
void pzr_blind_rotation_execute(const pzr_tables* t, size_t rank, size_t n_lwe, size_t block_size,
                                int64_t* res, size_t res_size, size_t base2k,
                                const int64_t* lwe_2n, const int64_t* lut, size_t lut_size,
                                const double* brk, size_t dnum, size_t brk_size, const double* x_pow_a) {
    size_t n = t->m << 1;
    size_t cols = rank + 1;
    size_t two_n = 2 * n;
    size_t pmat_doubles = n * dnum * cols * cols * brk_size;
    const int64_t* a = lwe_2n + 1;
    int64_t b = lwe_2n[0];
    memset(res, 0, n * cols * res_size * sizeof(int64_t));                /* :298 / :413 out_mut.data_mut().zero() */
    pzr_vec_znx_rotate(n, b, res, cols, res_size, 0, lut, 1, lut_size, 0); /* :301 / :416 */
    if (block_size > 1) {
        double* acc_dft = (double*)calloc(n * cols * dnum, sizeof(double));
        double* vmp_res = (double*)calloc(n * cols * brk_size, sizeof(double));
        double* acc_add_dft = (double*)calloc(n * cols * brk_size, sizeof(double));
        double* vmp_xai = (double*)calloc(n * brk_size, sizeof(double));
        int64_t* acc_add_big = (int64_t*)calloc(n * brk_size, sizeof(int64_t));
        for (size_t blk = 0; blk + block_size <= n_lwe; blk += block_size) { /* chunks_exact: a trailing partial block is dropped */
            for (size_t j = 0; j < cols; ++j) { /* :319-322 */
                pzr_vec_znx_dft_apply(t, 1, 0, acc_dft, cols, dnum, j, res, cols, res_size, j);
                pzr_vec_znx_dft_zero(n, acc_add_dft, cols, brk_size, j);
            }
            for (size_t k = 0; k < block_size; ++k) { /* :324-337 */
                size_t idx = blk + k;
                size_t ai_pos = (size_t)((a[idx] + (int64_t)two_n) & (int64_t)(two_n - 1));
                pzr_vmp_apply_dft_to_dft(n, vmp_res, cols, brk_size, acc_dft, cols, dnum, brk + idx * pmat_doubles, dnum, cols, cols, brk_size, 0);
                for (size_t i = 0; i < cols; ++i) {
                    pzr_svp_apply_dft_to_dft(n, vmp_xai, 1, brk_size, 0, x_pow_a + ai_pos * n, 1, 0, vmp_res, cols, brk_size, i);
                    pzr_vec_znx_dft_add_assign(n, acc_add_dft, cols, brk_size, i, vmp_xai, 1, brk_size, 0);
                    pzr_vec_znx_dft_sub_assign(n, acc_add_dft, cols, brk_size, i, vmp_res, cols, brk_size, i);
                }
            }
            for (size_t i = 0; i < cols; ++i) { /* :342-346 */
                pzr_vec_znx_idft_apply(t, acc_add_big, 1, brk_size, 0, acc_add_dft, cols, brk_size, i);
                pzr_vec_znx_big_add_small_assign(n, acc_add_big, 1, brk_size, 0, res, cols, res_size, i);
                pzr_vec_znx_normalize(n, res, cols, res_size, base2k, 0, i, acc_add_big, 1, brk_size, base2k, 0);
            }
        }
        free(acc_dft);
        free(vmp_res);
        free(acc_add_dft);
        free(vmp_xai);
        free(acc_add_big);
    } else {
        int64_t* acc_tmp = (int64_t*)calloc(n * cols * res_size, sizeof(int64_t));
        for (size_t idx = 0; idx < n_lwe; ++idx) { /* :423-433 */
            pzr_glwe_external_product(t, rank, acc_tmp, res_size, base2k, res, res_size, base2k, brk + idx * pmat_doubles, dnum, brk_size, 1, base2k);
            for (size_t i = 0; i < cols; ++i) pzr_vec_znx_mul_xp_minus_one_assign(n, a[idx], acc_tmp, cols, res_size, i);
            for (size_t i = 0; i < cols; ++i) pzr_vec_znx_big_add_small_assign(n, res, cols, res_size, i, acc_tmp, cols, res_size, i); /* glwe_add_assign */
        }
        for (size_t i = 0; i < cols; ++i) pzr_vec_znx_normalize_assign(n, base2k, res, cols, res_size, i); /* :437 */
        free(acc_tmp);
    }
}

/* blind_rotation/algorithms/cggi/algorithm.rs:121-273 (execute_block_binary_extended): extension_factor `ext` > 1 accumulators
 * acc[0..ext) of ring degree n hold the table of domain n*ext; lwe_2n = mod_switch_2n(2*n*ext) output; lut = the ext
 * polynomials lut.data[j], each VecZnx(1, lut_size), contiguous; res = acc[0] at the end (:270-272).  The skipped updates of
 * :217, :233, :244 (a multiplier that would be X^0) are part of the reference's behaviour and are restated as they are. */
void pzr_blind_rotation_execute_extended(const pzr_tables* t, size_t rank, size_t n_lwe, size_t block_size, size_t ext,
                                         int64_t* res, size_t res_size, size_t base2k,
                                         const int64_t* lwe_2n, const int64_t* lut, size_t lut_size,
                                         const double* brk, size_t dnum, size_t brk_size, const double* x_pow_a) {
    size_t n = t->m << 1, cols = rank + 1, two_n = 2 * n, two_n_ext = 2 * n * ext;
    size_t pmat_doubles = n * dnum * cols * cols * brk_size;
    size_t ct = n * cols * res_size, ctd = n * cols * dnum, ctb = n * cols * brk_size;
    int64_t* acc = (int64_t*)calloc(ext * ct, sizeof(int64_t)); /* :159-161 zero */
    double* acc_dft = (double*)calloc(ext * ctd, sizeof(double));
    double* vmp_res = (double*)calloc(ext * ctb, sizeof(double));
    double* acc_add_dft = (double*)calloc(ext * ctb, sizeof(double));
    double* vmp_xai = (double*)calloc(n * brk_size, sizeof(double));
    int64_t* acc_add_big = (int64_t*)calloc(n * brk_size, sizeof(int64_t));
    const int64_t* a = lwe_2n + 1;
    size_t b_pos = (size_t)((lwe_2n[0] + (int64_t)two_n_ext) & (int64_t)(two_n_ext - 1)); /* :180 */
    size_t b_hi = b_pos / ext, b_lo = b_pos & (ext - 1);
    for (size_t i = 0; i < b_lo; ++i) /* :185-187 */
        pzr_vec_znx_rotate(n, (int64_t)b_hi + 1, acc + i * ct, cols, res_size, 0, lut + (ext - b_lo + i) * n * lut_size, 1, lut_size, 0);
    for (size_t i = b_lo; i < ext; ++i) /* :188-190 */
        pzr_vec_znx_rotate(n, (int64_t)b_hi, acc + i * ct, cols, res_size, 0, lut + (i - b_lo) * n * lut_size, 1, lut_size, 0);
    for (size_t blk = 0; blk + block_size <= n_lwe; blk += block_size) {
        for (size_t i = 0; i < ext; ++i)
            for (size_t j = 0; j < cols; ++j) { /* :195-200 */
                pzr_vec_znx_dft_apply(t, 1, 0, acc_dft + i * ctd, cols, dnum, j, acc + i * ct, cols, res_size, j);
                pzr_vec_znx_dft_zero(n, acc_add_dft + i * ctb, cols, brk_size, j);
            }
        for (size_t kk = 0; kk < block_size; ++kk) {
            size_t idx = blk + kk;
            size_t ai_pos = (size_t)((a[idx] + (int64_t)two_n_ext) & (int64_t)(two_n_ext - 1));
            size_t ai_hi = ai_pos / ext, ai_lo = ai_pos & (ext - 1);
            for (size_t i = 0; i < ext; ++i) /* :209-211 */
                pzr_vmp_apply_dft_to_dft(n, vmp_res + i * ctb, cols, brk_size, acc_dft + i * ctd, cols, dnum, brk + idx * pmat_doubles, dnum, cols,
                                         cols, brk_size, 0);
#define PZR_EXT_UPD(I_, J_, HI_)                                                                                              \
    for (size_t c_ = 0; c_ < cols; ++c_) {                                                                                    \
        pzr_svp_apply_dft_to_dft(n, vmp_xai, 1, brk_size, 0, x_pow_a + (HI_) * n, 1, 0, vmp_res + (J_) * ctb, cols, brk_size, c_); \
        pzr_vec_znx_dft_add_assign(n, acc_add_dft + (I_) * ctb, cols, brk_size, c_, vmp_xai, 1, brk_size, 0);                 \
        pzr_vec_znx_dft_sub_assign(n, acc_add_dft + (I_) * ctb, cols, brk_size, c_, vmp_res + (I_) * ctb, cols, brk_size, c_); \
    }
            if (ai_lo == 0) { /* :214-226 */
                if (ai_hi != 0)
                    for (size_t j = 0; j < ext; ++j) PZR_EXT_UPD(j, j, ai_hi)
            } else {
                if (((ai_hi + 1) & (two_n - 1)) != 0) /* :233-241 */
                    for (size_t i = 0; i < ai_lo; ++i) PZR_EXT_UPD(i, ext - ai_lo + i, ai_hi + 1)
                if (ai_hi != 0) /* :244-253 */
                    for (size_t i = ai_lo; i < ext; ++i) PZR_EXT_UPD(i, i - ai_lo, ai_hi)
            }
#undef PZR_EXT_UPD
        }
        for (size_t j = 0; j < ext; ++j)
            for (size_t i = 0; i < cols; ++i) { /* :260-266 */
                pzr_vec_znx_idft_apply(t, acc_add_big, 1, brk_size, 0, acc_add_dft + j * ctb, cols, brk_size, i);
                pzr_vec_znx_big_add_small_assign(n, acc_add_big, 1, brk_size, 0, acc + j * ct, cols, res_size, i);
                pzr_vec_znx_normalize(n, acc + j * ct, cols, res_size, base2k, 0, i, acc_add_big, 1, brk_size, base2k, 0);
            }
    }
    memcpy(res, acc, ct * sizeof(int64_t)); /* :270-272 */
    free(acc); free(acc_dft); free(vmp_res); free(acc_add_dft); free(vmp_xai); free(acc_add_big);
}

/* ------------------------------------------------------------------------ */
/* glwe_trace (circuit bootstrapping around the blind rotation) */
/* ------------------------------------------------------------------------ */

/* reference/vec_znx/shift.rs:186-243 : res >>= k bits (in place, normalizing), through the shifted normalization steps */
void pzr_vec_znx_rsh_assign(size_t n, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col) {
    size_t size = res_size;
    size_t steps = k / base2k;
    size_t k_rem = k % base2k;
    if (k % base2k != 0) steps += 1; /* rsh by one more limb, then lsh by base2k - k_rem inside the steps */
    size_t lsh = (base2k - k_rem) % base2k;
    int64_t* carry = (int64_t*)calloc(n, sizeof(int64_t));
    int64_t* tmp = (int64_t*)malloc(n * sizeof(int64_t));
    if (steps > size) steps = size; /* (the reference indexes res.at(size - j - 1): k beyond the precision is not a supported call) */
    for (size_t j = 0; j < steps; ++j) {
        const int64_t* x = at_ci64(res, n, res_cols, res_col, size - j - 1);
        if (j == 0) nz_first_step_carry_only(base2k, lsh, x, carry, n);
        else nz_middle_step_carry_only(base2k, lsh, x, carry, n);
    }
    for (size_t j = 0; j + steps < size; ++j) {
        memcpy(tmp, at_ci64(res, n, res_cols, res_col, size - steps - j - 1), n * sizeof(int64_t));
        nz_middle_step_assign(base2k, lsh, tmp, carry, n);
        memcpy(at_i64(res, n, res_cols, res_col, size - j - 1), tmp, n * sizeof(int64_t));
    }
    for (size_t j = 0; j < steps; ++j) {
        memset(at_i64(res, n, res_cols, res_col, j), 0, n * sizeof(int64_t));
        int64_t* x = at_i64(res, n, res_cols, res_col, steps - j - 1);
        if (j == 0) nz_final_step_assign(base2k, lsh, x, carry, n);
        else nz_middle_step_assign(base2k, lsh, x, carry, n);
    }
    free(carry);
    free(tmp);
}

/* glwe_trace.rs:129-176 (same base2k for res and keys): for every step  res = rsh(res, 1);  res = automorphism_add_assign(res, key_p)
 * (:164-174).  The caller resolves the Galois elements p (i = 0: -1, else galois_element(2^(i-1))) and the matching prepared keys:
 * gals[s] / key_pmats[s] for the steps skip..log_n in order. */
void pzr_glwe_trace_assign(const pzr_tables* t, size_t rank, int64_t* res, size_t res_size, size_t base2k,
                           size_t nsteps, const int64_t* gals, const double* const* key_pmats,
                           size_t dnum, size_t key_size, size_t dsize) {
    size_t n = t->m << 1;
    size_t cols = rank + 1;
    int64_t* a = (int64_t*)malloc(n * cols * res_size * sizeof(int64_t));
    for (size_t s = 0; s < nsteps; ++s) {
        for (size_t c = 0; c < cols; ++c) pzr_vec_znx_rsh_assign(n, base2k, 1, res, cols, res_size, c); /* glwe_rsh(1, res), operations/glwe.rs:1096-1112 */
        memcpy(a, res, n * cols * res_size * sizeof(int64_t));
        pzr_glwe_automorphism(t, rank, PZR_KS_AUTO_ADD, gals[s], res, res_size, base2k, a, res_size, base2k, key_pmats[s], dnum, key_size,
                              dsize, base2k);
    }
    free(a);
}

/* poulpy-bin-fhe/src/circuit_bootstrapping/circuit.rs:219-370 (circuit_bootstrap_core) with to_exponent = false and one
 * base2k for the blind-rotation key, the automorphism keys, the tensor keys and the result, i.e. the `copy` branches of
 * :326-327 and of glwe_trace (poulpy-core/src/glwe_trace.rs:114-115, :121-122).  The lookup table (:274-301, host code
 * of lookup_table.rs) and gap = 2*lut.drift/extension_factor (:333) are inputs.
 *   :321-331  acc = blind_rotation(lwe, lut) ; copy into the atk layout (same limbs)
 *   :344-366  row i of the GGSW, column 0 = glwe_trace(acc, skip = 0) truncated to res_size limbs; acc = X^-gap * acc
 *   :369      ggsw_expand_row */
void pzr_circuit_bootstrap_to_constant(const pzr_tables* t, size_t rank, size_t base2k,
                                       size_t n_lwe, size_t block_size, const int64_t* lwe_2n, const int64_t* lut, size_t lut_size,
                                       const double* brk, size_t brk_dnum, size_t brk_size, size_t glwe_size, const double* x_pow_a,
                                       size_t nsteps, const int64_t* gals, const double* const* atk, size_t atk_dnum, size_t atk_size,
                                       int64_t* ggsw, size_t res_dnum, size_t res_size, size_t gap,
                                       const double* const* tsk, size_t tsk_dnum, size_t tsk_size) {
    size_t bases[4] = {base2k, base2k, base2k, base2k};
    size_t tmp_size = glwe_size > res_size ? glwe_size : res_size; /* glwe_trace.rs:107-112: k = max(a.k, res.k) */
    pzr_circuit_bootstrap_bases(t, rank, bases, 0, n_lwe, block_size, lwe_2n, lut, lut_size, brk, brk_dnum, brk_size, glwe_size, glwe_size,
                                tmp_size, x_pow_a, nsteps, gals, atk, atk_dnum, atk_size, ggsw, res_dnum, res_size, gap, 0, 0, 0, tsk, tsk_dnum,
                                tsk_size);
}

/* ------------------------------------------------------------------------ */
/* poulpy-core/src/glwe_packing.rs: pack_internal :15-87, glwe_pack_default :122-176                                  */
/* One base2k and one size for the ciphertexts, the keys' output and the result.  slots[j] (j < n) is the GLWE at index j   */
/* of the reference's HashMap or NULL; present entries are modified in place exactly as the reference's `&mut` entries.   */
/* gals[i] / keys[i], i < log_n: the automorphism keys of trace step i (i = 0: -1, else galois_element(2^(i-1))).        */
/* ------------------------------------------------------------------------ */
static void glwe_rsh1(size_t n, size_t cols, size_t size, size_t base2k, int64_t* x) {
    for (size_t c = 0; c < cols; ++c) pzr_vec_znx_rsh_assign(n, base2k, 1, x, cols, size, c); /* operations/glwe.rs:1096-1112 */
}
static void glwe_rotate_to(size_t n, size_t cols, size_t size, int64_t k, int64_t* res, const int64_t* a) {
    for (size_t c = 0; c < cols; ++c) pzr_vec_znx_rotate(n, k, res, cols, size, c, a, cols, size, c);
}
void pzr_glwe_pack(const pzr_tables* t, size_t rank, int64_t* res, int64_t** slots, size_t size, size_t base2k, size_t log_gap_out,
                   const int64_t* gals, const double* const* keys, size_t dnum, size_t key_size) {
    pzr_glwe_pack_bases(t, rank, res, slots, size, base2k, base2k, size, log_gap_out, gals, keys, dnum, key_size);
}

/* the same with the automorphism keys in their own base (test_suite/glwe_packing.rs:40-42: ciphertexts and result base2k - 1, keys
 * base2k): pack_internal's arithmetic stays in the ciphertexts' base, the automorphisms convert (automorphism/glwe_ct.rs), the closing
 * glwe_trace (glwe_trace.rs:91-127) runs on a temporary of trace_size = ceil(max(a.k, res.k) / key_base2k) limbs in the keys' base */
void pzr_glwe_pack_bases(const pzr_tables* t, size_t rank, int64_t* res, int64_t** slots, size_t size, size_t base2k, size_t key_base2k,
                         size_t trace_size, size_t log_gap_out, const int64_t* gals, const double* const* keys, size_t dnum, size_t key_size) {
    size_t n = t->m << 1, cols = rank + 1, ct = n * cols * size;
    size_t log_n = 0;
    while (((size_t)1 << log_n) < n) ++log_n;
    int64_t* tmp_b = (int64_t*)malloc(ct * sizeof(int64_t));
    int64_t* tmp = (int64_t*)malloc(ct * sizeof(int64_t));
    for (size_t i = 0; i + log_gap_out < log_n; ++i) { /* :156 */
        size_t tt = (size_t)1 << (log_n - 1 - i);      /* :157 and pack_internal :39 */
        for (size_t j = 0; j < tt; ++j) {
            int64_t* a = slots[j];
            int64_t* b = slots[j + tt];
            slots[j] = NULL;
            slots[j + tt] = NULL;
            if (a) {
                if (b) { /* :41-70 */
                    glwe_rotate_to(n, cols, size, -(int64_t)tt, tmp, a);           /* a = a * X^-t */
                    memcpy(a, tmp, ct * sizeof(int64_t));
                    for (size_t c = 0; c < cols; ++c) pzr_vec_znx_sub(n, tmp_b, cols, size, c, a, cols, size, c, b, cols, size, c);
                    glwe_rsh1(n, cols, size, base2k, tmp_b);
                    for (size_t c = 0; c < cols; ++c) pzr_vec_znx_assign_op(0, n, a, cols, size, c, b, cols, size, c); /* a += b */
                    glwe_rsh1(n, cols, size, base2k, a);
                    for (size_t c = 0; c < cols; ++c) pzr_vec_znx_normalize_assign(n, base2k, tmp_b, cols, size, c);
                    memcpy(tmp, tmp_b, ct * sizeof(int64_t));                         /* tmp_b = phi(tmp_b) (glwe_automorphism_assign) */
                    pzr_glwe_automorphism(t, rank, PZR_KS_AUTO, gals[i], tmp_b, size, base2k, tmp, size, base2k, keys[i], dnum, key_size, 1, key_base2k);
                    for (size_t c = 0; c < cols; ++c) pzr_vec_znx_assign_op(1, n, a, cols, size, c, tmp_b, cols, size, c); /* a -= tmp_b */
                    for (size_t c = 0; c < cols; ++c) pzr_vec_znx_normalize_assign(n, base2k, a, cols, size, c);
                    glwe_rotate_to(n, cols, size, (int64_t)tt, tmp, a);              /* a = a * X^t */
                    memcpy(a, tmp, ct * sizeof(int64_t));
                } else { /* :71-75 */
                    glwe_rsh1(n, cols, size, base2k, a);
                    memcpy(tmp, a, ct * sizeof(int64_t));
                    pzr_glwe_automorphism(t, rank, PZR_KS_AUTO_ADD, gals[i], a, size, base2k, tmp, size, base2k, keys[i], dnum, key_size, 1, key_base2k);
                }
                slots[j] = a; /* :168-169 */
            } else if (b) { /* :76-86 */
                glwe_rotate_to(n, cols, size, (int64_t)tt, tmp_b, b);
                glwe_rsh1(n, cols, size, base2k, tmp_b);
                pzr_glwe_automorphism(t, rank, PZR_KS_AUTO_SUB_NEGATE, gals[i], b, size, base2k, tmp_b, size, base2k, keys[i], dnum, key_size, 1,
                                      key_base2k);
                slots[j] = b; /* :170-171 */
            }
        }
    }
    /* :175 glwe_trace(res, log_n - log_gap_out, a[0]) : copy, trace_assign over the remaining steps, copy (glwe_trace.rs:92-123) */
    if (!slots[0]) { /* the reference panics here (a.get(&0).unwrap()): res is left untouched */
        free(tmp_b);
        free(tmp);
        return;
    }
    size_t skip = log_n - log_gap_out;
    if (key_base2k == base2k) { /* glwe_copy both ways (trace_size == size) */
        memcpy(res, slots[0], ct * sizeof(int64_t));
        pzr_glwe_trace_assign(t, rank, res, size, base2k, log_n - skip, gals + skip, keys + skip, dnum, key_size, 1);
    } else { /* glwe_normalize into the keys' base, trace there, glwe_normalize back */
        int64_t* tr = (int64_t*)calloc(n * cols * trace_size, sizeof(int64_t));
        for (size_t c = 0; c < cols; ++c) pzr_vec_znx_normalize(n, tr, cols, trace_size, key_base2k, 0, c, slots[0], cols, size, base2k, c);
        pzr_glwe_trace_assign(t, rank, tr, trace_size, key_base2k, log_n - skip, gals + skip, keys + skip, dnum, key_size, 1);
        for (size_t c = 0; c < cols; ++c) pzr_vec_znx_normalize(n, res, cols, size, base2k, 0, c, tr, cols, trace_size, key_base2k, c);
        free(tr);
    }
    free(tmp_b);
    free(tmp);
}

/* glwe_trace_assign with res in another base than the keys (glwe_trace.rs:153-163; test_suite/trace.rs:36-39 runs result base2k, keys
 * base2k - 1): res re-expressed in the keys' base on conv_size = ceil(res.max_k / key_base2k) limbs, traced there, normalized back */
void pzr_glwe_trace_assign_bases(const pzr_tables* t, size_t rank, int64_t* res, size_t res_size, size_t res_base2k, size_t conv_size,
                                 size_t key_base2k, size_t nsteps, const int64_t* gals, const double* const* key_pmats,
                                 size_t dnum, size_t key_size, size_t dsize) {
    if (res_base2k == key_base2k) {
        pzr_glwe_trace_assign(t, rank, res, res_size, res_base2k, nsteps, gals, key_pmats, dnum, key_size, dsize);
        return;
    }
    size_t n = t->m << 1, cols = rank + 1;
    int64_t* conv = (int64_t*)calloc(n * cols * conv_size, sizeof(int64_t));
    for (size_t c = 0; c < cols; ++c) pzr_vec_znx_normalize(n, conv, cols, conv_size, key_base2k, 0, c, res, cols, res_size, res_base2k, c);
    pzr_glwe_trace_assign(t, rank, conv, conv_size, key_base2k, nsteps, gals, key_pmats, dnum, key_size, dsize);
    for (size_t c = 0; c < cols; ++c) pzr_vec_znx_normalize(n, res, cols, res_size, res_base2k, 0, c, conv, cols, conv_size, key_base2k, c);
    free(conv);
}


/* poulpy-bin-fhe/src/circuit_bootstrapping/circuit.rs:219-370 with to_exponent = true and post_process :373-421, one base2k,
 * res_size <= glwe_size (so that every intermediate has the GLWE's size).  The lookup table, gap (:333) and log_gap_in (:342)
 * are inputs; gals / atk: all log_n trace steps.
 *   log_gap_in == log_gap_out: res_row = glwe_trace(a, skip = log_n - log_gap_in + 1)                         (:418-420)
 *   otherwise: a_trace = that partial trace; cts[s] = X^(-s 2^log_gap_in) a_trace at index s 2^log_gap_out,
 *              s < 2^log_domain; res_row = glwe_pack(cts, log_gap_out)                                          (:392-417) */
void pzr_circuit_bootstrap_to_exponent(const pzr_tables* t, size_t rank, size_t base2k,
                                       size_t n_lwe, size_t block_size, const int64_t* lwe_2n, const int64_t* lut, size_t lut_size,
                                       const double* brk, size_t brk_dnum, size_t brk_size, size_t glwe_size, const double* x_pow_a,
                                       const int64_t* gals, const double* const* atk, size_t atk_dnum, size_t atk_size,
                                       int64_t* ggsw, size_t res_dnum, size_t res_size, size_t gap,
                                       size_t log_gap_in, size_t log_gap_out, size_t log_domain,
                                       const double* const* tsk, size_t tsk_dnum, size_t tsk_size) {
    size_t bases[4] = {base2k, base2k, base2k, base2k};
    pzr_circuit_bootstrap_bases(t, rank, bases, 1, n_lwe, block_size, lwe_2n, lut, lut_size, brk, brk_dnum, brk_size, glwe_size, glwe_size,
                                glwe_size, x_pow_a, 0, gals, atk, atk_dnum, atk_size, ggsw, res_dnum, res_size, gap, log_gap_in, log_gap_out,
                                log_domain, tsk, tsk_dnum, tsk_size);
}

/* circuit.rs:219-370 (+ post_process :373-421) with one base2k per object, as the reference's tests run it
 * (circuit_bootstrapping/tests/circuit_bootstrapping.rs:49-53).  bases = {brk, atk, tsk, res}.
 *   :321-331  acc = blind_rotation(lwe, lut) in the brk base (glwe_size limbs); glwe_copy / glwe_normalize (operations/glwe.rs:1286-1310)
 *             into the atk layout (atk_glwe_size limbs)
 *   :344-366  per row: glwe_trace (poulpy-core/src/glwe_trace.rs:91-127: temporary of trace_size limbs in the atk base = zero-extended
 *             copy of a, trace_assign, then glwe_copy or glwe_normalize into the row) or post_process; acc = X^-gap * acc
 *   :369      ggsw_expand_row (res base, tsk base)
 * constant mode: gals / atk = the nsteps steps of the full trace; exponent mode: all log2(n) steps (nsteps ignored). */
static void cbt_finish_row(size_t n, size_t cols, int64_t* row, size_t res_size, size_t k_res, const int64_t* tmp, size_t tmp_size, size_t k_atk) {
    if (k_res == k_atk) { /* glwe_copy: common limbs, zero tail */
        size_t common = res_size < tmp_size ? res_size : tmp_size;
        memset(row, 0, n * cols * res_size * sizeof(int64_t));
        memcpy(row, tmp, n * cols * common * sizeof(int64_t));
    } else {
        for (size_t c = 0; c < cols; ++c) pzr_vec_znx_normalize(n, row, cols, res_size, k_res, 0, c, tmp, cols, tmp_size, k_atk, c);
    }
}
void pzr_circuit_bootstrap_bases(const pzr_tables* t, size_t rank, const size_t* bases, int to_exponent,
                                 size_t n_lwe, size_t block_size, const int64_t* lwe_2n, const int64_t* lut, size_t lut_size,
                                 const double* brk, size_t brk_dnum, size_t brk_size, size_t glwe_size, size_t atk_glwe_size, size_t trace_size,
                                 const double* x_pow_a, size_t nsteps, const int64_t* gals, const double* const* atk, size_t atk_dnum,
                                 size_t atk_size, int64_t* ggsw, size_t res_dnum, size_t res_size, size_t gap,
                                 size_t log_gap_in, size_t log_gap_out, size_t log_domain,
                                 const double* const* tsk, size_t tsk_dnum, size_t tsk_size) {
    const size_t k_brk = bases[0], k_atk = bases[1], k_tsk = bases[2], k_res = bases[3];
    size_t n = t->m << 1, cols = rank + 1, asz = atk_glwe_size, tsz = trace_size;
    size_t ct_a = n * cols * asz, ct_t = n * cols * tsz, ct_res = n * cols * res_size;
    size_t log_n = 0;
    while (((size_t)1 << log_n) < n) ++log_n;
    int64_t* acc_brk = (int64_t*)calloc(n * cols * glwe_size, sizeof(int64_t));
    int64_t* acc = (int64_t*)calloc(ct_a, sizeof(int64_t));
    int64_t* rot = (int64_t*)calloc(ct_a, sizeof(int64_t));
    int64_t* tmp = (int64_t*)calloc(ct_t, sizeof(int64_t));
    pzr_blind_rotation_execute(t, rank, n_lwe, block_size, acc_brk, glwe_size, k_brk, lwe_2n, lut, lut_size, brk, brk_dnum, brk_size, x_pow_a);
    if (k_atk == k_brk) { /* glwe_copy */
        memcpy(acc, acc_brk, n * cols * (asz < glwe_size ? asz : glwe_size) * sizeof(int64_t));
    } else {
        for (size_t c = 0; c < cols; ++c) pzr_vec_znx_normalize(n, acc, cols, asz, k_atk, 0, c, acc_brk, cols, glwe_size, k_brk, c);
    }
    size_t steps = (size_t)1 << log_domain;
    int64_t* a_trace = NULL; int64_t* packed = NULL; int64_t* cts = NULL; int64_t** slots = NULL;
    const int repack = to_exponent && log_gap_in != log_gap_out;
    if (repack) {
        a_trace = (int64_t*)calloc(ct_a, sizeof(int64_t));
        packed = (int64_t*)calloc(ct_a, sizeof(int64_t));
        cts = (int64_t*)calloc(steps * ct_a, sizeof(int64_t));
        slots = (int64_t**)calloc(n, sizeof(int64_t*));
    }
    size_t skip = to_exponent ? log_n - log_gap_in + 1 : 0;
    size_t tsteps = to_exponent ? log_n - skip : nsteps;
    for (size_t i = 0; i < res_dnum; ++i) {
        int64_t* row = ggsw + (i * cols) * ct_res;
        if (!repack) {
            /* glwe_trace.rs:107-119: tmp = zero-extended copy of a, trace over the steps skip.. */
            memset(tmp, 0, ct_t * sizeof(int64_t));
            memcpy(tmp, acc, (ct_a < ct_t ? ct_a : ct_t) * sizeof(int64_t));
            pzr_glwe_trace_assign(t, rank, tmp, tsz, k_atk, tsteps, gals + skip, atk + skip, atk_dnum, atk_size, 1);
            cbt_finish_row(n, cols, row, res_size, k_res, tmp, tsz, k_atk);
        } else {
            /* post_process :392-417: partial trace in a's layout, 2^log_domain shifted copies, glwe_pack whose closing glwe_trace
             * (glwe_packing.rs:166) runs on a temporary of trace_size limbs — here trace_size == atk_glwe_size (brk.max_k >= res.max_k) */
            memcpy(a_trace, acc, ct_a * sizeof(int64_t));
            pzr_glwe_trace_assign(t, rank, a_trace, asz, k_atk, tsteps, gals + skip, atk + skip, atk_dnum, atk_size, 1);
            memset(slots, 0, n * sizeof(int64_t*));
            for (size_t sidx = 0; sidx < steps; ++sidx) {
                if (sidx != 0) { /* :405-407 glwe_rotate_assign(-(1 << log_gap_in), a_trace) */
                    for (size_t c = 0; c < cols; ++c)
                        pzr_vec_znx_rotate(n, -((int64_t)1 << log_gap_in), rot, cols, asz, c, a_trace, cols, asz, c);
                    memcpy(a_trace, rot, ct_a * sizeof(int64_t));
                }
                memcpy(cts + sidx * ct_a, a_trace, ct_a * sizeof(int64_t));
                slots[sidx << log_gap_out] = cts + sidx * ct_a;
            }
            pzr_glwe_pack(t, rank, packed, slots, asz, k_atk, log_gap_out, gals, atk, atk_dnum, atk_size);
            cbt_finish_row(n, cols, row, res_size, k_res, packed, asz, k_atk);
        }
        if (i + 1 < res_dnum) { /* circuit.rs:363-365 glwe_rotate_assign(-gap) */
            for (size_t c = 0; c < cols; ++c) pzr_vec_znx_rotate(n, -(int64_t)gap, rot, cols, asz, c, acc, cols, asz, c);
            memcpy(acc, rot, ct_a * sizeof(int64_t));
        }
    }
    pzr_ggsw_expand_row(t, rank, ggsw, res_dnum, res_size, k_res, tsk, tsk_dnum, tsk_size, 1, k_tsk);
    free(acc_brk); free(acc); free(rot); free(tmp); free(a_trace); free(packed); free(cts); free(slots);
}

/* ------------------------------------------------------------------------ */
/* reference/fft64/convolution.rs (HalImpl cnv_*, hal_impl.rs:670-754)        */
/* CnvPVecL / CnvPVecR (FFT64): [col][blk < m/4][limb < size][re x4 | im x4]   */
/* ------------------------------------------------------------------------ */

/* reim/conversion.rs:31-40 */
static void reim_from_znx_i64_masked(double* res, const int64_t* a, int64_t mask, size_t len) {
    for (size_t i = 0; i < len; ++i) res[i] = (double)(a[i] & mask);
}

/* hal_defaults/convolution.rs:40-45, :63-68 */
size_t pzr_cnv_prepare_tmp_bytes(size_t n, size_t res_size, size_t a_size) { return n * zmin(res_size, a_size) * sizeof(double); }

/* convolution.rs:35-80 (convolution_prepare; _left and _right are the same function for FFT64) */
void pzr_cnv_prepare(const pzr_tables* t, double* res, size_t res_cols, size_t res_size,
                     const int64_t* a, size_t a_cols, size_t a_size, int64_t mask) {
    size_t m = t->m, n = m << 1;
    size_t min_size = zmin(res_size, a_size);
    double* tmp = (double*)calloc(n * (min_size ? min_size : 1), sizeof(double)); /* VecZnxDft(1, min(res.size, a.size)) */
    for (size_t i = 0; i < res_cols; ++i) {
        pzr_vec_znx_dft_apply(t, 1, 0, tmp, 1, min_size, 0, a, a_cols, a_size, i);
        if (min_size > 0) { /* :56-61: the last active limb again, masked */
            size_t last = min_size - 1;
            reim_from_znx_i64_masked(tmp + n * last, at_ci64(a, n, a_cols, i, last), mask, n);
            pzr_fft(t, tmp + n * last);
        }
        double* res_col = res + i * n * res_size;
        for (size_t blk = 0; blk < m / 4; ++blk) {
            reim4_extract_1blk(m, min_size, blk, res_col + blk * res_size * 8, tmp);
            reim_zero(res_col + blk * res_size * 8 + min_size * 8, (res_size - min_size) * 8);
        }
    }
    free(tmp);
}

/* convolution.rs:82-140 (convolution_prepare_self): left prepared as above, right = a copy of it */
void pzr_cnv_prepare_self(const pzr_tables* t, double* left, double* right, size_t cols, size_t size,
                          const int64_t* a, size_t a_cols, size_t a_size, int64_t mask) {
    size_t n = t->m << 1;
    pzr_cnv_prepare(t, left, cols, size, a, a_cols, a_size, mask);
    memcpy(right, left, n * cols * size * sizeof(double));
}

/* reim4/arithmetic_ref.rs:235-247 */
static void reim4_convolution_1coeff(size_t k, double* dst, const double* a, size_t a_size, const double* b, size_t b_size) {
    reim_zero(dst, 8);
    if (k >= a_size + b_size) return;
    size_t j_min = k >= a_size - 1 ? k - (a_size - 1) : 0;
    size_t j_max = zmin(k + 1, b_size);
    for (size_t j = j_min; j < j_max; ++j) reim4_add_mul(dst, a + 8 * (k - j), b + 8 * j);
}

/* reim4/mod.rs:46-58 */
static void reim4_convolution(double* dst, size_t dst_size, size_t offset, const double* a, size_t a_size, const double* b, size_t b_size) {
    for (size_t k = 0; k < dst_size; ++k) reim4_convolution_1coeff(k + offset, dst + 8 * k, a, a_size, b, b_size);
}

/* reim4/arithmetic_ref.rs:37-50: 2*rows rows of 4 values, row r at dst + r*m + 4*blk (the [re | im] polynomials of a one-column
 * VecZnxDft are exactly such rows) */
static void reim4_save_1blk_contiguous(size_t m, size_t rows, size_t blk, double* dst, const double* src) {
    size_t off = blk << 2;
    for (size_t r = 0; r < 2 * rows; ++r) memcpy(dst + r * m + off, src + 4 * r, 4 * sizeof(double));
}

size_t pzr_cnv_apply_dft_tmp_bytes(size_t res_size, size_t a_size, size_t b_size) { /* convolution.rs:205-208 */
    return sizeof(double) * 8 * zmin(res_size, a_size + b_size - 1);
}
size_t pzr_cnv_pairwise_apply_dft_tmp_bytes(size_t res_size, size_t a_size, size_t b_size) { /* :261-263 */
    return pzr_cnv_apply_dft_tmp_bytes(res_size, a_size, b_size) + (a_size + b_size) * sizeof(double) * 8;
}
size_t pzr_cnv_by_const_apply_tmp_bytes(size_t res_size, size_t a_size, size_t b_size) { /* :142-145 */
    return sizeof(int64_t) * (zmin(res_size, a_size + b_size - 1) + a_size) * 8;
}

/* convolution.rs:210-259 (convolution_apply_dft).  NOTE: the reference saves the blocks at the RAW start of `res`
 * (:232, :251: res.raw_mut() with rows spaced by m), i.e. it ignores res_col and assumes a one-column res; only the zero fill
 * (:256-258) uses res_col.  Restated literally; every caller passes a one-column res_dft with res_col = 0. */
void pzr_cnv_apply_dft(size_t n, size_t cnv_offset, double* res, size_t res_cols, size_t res_size, size_t res_col,
                       const double* a, size_t a_size, size_t a_col, const double* b, size_t b_size, size_t b_col) {
    size_t m = n >> 1;
    size_t bound = a_size + b_size - 1;
    size_t min_size = zmin(res_size, bound);
    size_t offset = zmin(cnv_offset, bound);
    double* tmp = (double*)calloc(8 * (min_size ? min_size : 1), sizeof(double));
    const double* ap = a + a_col * n * a_size;
    const double* bp = b + b_col * n * b_size;
    for (size_t blk = 0; blk < m / 4; ++blk) {
        reim4_convolution(tmp, min_size, offset, ap, a_size, bp, b_size);
        reim4_save_1blk_contiguous(m, min_size, blk, res, tmp);
        ap += a_size * 8;
        bp += b_size * 8;
    }
    for (size_t j = min_size; j < res_size; ++j) reim_zero(at_f64(res, n, res_cols, res_col, j), n);
    free(tmp);
}

/* convolution.rs:265-345 (convolution_pairwise_apply_dft): (a_i + a_j) * (b_i + b_j) */
void pzr_cnv_pairwise_apply_dft(size_t n, size_t cnv_offset, double* res, size_t res_cols, size_t res_size, size_t res_col,
                                const double* a, size_t a_size, const double* b, size_t b_size, size_t col_i, size_t col_j) {
    if (col_i == col_j) {
        pzr_cnv_apply_dft(n, cnv_offset, res, res_cols, res_size, res_col, a, a_size, col_i, b, b_size, col_j);
        return;
    }
    size_t m = n >> 1;
    size_t bound = a_size + b_size - 1;
    size_t min_size = zmin(res_size, bound);
    size_t offset = zmin(cnv_offset, bound);
    double* tmp_a = (double*)calloc(8 * a_size, sizeof(double));
    double* tmp_b = (double*)calloc(8 * b_size, sizeof(double));
    double* tmp_res = (double*)calloc(8 * (min_size ? min_size : 1), sizeof(double));
    const double *a0 = a + col_i * n * a_size, *a1 = a + col_j * n * a_size;
    const double *b0 = b + col_i * n * b_size, *b1 = b + col_j * n * b_size;
    for (size_t blk = 0; blk < m / 4; ++blk) {
        for (size_t x = 0; x < 8 * a_size; ++x) tmp_a[x] = a0[x] + a1[x]; /* reim_add */
        for (size_t x = 0; x < 8 * b_size; ++x) tmp_b[x] = b0[x] + b1[x];
        reim4_convolution(tmp_res, min_size, offset, tmp_a, a_size, tmp_b, b_size);
        reim4_save_1blk_contiguous(m, min_size, blk, res, tmp_res);
        a0 += 8 * a_size; a1 += 8 * a_size; b0 += 8 * b_size; b1 += 8 * b_size;
    }
    for (size_t j = min_size; j < res_size; ++j) reim_zero(at_f64(res, n, res_cols, res_col, j), n);
    free(tmp_a); free(tmp_b); free(tmp_res);
}

/* convolution.rs:147-203 (convolution_by_const_apply) with :395-421 (i64_convolution_by_const_1coeff_ref): res limb k =
 * sum_j a[k + offset - j] * b[j], wrapping i64, coefficient-wise */
void pzr_cnv_by_const_apply(size_t n, size_t cnv_offset, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                            const int64_t* a, size_t a_cols, size_t a_size, size_t a_col, const int64_t* b, size_t b_size) {
    size_t bound = a_size + b_size - 1;
    size_t min_size = zmin(res_size, bound);
    size_t offset = zmin(cnv_offset, bound);
    for (size_t kk = 0; kk < min_size; ++kk) {
        size_t k = kk + offset;
        int64_t* dst = at_i64(res, n, res_cols, res_col, kk);
        memset(dst, 0, n * sizeof(int64_t));
        if (k >= a_size + b_size) continue;
        size_t j_min = k >= a_size - 1 ? k - (a_size - 1) : 0;
        size_t j_max = zmin(k + 1, b_size);
        for (size_t j = j_min; j < j_max; ++j) {
            const int64_t* ai = at_ci64(a, n, a_cols, a_col, k - j);
            uint64_t bj = (uint64_t)b[j];
            for (size_t x = 0; x < n; ++x) dst[x] = (int64_t)((uint64_t)dst[x] + (uint64_t)ai[x] * bj);
        }
    }
    for (size_t j = min_size; j < res_size; ++j) memset(at_i64(res, n, res_cols, res_col, j), 0, n * sizeof(int64_t));
}

/* ------------------------------------------------------------------------ */
/* poulpy-core/src/operations/glwe.rs: GLWE tensoring (CKKS multiplication)   */
/* ------------------------------------------------------------------------ */

/* operations/glwe.rs:921-926 */
int64_t pzr_msb_mask_bottom_limb(size_t base2k, size_t k) {
    size_t r = k % base2k;
    return r == 0 ? ~(int64_t)0 : (int64_t)(~(uint64_t)0 << (base2k - r));
}
/* operations/glwe.rs:929-957 */
static size_t normalize_input_limb_bound_with_offset(size_t full_size, size_t res_size, size_t res_base2k, size_t in_base2k, int64_t res_offset) {
    int64_t offset_bits = res_offset % (int64_t)in_base2k;
    if (res_offset < 0 && offset_bits != 0) offset_bits += (int64_t)in_base2k;
    return zmin(full_size, div_ceil(res_size * res_base2k + (size_t)offset_bits, in_base2k));
}

/* One product term of the tensor (operations/glwe.rs:752-760 / :785-792): cnv -> idft (consume) -> normalize with
 * cnv_offset_lo into a one-column VecZnx `tmp` of res_size limbs */
static void tensor_term(const pzr_tables* t, size_t cnv_offset_hi, int64_t cnv_offset_lo, size_t dft_size, int64_t* tmp, size_t res_size,
                        size_t res_base2k, const double* a_prep, size_t a_size, const double* b_prep, size_t b_size, size_t ab_base2k,
                        size_t i, size_t j) {
    size_t n = t->m << 1;
    double* res_dft = (double*)calloc(n * (dft_size ? dft_size : 1), sizeof(double));
    if (i == j) pzr_cnv_apply_dft(n, cnv_offset_hi, res_dft, 1, dft_size, 0, a_prep, a_size, i, b_prep, b_size, i);
    else pzr_cnv_pairwise_apply_dft(n, cnv_offset_hi, res_dft, 1, dft_size, 0, a_prep, a_size, b_prep, b_size, i, j);
    pzr_vec_znx_idft_apply_consume(t, res_dft, 1, dft_size);
    pzr_vec_znx_normalize(n, tmp, 1, res_size, res_base2k, cnv_offset_lo, 0, (const int64_t*)res_dft, 1, dft_size, ab_base2k, 0);
    free(res_dft);
}

/* operations/glwe.rs:700-807 (glwe_tensor_apply, add_assign = 0) and :809-913 (glwe_tensor_apply_add_assign, add_assign = 1).
 * res: GLWETensor data = VecZnx(cols (cols + 1) / 2, res_size); column of the pair (i, j >= i) = i cols - i (i + 1) / 2 + j. */
void pzr_glwe_tensor_apply(const pzr_tables* t, size_t rank, size_t cnv_offset, int add_assign,
                           int64_t* res, size_t res_size, size_t res_base2k,
                           const int64_t* a, size_t a_size, size_t a_effective_k,
                           const int64_t* b, size_t b_size, size_t b_effective_k, size_t ab_base2k) {
    size_t n = t->m << 1;
    size_t cols = rank + 1, tcols = cols * (cols + 1) / 2;
    double* a_prep = (double*)calloc(n * cols * a_size, sizeof(double));
    double* b_prep = (double*)calloc(n * cols * b_size, sizeof(double));
    pzr_cnv_prepare(t, a_prep, cols, a_size, a, cols, a_size, pzr_msb_mask_bottom_limb(ab_base2k, a_effective_k));
    pzr_cnv_prepare(t, b_prep, cols, b_size, b, cols, b_size, pzr_msb_mask_bottom_limb(ab_base2k, b_effective_k));
    size_t hi;
    int64_t lo;
    if (cnv_offset < ab_base2k) { hi = 0; lo = -(int64_t)(ab_base2k - (cnv_offset % ab_base2k)); }
    else { size_t q = cnv_offset / ab_base2k; hi = q ? q - 1 : 0; lo = (int64_t)(cnv_offset % ab_base2k); }
    size_t dft_size = normalize_input_limb_bound_with_offset(a_size + b_size - hi, res_size, res_base2k, ab_base2k, lo);
    int64_t* tmp = (int64_t*)calloc(n * res_size, sizeof(int64_t));
    for (size_t i = 0; i < cols; ++i) {
        size_t col_i = i * cols - (i * (i + 1) / 2);
        tensor_term(t, hi, lo, dft_size, tmp, res_size, res_base2k, a_prep, a_size, b_prep, b_size, ab_base2k, i, i);
        if (add_assign) pzr_vec_znx_assign_op(0, n, res, tcols, res_size, col_i + i, tmp, 1, res_size, 0);
        else pzr_vec_znx_copy(n, res, tcols, res_size, col_i + i, tmp, 1, res_size, 0);
        for (size_t j = 0; j < cols; ++j) {
            if (j == i) continue;
            if (j < i) {
                size_t col_j = j * cols - (j * (j + 1) / 2);
                pzr_vec_znx_assign_op(1, n, res, tcols, res_size, col_j + i, tmp, 1, res_size, 0);
            } else if (add_assign) {
                pzr_vec_znx_assign_op(1, n, res, tcols, res_size, col_i + j, tmp, 1, res_size, 0);
            } else {
                pzr_vec_znx_negate(n, res, tcols, res_size, col_i + j, tmp, 1, res_size, 0);
            }
        }
    }
    for (size_t i = 0; i < cols; ++i) {
        size_t col_i = i * cols - (i * (i + 1) / 2);
        for (size_t j = i + 1; j < cols; ++j) {
            tensor_term(t, hi, lo, dft_size, tmp, res_size, res_base2k, a_prep, a_size, b_prep, b_size, ab_base2k, i, j);
            pzr_vec_znx_assign_op(0, n, res, tcols, res_size, col_i + j, tmp, 1, res_size, 0);
        }
    }
    free(tmp); free(a_prep); free(b_prep);
}

/* operations/glwe.rs:609-698 (glwe_tensor_square_apply) */
void pzr_glwe_tensor_square_apply(const pzr_tables* t, size_t rank, size_t cnv_offset,
                                  int64_t* res, size_t res_size, size_t res_base2k,
                                  const int64_t* a, size_t a_size, size_t a_effective_k, size_t a_base2k) {
    size_t n = t->m << 1;
    size_t cols = rank + 1, tcols = cols * (cols + 1) / 2;
    double* a_prep = (double*)calloc(n * cols * a_size, sizeof(double));
    double* b_prep = (double*)calloc(n * cols * a_size, sizeof(double));
    pzr_cnv_prepare_self(t, a_prep, b_prep, cols, a_size, a, cols, a_size, pzr_msb_mask_bottom_limb(a_base2k, a_effective_k));
    int64_t* diag = (int64_t*)calloc(n * cols * res_size, sizeof(int64_t));
    size_t hi;
    int64_t lo;
    if (cnv_offset < a_base2k) { hi = 0; lo = -(int64_t)(a_base2k - (cnv_offset % a_base2k)); }
    else { size_t q = cnv_offset / a_base2k; hi = q ? q - 1 : 0; lo = (int64_t)(cnv_offset % a_base2k); }
    size_t dft_size = normalize_input_limb_bound_with_offset(2 * a_size - hi, res_size, res_base2k, a_base2k, lo);
    int64_t* tmp = (int64_t*)calloc(n * res_size, sizeof(int64_t));
    for (size_t i = 0; i < cols; ++i) {
        size_t col_i = i * cols - (i * (i + 1) / 2);
        tensor_term(t, hi, lo, dft_size, tmp, res_size, res_base2k, a_prep, a_size, b_prep, a_size, a_base2k, i, i);
        pzr_vec_znx_copy(n, diag, cols, res_size, i, tmp, 1, res_size, 0);            /* normalized straight into diag_terms col i */
        pzr_vec_znx_copy(n, res, tcols, res_size, col_i + i, diag, cols, res_size, i);
    }
    for (size_t i = 0; i < cols; ++i) {
        size_t col_i = i * cols - (i * (i + 1) / 2);
        for (size_t j = i + 1; j < cols; ++j) {
            tensor_term(t, hi, lo, dft_size, tmp, res_size, res_base2k, a_prep, a_size, b_prep, a_size, a_base2k, i, j);
            pzr_vec_znx_copy(n, res, tcols, res_size, col_i + j, tmp, 1, res_size, 0); /* normalized straight into res col */
            pzr_vec_znx_assign_op(1, n, res, tcols, res_size, col_i + j, diag, cols, res_size, i);
            pzr_vec_znx_assign_op(1, n, res, tcols, res_size, col_i + j, diag, cols, res_size, j);
        }
    }
    free(tmp); free(diag); free(a_prep); free(b_prep);
}

/* operations/glwe.rs:541-607 (glwe_tensor_relinearize): a = GLWETensor data VecZnx(cols + pairs, a_size); the `pairs` = rank (rank + 1) / 2
 * columns behind the first cols are key-switched by tsk (prepared GGLWE pairs -> rank, tsk_size limbs) and the first cols are
 * added to the big value.  NOTE (:588-598): the un-normalized a is added whenever res_base2k == key_base2k, even if a_base2k
 * differs; restated literally. */
void pzr_glwe_tensor_relinearize(const pzr_tables* t, size_t rank,
                                 int64_t* res, size_t res_size, size_t res_base2k,
                                 const int64_t* a, size_t a_size, size_t a_base2k,
                                 const double* tsk_pmat, size_t dnum, size_t tsk_size, size_t dsize, size_t key_base2k) {
    size_t n = t->m << 1;
    size_t cols = rank + 1, pairs = rank * (rank + 1) / 2, acols = cols + pairs;
    size_t a_dft_size = div_ceil(a_size * a_base2k, key_base2k);
    double* a_dft = (double*)calloc(n * pairs * a_dft_size + 8, sizeof(double));
    int64_t* a_conv = (int64_t*)calloc(n * a_dft_size, sizeof(int64_t));
    if (a_base2k != key_base2k) {
        for (size_t i = 0; i < pairs; ++i) {
            pzr_vec_znx_normalize(n, a_conv, 1, a_dft_size, key_base2k, 0, 0, a, acols, a_size, a_base2k, cols + i);
            pzr_vec_znx_dft_apply(t, 1, 0, a_dft, pairs, a_dft_size, i, a_conv, 1, a_dft_size, 0);
        }
    } else {
        for (size_t i = 0; i < pairs; ++i) pzr_vec_znx_dft_apply(t, 1, 0, a_dft, pairs, a_dft_size, i, a, acols, a_size, cols + i);
    }
    double* res_dft = (double*)calloc(n * cols * tsk_size, sizeof(double));
    gglwe_product_dft(n, res_dft, cols, tsk_size, a_dft, pairs, a_dft_size, tsk_pmat, dnum, dsize);
    pzr_vec_znx_idft_apply_consume(t, res_dft, cols, tsk_size);
    int64_t* res_big = (int64_t*)res_dft;
    if (res_base2k == key_base2k) {
        for (size_t i = 0; i < cols; ++i) pzr_vec_znx_big_add_small_assign(n, res_big, cols, tsk_size, i, a, acols, a_size, i);
    } else {
        for (size_t i = 0; i < cols; ++i) {
            pzr_vec_znx_normalize(n, a_conv, 1, a_dft_size, key_base2k, 0, 0, a, acols, a_size, a_base2k, i);
            pzr_vec_znx_big_add_small_assign(n, res_big, cols, tsk_size, i, a_conv, 1, a_dft_size, 0);
        }
    }
    for (size_t i = 0; i < cols; ++i)
        pzr_vec_znx_normalize(n, res, cols, res_size, res_base2k, 0, i, res_big, cols, tsk_size, key_base2k, i);
    free(a_dft); free(a_conv); free(res_dft);
}

/* ------------------------------------------------------------------------ */
/* LWE glue of the gate bootstrap (BASELINE configs[3]: blind-rotate + key switch) */
/* ------------------------------------------------------------------------ */

/* poulpy-bin-fhe/src/blind_rotation/algorithms/mod.rs:136-171 (`mod_switch_2n`) and :173-176 (`div_round_by_pow2`).
 * lwe: VecZnx(n = n_lwe + 1, cols = 1, lwe_size); res: n_lwe + 1 values.  Restated literally, including that only limb 0 is
 * negated for `Left` and that the low limbs are appended without rounding in the multi-limb branch. */
void pzr_mod_switch_2n(size_t n2, int64_t* res, const int64_t* lwe, size_t n_lwe, size_t lwe_size, size_t base2k, int negate) {
    size_t len = n_lwe + 1;
    size_t log2n = 1;
    {   /* usize::BITS - (n - 1).leading_zeros() + 1 */
        size_t v = n2 - 1, bits = 0;
        while (v) { ++bits; v >>= 1; }
        log2n = bits + 1;
    }
    for (size_t i = 0; i < len; ++i) res[i] = negate ? (int64_t)(0 - (uint64_t)lwe[i]) : lwe[i];
    if (base2k > log2n) {
        size_t diff = base2k - (log2n - 1);
        for (size_t i = 0; i < len; ++i) res[i] = (int64_t)((uint64_t)res[i] + ((uint64_t)1 << (diff - 1))) >> diff;
    } else {
        size_t rem = base2k - (log2n % base2k);
        size_t size = div_ceil(log2n, base2k);
        (void)lwe_size;
        for (size_t i = 1; i < size; ++i) {
            const int64_t* x = lwe + i * len;
            if (i == size - 1 && rem != base2k) {
                size_t k_rem = base2k - rem;
                for (size_t j = 0; j < len; ++j) res[j] = (int64_t)(((uint64_t)res[j] << k_rem) + (uint64_t)(x[j] >> rem));
            } else {
                for (size_t j = 0; j < len; ++j) res[j] = (int64_t)(((uint64_t)res[j] << base2k) + (uint64_t)x[j]);
            }
        }
    }
}

/* poulpy-core/src/api/conversion.rs:15-40 (`lwe_sample_extract`): constant coefficient of column 0 and the first res_n_lwe
 * coefficients of column 1, limb by limb; limbs beyond min(res_size, a_size) are zero.  res: VecZnx(res_n_lwe + 1, 1, res_size). */
void pzr_lwe_sample_extract(size_t n, int64_t* res, size_t res_n_lwe, size_t res_size, const int64_t* a, size_t a_cols, size_t a_size) {
    size_t len = res_n_lwe + 1;
    size_t min_size = zmin(res_size, a_size);
    memset(res, 0, len * res_size * sizeof(int64_t));
    for (size_t i = 0; i < min_size; ++i) {
        int64_t* r = res + i * len;
        r[0] = a[n * (i * a_cols + 0)];
        memcpy(r + 1, a + n * (i * a_cols + 1), res_n_lwe * sizeof(int64_t));
    }
}

/* the LWE -> rank-1 GLWE embedding shared by keyswitching/lwe.rs:69-80 and conversion/lwe_to_glwe.rs:71-80: b to the constant
 * coefficient of column 0, a_0.. to the first n_lwe coefficients of column 1; glwe is zeroed first */
static void lwe_embed(size_t n, int64_t* glwe, size_t glwe_size, const int64_t* lwe, size_t n_lwe, size_t lwe_size) {
    size_t len = n_lwe + 1;
    memset(glwe, 0, n * 2 * glwe_size * sizeof(int64_t));
    for (size_t i = 0; i < lwe_size; ++i) {
        glwe[n * (i * 2 + 0)] = lwe[i * len];
        memcpy(glwe + n * (i * 2 + 1), lwe + i * len + 1, n_lwe * sizeof(int64_t));
    }
}

/* poulpy-core/src/keyswitching/lwe.rs:49-94 (`lwe_keyswitch_default`): embed (glwe_in has a's base and size), rank-1 -> rank-1
 * glwe_keyswitch into a GLWE with res's base and size, sample extract */
void pzr_lwe_keyswitch(const pzr_tables* t, int64_t* res, size_t res_n_lwe, size_t res_size, size_t res_base2k,
                       const int64_t* a, size_t a_n_lwe, size_t a_size, size_t a_base2k,
                       const double* key_pmat, size_t dnum, size_t key_size, size_t dsize, size_t key_base2k) {
    size_t n = t->m << 1;
    int64_t* glwe_in = (int64_t*)malloc(n * 2 * a_size * sizeof(int64_t));
    int64_t* glwe_out = (int64_t*)calloc(n * 2 * res_size, sizeof(int64_t));
    lwe_embed(n, glwe_in, a_size, a, a_n_lwe, a_size);
    pzr_glwe_keyswitch(t, 1, 1, glwe_out, res_size, res_base2k, glwe_in, a_size, a_base2k, key_pmat, dnum, key_size, dsize, key_base2k);
    pzr_lwe_sample_extract(n, res, res_n_lwe, res_size, glwe_out, 2, res_size);
    free(glwe_in); free(glwe_out);
}

/* poulpy-core/src/conversion/lwe_to_glwe.rs:46-121 (`glwe_from_lwe_default`): the rank-1 GLWE has the KEY's base and
 * glwe_size = ceil(lwe.max_k / key_base2k) limbs; same base: plain embedding (:75-80), else the two columns are embedded into a
 * one-column VecZnx and normalized across bases (:82-116); then glwe_keyswitch (rank 1 -> rank_out) */
void pzr_glwe_from_lwe(const pzr_tables* t, size_t rank_out, int64_t* res, size_t res_size, size_t res_base2k,
                       const int64_t* lwe, size_t n_lwe, size_t lwe_size, size_t lwe_base2k, size_t glwe_size,
                       const double* key_pmat, size_t dnum, size_t key_size, size_t dsize, size_t key_base2k) {
    size_t n = t->m << 1;
    size_t len = n_lwe + 1;
    int64_t* glwe = (int64_t*)calloc(n * 2 * glwe_size, sizeof(int64_t));
    if (lwe_base2k == key_base2k) {
        for (size_t i = 0; i < lwe_size; ++i) {
            glwe[n * (i * 2 + 0)] = lwe[i * len];
            memcpy(glwe + n * (i * 2 + 1), lwe + i * len + 1, n_lwe * sizeof(int64_t));
        }
    } else {
        int64_t* a_conv = (int64_t*)calloc(n * lwe_size, sizeof(int64_t));
        for (size_t j = 0; j < lwe_size; ++j) a_conv[n * j] = lwe[j * len];
        pzr_vec_znx_normalize(n, glwe, 2, glwe_size, key_base2k, 0, 0, a_conv, 1, lwe_size, lwe_base2k, 0);
        memset(a_conv, 0, n * lwe_size * sizeof(int64_t));
        for (size_t j = 0; j < lwe_size; ++j) memcpy(a_conv + n * j, lwe + j * len + 1, n_lwe * sizeof(int64_t));
        pzr_vec_znx_normalize(n, glwe, 2, glwe_size, key_base2k, 0, 1, a_conv, 1, lwe_size, lwe_base2k, 0);
        free(a_conv);
    }
    pzr_glwe_keyswitch(t, 1, rank_out, res, res_size, res_base2k, glwe, glwe_size, key_base2k, key_pmat, dnum, key_size, dsize, key_base2k);
    free(glwe);
}

/* poulpy-core/src/conversion/glwe_to_lwe.rs:42-90 (`lwe_from_glwe_default`): rotate by X^-a_idx when a_idx != 0 (glwe_rotate =
 * vec_znx_rotate on every column, all limbs), key switch rank_in -> 1 into a GLWE with res's base and size, sample extract */
void pzr_lwe_from_glwe(const pzr_tables* t, size_t rank_in, int64_t* res, size_t res_n_lwe, size_t res_size, size_t res_base2k,
                       const int64_t* a, size_t a_size, size_t a_base2k, size_t a_idx,
                       const double* key_pmat, size_t dnum, size_t key_size, size_t dsize, size_t key_base2k) {
    size_t n = t->m << 1;
    size_t cols = rank_in + 1;
    int64_t* tmp_in = (int64_t*)malloc(n * cols * a_size * sizeof(int64_t));
    int64_t* tmp1 = (int64_t*)calloc(n * 2 * res_size, sizeof(int64_t));
    if (a_idx == 0) memcpy(tmp_in, a, n * cols * a_size * sizeof(int64_t));
    else
        for (size_t c = 0; c < cols; ++c) pzr_vec_znx_rotate(n, -(int64_t)a_idx, tmp_in, cols, a_size, c, a, cols, a_size, c);
    pzr_glwe_keyswitch(t, rank_in, 1, tmp1, res_size, res_base2k, tmp_in, a_size, a_base2k, key_pmat, dnum, key_size, dsize, key_base2k);
    pzr_lwe_sample_extract(n, res, res_n_lwe, res_size, tmp1, 2, res_size);
    free(tmp_in); free(tmp1);
}
