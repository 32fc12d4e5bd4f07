/* tfhe_oracle_fast.c -- TUNED CPU port of the same KS-PBS algorithm, used ONLY by bench.py's cpu_baseline leg.
 *
 * TEST INFRASTRUCTURE / BASELINE ONLY: the product never links this.  It computes exactly what tfhe_oracle.c computes
 * (same "FFT schedule v1", same operation order per element, -ffp-contract=off with explicit fma(): the outputs are
 * BIT-IDENTICAL, tests/test_oracle_golden.py::test_fast_port_matches_the_oracle), but is organised for the host CPU the
 * way a production CPU library is, so that the GPU is not compared with a strawman:
 *   - no allocation in the hot loop (per-thread scratch), structure-of-arrays polynomials;
 *   - per-stage twiddle tables contiguous in the butterfly index, so the radix-4 loops of stages with >= 4 butterflies
 *     per block auto-vectorise (AVX2 / AVX-512 under -march=native); the last stage runs on 4 x 4 blocks;
 *   - rotation + subtraction + decomposition + fold + twist fused into one pass, inverse twist + rounding + accumulate
 *     into another; the keyswitch subtracts digit * row with the five levels of a coefficient in one pass.
 * Reference functions restated: see tfhe_oracle.c (every function there cites its Rust source); the path is
 * shortint/server_key/mod.rs:783-857 -> lwe_keyswitch.rs:96-170 + fft64/crypto/bootstrap.rs:242-364 + ggsw.rs:477-598.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "tfhe_oracle.h"

#if defined(__AVX2__)
#include <immintrin.h>
/* Last stage (L = 4: the four inputs of a butterfly are CONSECUTIVE elements): four butterflies at a time on 4 x 4
 * blocks -- transpose, the butterfly of tfhe_oracle.c vertically (same additions, so the same bits), transpose back. */
static inline void transpose4(__m256d *r0, __m256d *r1, __m256d *r2, __m256d *r3)
{
    const __m256d t0 = _mm256_unpacklo_pd(*r0, *r1), t1 = _mm256_unpackhi_pd(*r0, *r1);
    const __m256d t2 = _mm256_unpacklo_pd(*r2, *r3), t3 = _mm256_unpackhi_pd(*r2, *r3);
    *r0 = _mm256_permute2f128_pd(t0, t2, 0x20);
    *r1 = _mm256_permute2f128_pd(t1, t3, 0x20);
    *r2 = _mm256_permute2f128_pd(t0, t2, 0x31);
    *r3 = _mm256_permute2f128_pd(t1, t3, 0x31);
}
static inline void last_stage16(double *re, double *im, int inverse)
{
    __m256d ar = _mm256_loadu_pd(re), br = _mm256_loadu_pd(re + 4), cr = _mm256_loadu_pd(re + 8), dr = _mm256_loadu_pd(re + 12);
    __m256d ai = _mm256_loadu_pd(im), bi = _mm256_loadu_pd(im + 4), ci = _mm256_loadu_pd(im + 8), di = _mm256_loadu_pd(im + 12);
    transpose4(&ar, &br, &cr, &dr);
    transpose4(&ai, &bi, &ci, &di);
    const __m256d t0r = _mm256_add_pd(ar, cr), t0i = _mm256_add_pd(ai, ci), t1r = _mm256_sub_pd(ar, cr), t1i = _mm256_sub_pd(ai, ci);
    const __m256d t2r = _mm256_add_pd(br, dr), t2i = _mm256_add_pd(bi, di), t3r = _mm256_sub_pd(br, dr), t3i = _mm256_sub_pd(bi, di);
    __m256d y0r = _mm256_add_pd(t0r, t2r), y0i = _mm256_add_pd(t0i, t2i), y2r = _mm256_sub_pd(t0r, t2r), y2i = _mm256_sub_pd(t0i, t2i);
    __m256d y1r, y1i, y3r, y3i;
    if (!inverse) { /* y1 = t1 - i t3, y3 = t1 + i t3 */
        y1r = _mm256_add_pd(t1r, t3i); y1i = _mm256_sub_pd(t1i, t3r);
        y3r = _mm256_sub_pd(t1r, t3i); y3i = _mm256_add_pd(t1i, t3r);
    } else {        /* x1 = t1 + i t3, x3 = t1 - i t3 */
        y1r = _mm256_sub_pd(t1r, t3i); y1i = _mm256_add_pd(t1i, t3r);
        y3r = _mm256_add_pd(t1r, t3i); y3i = _mm256_sub_pd(t1i, t3r);
    }
    transpose4(&y0r, &y1r, &y2r, &y3r);
    transpose4(&y0i, &y1i, &y2i, &y3i);
    _mm256_storeu_pd(re, y0r); _mm256_storeu_pd(re + 4, y1r); _mm256_storeu_pd(re + 8, y2r); _mm256_storeu_pd(re + 12, y3r);
    _mm256_storeu_pd(im, y0i); _mm256_storeu_pd(im + 4, y1i); _mm256_storeu_pd(im + 8, y2i); _mm256_storeu_pd(im + 12, y3i);
}
#endif

typedef struct {
    uint32_t N, n, stages;
    const double *tw_re, *tw_im; /* twist (borrowed from the orc_fft) */
    /* stage s (L = n >> 2s, Q = L / 4): twiddles of outputs 1..3 for butterfly j < Q: w[s][m-1][re/im][j] */
    double *w[8][3][2];
} fast_fft;

typedef struct {
    uint64_t *ct1;           /* [2][N] */
    double *fr[2], *fi[2];   /* Fourier images of the two digit polynomials */
    double *orr[2], *oi[2];  /* the two output columns */
} fast_scratch;

static fast_fft *fast_fft_new(const orc_fft *f, uint32_t N)
{
    fast_fft *g = (fast_fft *)calloc(1, sizeof(fast_fft));
    g->N = N;
    g->n = N / 2;
    g->tw_re = orc_fft_twist_re(f);
    g->tw_im = orc_fft_twist_im(f);
    const double *wr = orc_fft_w_re(f), *wi = orc_fft_w_im(f);
    uint32_t st = 0;
    while ((1u << (2 * st)) < g->n) st++;
    g->stages = st;
    for (uint32_t s = 0; s < st; s++) {
        const uint32_t L = g->n >> (2 * s), Q = L / 4, step = g->n / L;
        for (int m = 0; m < 3; m++) {
            g->w[s][m][0] = (double *)aligned_alloc(64, sizeof(double) * (Q < 8 ? 8 : Q));
            g->w[s][m][1] = (double *)aligned_alloc(64, sizeof(double) * (Q < 8 ? 8 : Q));
            for (uint32_t j = 0; j < Q; j++) {
                g->w[s][m][0][j] = wr[(size_t)j * step * (m + 1)];
                g->w[s][m][1][j] = wi[(size_t)j * step * (m + 1)];
            }
        }
    }
    return g;
}
static void fast_fft_free(fast_fft *g)
{
    if (!g) return;
    for (uint32_t s = 0; s < g->stages; s++)
        for (int m = 0; m < 3; m++) { free(g->w[s][m][0]); free(g->w[s][m][1]); }
    free(g);
}

/* one radix-4 DIF butterfly position (see fft_fwd in tfhe_oracle.c: identical expressions).  The oracle skips the
 * multiplication for j == 0; multiplying by (1, 0) differs only in the sign of zeros, which never reaches a u64. */
#define FWD_BFLY(j, W1R, W1I, W2R, W2I, W3R, W3I)                                                                   \
    do {                                                                                                            \
        const double ar = r0[j], ai = i0[j], br = r1[j], bi = i1[j];                                                \
        const double cr = r2[j], ci = i2[j], dr = r3[j], di = i3[j];                                                \
        const double t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;                                    \
        const double t2r = br + dr, t2i = bi + di, t3r = br - dr, t3i = bi - di;                                    \
        const double y1r = t1r + t3i, y1i = t1i - t3r, y2r = t0r - t2r, y2i = t0i - t2i;                            \
        const double y3r = t1r - t3i, y3i = t1i + t3r;                                                              \
        r0[j] = t0r + t2r; i0[j] = t0i + t2i;                                                                       \
        r1[j] = fma(y1r, W1R, -(y1i * W1I)); i1[j] = fma(y1r, W1I, y1i * W1R);                                      \
        r2[j] = fma(y2r, W2R, -(y2i * W2I)); i2[j] = fma(y2r, W2I, y2i * W2R);                                      \
        r3[j] = fma(y3r, W3R, -(y3i * W3I)); i3[j] = fma(y3r, W3I, y3i * W3R);                                      \
    } while (0)
#define QUARTERS(re, im, b, Q)                                                                                      \
    double *restrict r0 = (re) + (b), *restrict r1 = r0 + (Q), *restrict r2 = r1 + (Q), *restrict r3 = r2 + (Q);    \
    double *restrict i0 = (im) + (b), *restrict i1 = i0 + (Q), *restrict i2 = i1 + (Q), *restrict i3 = i2 + (Q)

static void fast_fwd(const fast_fft *g, double *restrict re, double *restrict im)
{
    const uint32_t n = g->n;
    for (uint32_t s = 0; s + 1 < g->stages; s++) { /* stages with Q >= 4: unit-stride loops over j */
        const uint32_t L = n >> (2 * s), Q = L / 4;
        const double *restrict w1r = g->w[s][0][0], *restrict w1i = g->w[s][0][1];
        const double *restrict w2r = g->w[s][1][0], *restrict w2i = g->w[s][1][1];
        const double *restrict w3r = g->w[s][2][0], *restrict w3i = g->w[s][2][1];
        for (uint32_t b = 0; b < n; b += L) {
            QUARTERS(re, im, b, Q);
#pragma omp simd
            for (uint32_t j = 0; j < Q; j++) FWD_BFLY(j, w1r[j], w1i[j], w2r[j], w2i[j], w3r[j], w3i[j]);
        }
    }
    /* last stage (L = 4, Q = 1): no twiddles */
#if defined(__AVX2__)
    if (n % 16 == 0) {
        for (uint32_t b = 0; b < n; b += 16) last_stage16(re + b, im + b, 0);
        return;
    }
#endif
    for (uint32_t b = 0; b < n; b += 4) {
        const double ar = re[b], ai = im[b], br = re[b + 1], bi = im[b + 1];
        const double cr = re[b + 2], ci = im[b + 2], dr = re[b + 3], di = im[b + 3];
        const double t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;
        const double t2r = br + dr, t2i = bi + di, t3r = br - dr, t3i = bi - di;
        re[b] = t0r + t2r; im[b] = t0i + t2i;
        re[b + 2] = t0r - t2r; im[b + 2] = t0i - t2i;
        re[b + 1] = t1r + t3i; im[b + 1] = t1i - t3r;
        re[b + 3] = t1r - t3i; im[b + 3] = t1i + t3r;
    }
}

#define INV_BFLY(j, W1R, W1I, W2R, W2I, W3R, W3I)                                                                   \
    do {                                                                                                            \
        const double ar = r0[j], ai = i0[j];                                                                        \
        const double br = fma(r1[j], W1R, i1[j] * W1I), bi = fma(-r1[j], W1I, i1[j] * W1R);                         \
        const double cr = fma(r2[j], W2R, i2[j] * W2I), ci = fma(-r2[j], W2I, i2[j] * W2R);                         \
        const double dr = fma(r3[j], W3R, i3[j] * W3I), di = fma(-r3[j], W3I, i3[j] * W3R);                         \
        const double t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;                                    \
        const double t2r = br + dr, t2i = bi + di, t3r = br - dr, t3i = bi - di;                                    \
        r0[j] = t0r + t2r; i0[j] = t0i + t2i;                                                                       \
        r2[j] = t0r - t2r; i2[j] = t0i - t2i;                                                                       \
        r1[j] = t1r - t3i; i1[j] = t1i + t3r;                                                                       \
        r3[j] = t1r + t3i; i3[j] = t1i - t3r;                                                                       \
    } while (0)

static void fast_inv(const fast_fft *g, double *restrict re, double *restrict im)
{
    const uint32_t n = g->n;
#if defined(__AVX2__)
    const uint32_t scalar_from = (n % 16 == 0) ? n : 0;
    for (uint32_t b = 0; b < scalar_from; b += 16) last_stage16(re + b, im + b, 1);
#else
    const uint32_t scalar_from = 0;
#endif
    for (uint32_t b = scalar_from; b < n; b += 4) { /* stage with Q = 1 first */
        const double ar = re[b], ai = im[b], br = re[b + 1], bi = im[b + 1];
        const double cr = re[b + 2], ci = im[b + 2], dr = re[b + 3], di = im[b + 3];
        const double t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;
        const double t2r = br + dr, t2i = bi + di, t3r = br - dr, t3i = bi - di;
        re[b] = t0r + t2r; im[b] = t0i + t2i;
        re[b + 2] = t0r - t2r; im[b + 2] = t0i - t2i;
        re[b + 1] = t1r - t3i; im[b + 1] = t1i + t3r;
        re[b + 3] = t1r + t3i; im[b + 3] = t1i - t3r;
    }
    for (int32_t s = (int32_t)g->stages - 2; s >= 0; s--) {
        const uint32_t L = n >> (2 * s), Q = L / 4;
        const double *restrict w1r = g->w[s][0][0], *restrict w1i = g->w[s][0][1];
        const double *restrict w2r = g->w[s][1][0], *restrict w2i = g->w[s][1][1];
        const double *restrict w3r = g->w[s][2][0], *restrict w3i = g->w[s][2][1];
        for (uint32_t b = 0; b < n; b += L) {
            QUARTERS(re, im, b, Q);
#pragma omp simd
            for (uint32_t j = 0; j < Q; j++) INV_BFLY(j, w1r[j], w1i[j], w2r[j], w2i[j], w3r[j], w3i[j]);
        }
    }
}

/* ct1 = acc * X^a_hat - acc, one signed digit of base 2^bl (level count 1), fold, forward twist -- one pass.
 * src(c) = (c - a_hat) mod 2N with the negacyclic sign (orc_monomial_mul_and_subtract), digit as in
 * orc_add_external_product (closest_representable + decompose_one_level), twist as in orc_fft_forward_integer. */
static inline int64_t one_digit(uint64_t d, uint32_t bl)
{
    const uint32_t shift = 64 - bl - 1;
    uint64_t st = ((d >> shift) + 1) >> 1;        /* closest_representable(d) >> (64 - bl) */
    const uint64_t mask = ((uint64_t)1 << bl) - 1;
    uint64_t res = st & mask;
    st >>= bl;
    uint64_t carry = ((res - 1) | st) & res;
    carry >>= bl - 1;
    return (int64_t)(res - (carry << bl));
}

static void rotate_decompose_twist(const fast_fft *g, const uint64_t *restrict acc, uint32_t a_hat, uint32_t bl,
                                   double *restrict out_re, double *restrict out_im, int64_t *restrict dig)
{
    const size_t N = g->N, n = g->n;
    const size_t rem = a_hat % N;
    const uint64_t flip = ((a_hat / N) & 1) ? ~(uint64_t)0 : 0; /* all-ones: the rotated block is negated the other way round */
    /* digit polynomial first (two contiguous runs: no modulo in the loops).  (v ^ m) - m = -v for m = all-ones */
    const uint64_t *restrict hi = acc + (N - rem);
    {
        const uint64_t m = ~flip; /* i < rem: -v when cycles even */
#pragma omp simd
        for (size_t i = 0; i < rem; i++) dig[i] = one_digit(((hi[i] ^ m) - m) - acc[i], bl);
    }
    {
        const uint64_t m = flip;
        int64_t *restrict d2 = dig + rem;
        const uint64_t *restrict a2 = acc + rem;
#pragma omp simd
        for (size_t i = 0; i < N - rem; i++) d2[i] = one_digit(((acc[i] ^ m) - m) - a2[i], bl);
    }
    const double *restrict twr = g->tw_re, *restrict twi = g->tw_im;
    const int64_t *restrict dlo = dig, *restrict dhi = dig + n;
#pragma omp simd
    for (size_t j = 0; j < n; j++) {
        const double in_re = (double)dlo[j], in_im = (double)dhi[j];
        out_re[j] = fma(in_re, twr[j], -(in_im * twi[j]));
        out_im[j] = fma(in_re, twi[j], in_im * twr[j]);
    }
}

/* orc_torus_from_f64 (x - rne(x), * 2^64, rne, wrap to u64: orc_f64_to_u64_wrap), restated inline so that the loop
 * around it vectorises */
static inline uint64_t torus_inline(double x)
{
    double fr = x - rint(x);
    fr = rint(fr * 18446744073709551616.0);
    uint64_t bits;
    memcpy(&bits, &fr, 8);
    const uint64_t mant = (bits & 0xFFFFFFFFFFFFFull) | 0x10000000000000ull;
    const uint64_t rshift = (uint64_t)1086 - ((bits >> 52) & 0x7FF);
    const uint64_t v = rshift < 64 ? (mant << 11) >> rshift : 0;
    const uint64_t sm = (uint64_t)((int64_t)bits >> 63);
    return (v ^ sm) - sm;
}

static void ext_product_step(const orc_params *p, const fast_fft *g, fast_scratch *sc, uint64_t *acc, const double *gr,
                             const double *gi, uint32_t a_hat, int64_t *dig)
{
    const uint32_t N = p->poly_size, n = N / 2;
    for (int row = 0; row < 2; row++) {
        rotate_decompose_twist(g, acc + (size_t)row * N, a_hat, p->pbs_base_log, sc->fr[row], sc->fi[row], dig);
        fast_fwd(g, sc->fr[row], sc->fi[row]);
    }
    for (int col = 0; col < 2; col++) {
        const double *restrict a0r = gr + (size_t)(0 * 2 + col) * n, *restrict a0i = gi + (size_t)(0 * 2 + col) * n;
        const double *restrict a1r = gr + (size_t)(1 * 2 + col) * n, *restrict a1i = gi + (size_t)(1 * 2 + col) * n;
        const double *restrict f0r = sc->fr[0], *restrict f0i = sc->fi[0], *restrict f1r = sc->fr[1], *restrict f1i = sc->fi[1];
        double *restrict cr = sc->orr[col], *restrict ci = sc->oi[col];
#pragma omp simd
        for (uint32_t q = 0; q < n; q++) {
            const double r0 = fma(a0r[q], f0r[q], -(a0i[q] * f0i[q]));
            const double i0 = fma(a0r[q], f0i[q], a0i[q] * f0r[q]);
            cr[q] = fma(a1r[q], f1r[q], -fma(a1i[q], f1i[q], -r0));
            ci[q] = fma(a1r[q], f1i[q], fma(a1i[q], f1r[q], i0));
        }
        fast_inv(g, cr, ci);
        uint64_t *restrict plo = acc + (size_t)col * N, *restrict phi = plo + n;
        const double norm = 1.0 / (double)n;
        const double *restrict twr = g->tw_re, *restrict twi = g->tw_im;
#pragma omp simd
        for (uint32_t j = 0; j < n; j++) { /* orc_fft_add_backward_torus */
            const double wr = norm * twr[j], wi = norm * twi[j];
            const double mr = fma(cr[j], wr, ci[j] * wi);
            const double mi = fma(-cr[j], wi, ci[j] * wr);
            plo[j] += torus_inline(mr);
            phi[j] += torus_inline(mi);
        }
    }
}

/* keyswitch_lwe_ciphertext: out -= digit * row, the levels of one input coefficient in one pass over the output */
static void fast_keyswitch(const orc_params *p, const uint64_t *restrict ksk, const uint64_t *restrict in, uint64_t *restrict out)
{
    const uint32_t in_dim = p->glwe_dim * p->poly_size, out_size = p->lwe_dim + 1, lv = p->ks_level;
    memset(out, 0, sizeof(uint64_t) * out_size);
    out[p->lwe_dim] = in[in_dim];
    int64_t terms[64];
    for (uint32_t j = 0; j < in_dim; j++) {
        orc_decompose(in[j], p->ks_base_log, lv, terms);
        const uint64_t *block = ksk + (size_t)j * lv * out_size;
        for (uint32_t l = 0; l < lv; l++) {
            const uint64_t d = (uint64_t)terms[l];
            if (d == 0) continue; /* subtracting 0 * row changes nothing */
            const uint64_t *restrict row = block + (size_t)l * out_size;
#pragma omp simd
            for (uint32_t o = 0; o < out_size; o++) out[o] -= d * row[o];
        }
    }
}

void orcf_ks_pbs_batch(const orc_params *p, const orc_fft *f, const uint64_t *ksk, const double *bsk_re, const double *bsk_im,
                       const uint64_t *lwe_in, const uint64_t *luts, const uint32_t *lut_idx, uint64_t *lwe_out, size_t B,
                       int threads)
{
    const uint32_t N = p->poly_size, n = N / 2, gs = p->glwe_dim + 1;
    const size_t big = (size_t)p->glwe_dim * N + 1, lut_len = (size_t)gs * N, ggsw_len = (size_t)gs * gs * n;
    uint32_t lg_n = 0;
    while ((1u << lg_n) < n) lg_n++;
    if (gs != 2 || p->pbs_level != 1 || (lg_n & 1u)) { /* the tuned path is written for k = 1, one level, N/2 a power of 4 */
        orc_ks_pbs_batch(p, f, ksk, bsk_re, bsk_im, lwe_in, luts, lut_idx, lwe_out, B, threads);
        return;
    }
    fast_fft *g = fast_fft_new(f, N);
    uint32_t log2N = 0;
    while ((1u << log2N) < N) log2N++;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
    {
        fast_scratch sc;
        for (int i = 0; i < 2; i++) {
            sc.fr[i] = (double *)aligned_alloc(64, sizeof(double) * n); sc.fi[i] = (double *)aligned_alloc(64, sizeof(double) * n);
            sc.orr[i] = (double *)aligned_alloc(64, sizeof(double) * n); sc.oi[i] = (double *)aligned_alloc(64, sizeof(double) * n);
        }
        uint64_t *acc = (uint64_t *)aligned_alloc(64, sizeof(uint64_t) * gs * N);
        uint64_t *tmp = (uint64_t *)aligned_alloc(64, sizeof(uint64_t) * gs * N);
        uint64_t *small = (uint64_t *)aligned_alloc(64, sizeof(uint64_t) * ((p->lwe_dim + 8) & ~7u));
        int64_t *dig = (int64_t *)aligned_alloc(64, sizeof(int64_t) * N);
#pragma omp for schedule(dynamic, 1)
        for (size_t c = 0; c < B; c++) {
            fast_keyswitch(p, ksk, lwe_in + c * big, small);
            memcpy(tmp, luts + lut_idx[c] * lut_len, sizeof(uint64_t) * gs * N);
            const uint64_t b_hat = orc_modswitch(small[p->lwe_dim], log2N);
            for (uint32_t q = 0; q < gs; q++) orc_monomial_div(acc + (size_t)q * N, tmp + (size_t)q * N, b_hat, N);
            for (uint32_t i = 0; i < p->lwe_dim; i++) {
                if (small[i] == 0) continue;
                const uint32_t a_hat = (uint32_t)orc_modswitch(small[i], log2N);
                ext_product_step(p, g, &sc, acc, bsk_re + (size_t)i * ggsw_len, bsk_im + (size_t)i * ggsw_len, a_hat, dig);
            }
            orc_sample_extract(acc, p->glwe_dim, N, lwe_out + c * big);
        }
        for (int i = 0; i < 2; i++) { free(sc.fr[i]); free(sc.fi[i]); free(sc.orr[i]); free(sc.oi[i]); }
        free(acc); free(tmp); free(small); free(dig);
    }
    fast_fft_free(g);
}
