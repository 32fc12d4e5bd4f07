/*
 * tfhe_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See tfhe_oracle.h.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/tfhe/src).  Build: `make -C oracle` (gcc, -ffp-contract=off: every fused
 * multiply-add below is an explicit fma(); nothing else may be contracted).
 */
#include "tfhe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================================
 * T0: integer pieces
 * ====================================================================================== */

/* core_crypto/commons/math/decomposition/decomposer.rs:97-118 (closest_representable) */
uint64_t orc_closest_representable(uint64_t x, uint32_t base_log, uint32_t level)
{
    uint32_t non_rep_bit_count = 64 - level * base_log;
    uint32_t shift = non_rep_bit_count - 1;
    uint64_t res = x >> shift;
    res += 1;
    res &= ~(uint64_t)1; /* TWO.wrapping_neg() == ...1110 */
    return res << shift;
}

/* decomposition/iter.rs:120-127 == fft64/math/decomposition.rs:78-86 (decompose_one_level) */
static inline uint64_t decompose_one_level(uint32_t base_log, uint64_t *state, uint64_t mod_b_mask)
{
    uint64_t res = *state & mod_b_mask;
    *state >>= base_log;
    uint64_t carry = ((res - 1) | *state) & res;
    carry >>= base_log - 1;
    *state += carry;
    return res - (carry << base_log);
}

/* decomposer.rs:144-152 (decompose = closest_representable then iterate) + iter.rs:37-50,101-117.
 * The iterator yields level `level` first (lowest digit), level 1 last. */
void orc_decompose(uint64_t x, uint32_t base_log, uint32_t level, int64_t *terms)
{
    uint64_t state = orc_closest_representable(x, base_log, level) >> (64 - base_log * level);
    uint64_t mask = ((uint64_t)1 << base_log) - 1;
    for (uint32_t i = 0; i < level; i++)
        terms[i] = (int64_t)decompose_one_level(base_log, &state, mask);
}

/* core_crypto/fft_impl/common.rs:26-43 (fast_pbs_modulus_switch, offset 0, lut_count_log 0).
 * Returns a value in [0, 2N] (2N included, common.rs:20-25). */
uint64_t orc_modswitch(uint64_t x, uint32_t log2_poly_size)
{
    uint64_t out = x >> (64 - log2_poly_size - 2);
    out += 1;
    out >>= 1;
    return out;
}

/* core_crypto/algorithms/polynomial_algorithms.rs:315-354 (polynomial_wrapping_monic_monomial_div) */
void orc_monomial_div(uint64_t *out, const uint64_t *in, uint64_t degree, uint32_t N)
{
    uint64_t rem = degree % N;
    uint64_t cycles = degree / N;
    if (cycles % 2 == 0) {
        for (uint64_t i = 0; i < N - rem; i++) out[i] = in[rem + i];
        for (uint64_t i = 0; i < rem; i++) out[N - rem + i] = (uint64_t)0 - in[i];
    } else {
        for (uint64_t i = 0; i < N - rem; i++) out[i] = (uint64_t)0 - in[rem + i];
        for (uint64_t i = 0; i < rem; i++) out[N - rem + i] = in[i];
    }
}

/* polynomial_algorithms.rs:375-422 (polynomial_wrapping_monic_monomial_mul) */
void orc_monomial_mul(uint64_t *out, const uint64_t *in, uint64_t degree, uint32_t N)
{
    uint64_t rem = degree % N;
    uint64_t cycles = degree / N;
    if (cycles % 2 == 0) {
        for (uint64_t i = 0; i < rem; i++) out[i] = (uint64_t)0 - in[N - rem + i];
        for (uint64_t i = 0; i < N - rem; i++) out[rem + i] = in[i];
    } else {
        for (uint64_t i = 0; i < rem; i++) out[i] = in[N - rem + i];
        for (uint64_t i = 0; i < N - rem; i++) out[rem + i] = (uint64_t)0 - in[i];
    }
}

/* polynomial_algorithms.rs:425-490 (polynomial_wrapping_monic_monomial_mul_and_subtract):
 * out = in * X^degree - in */
void orc_monomial_mul_and_subtract(uint64_t *out, const uint64_t *in, uint64_t degree, uint32_t N)
{
    uint64_t rem = degree % N;
    uint64_t cycles = degree / N;
    if (cycles % 2 == 0) {
        for (uint64_t i = 0; i < rem; i++) out[i] = ((uint64_t)0 - in[N - rem + i]) - in[i];
        for (uint64_t i = 0; i < N - rem; i++) out[rem + i] = in[i] - in[rem + i];
    } else {
        for (uint64_t i = 0; i < rem; i++) out[i] = in[N - rem + i] - in[i];
        for (uint64_t i = 0; i < N - rem; i++) out[rem + i] = ((uint64_t)0 - in[i]) - in[rem + i];
    }
}

/* core_crypto/algorithms/glwe_sample_extraction.rs:91-147 with nth = 0:
 * body = glwe_body[0]; per mask poly: reverse, negate the first N-1, rotate_left(N-1). */
void orc_sample_extract(const uint64_t *glwe, uint32_t k, uint32_t N, uint64_t *lwe_out)
{
    lwe_out[(size_t)k * N] = glwe[(size_t)k * N + 0];
    for (uint32_t p = 0; p < k; p++) {
        const uint64_t *a = glwe + (size_t)p * N;
        uint64_t *o = lwe_out + (size_t)p * N;
        /* reversed r[i] = a[N-1-i]; negated for i < N-1; rotate_left(N-1): o[j] = r[(j+N-1) % N] */
        o[0] = a[0];
        for (uint32_t j = 1; j < N; j++) o[j] = (uint64_t)0 - a[N - j];
    }
}

/* core_crypto/algorithms/slice_algorithms.rs:363-455 (slice_wrapping_sub_scalar_mul_assign) */
void orc_slice_sub_scalar_mul(uint64_t *lhs, const uint64_t *rhs, uint64_t scalar, size_t len)
{
    for (size_t i = 0; i < len; i++) lhs[i] -= rhs[i] * scalar;
}

/* polynomial_algorithms.rs:130-199 (polynomial_wrapping_add_mul_assign): out += a*b mod X^N+1,
 * schoolbook */
void orc_poly_add_mul(uint64_t *out, const uint64_t *a, const uint64_t *b, uint32_t N)
{
    for (uint32_t i = 0; i < N; i++) {
        for (uint32_t j = 0; j < N; j++) {
            uint64_t prod = a[i] * b[j];
            if (i + j < N) out[i + j] += prod; else out[i + j - N] -= prod;
        }
    }
}

/* core_crypto/algorithms/lwe_keyswitch.rs:96-170 (keyswitch_lwe_ciphertext) with
 * slice_wrapping_sub_scalar_mul_assign (slice_algorithms.rs:363-455).
 * KSK layout [kN][ks_level blocks, level l first ... level 1 last][n+1]
 * (entities/lwe_keyswitch_key.rs:102-108; generation order lwe_keyswitch_key_generation.rs:109-120). */
void orc_keyswitch(const orc_params *p, const uint64_t *ksk, const uint64_t *lwe_in, uint64_t *lwe_out)
{
    const uint32_t in_dim = p->glwe_dim * p->poly_size;
    const uint32_t out_size = p->lwe_dim + 1;
    memset(lwe_out, 0, sizeof(uint64_t) * out_size);
    lwe_out[p->lwe_dim] = lwe_in[in_dim];
    int64_t terms[64];
    for (uint32_t j = 0; j < in_dim; j++) {
        orc_decompose(lwe_in[j], p->ks_base_log, p->ks_level, terms);
        const uint64_t *block = ksk + (size_t)j * p->ks_level * out_size;
        for (uint32_t l = 0; l < p->ks_level; l++) {
            const uint64_t d = (uint64_t)terms[l];
            const uint64_t *row = block + (size_t)l * out_size;
            orc_slice_sub_scalar_mul(lwe_out, row, d, out_size);
        }
    }
}

/* shortint/engine/mod.rs:72-128 (fill_accumulator) / server_key/mod.rs:383-399.
 * table[i] = f(i) for i in 0..message_modulus*carry_modulus. Returns max f (the degree). */
uint64_t orc_lut_generate(const orc_params *p, const uint64_t *table, uint64_t *glwe_out)
{
    const uint32_t N = p->poly_size, k = p->glwe_dim;
    const uint32_t modulus_sup = p->message_modulus * p->carry_modulus;
    const uint32_t box_size = N / modulus_sup;
    const uint64_t delta = ((uint64_t)1 << 63) / modulus_sup;
    memset(glwe_out, 0, sizeof(uint64_t) * (size_t)k * N);
    uint64_t *body = glwe_out + (size_t)k * N;
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * N);
    uint64_t max_value = 0;
    for (uint32_t i = 0; i < modulus_sup; i++) {
        uint64_t f_eval = table[i];
        if (f_eval > max_value) max_value = f_eval;
        for (uint32_t j = 0; j < box_size; j++) tmp[i * box_size + j] = f_eval * delta;
    }
    const uint32_t half = box_size / 2;
    for (uint32_t j = 0; j < half; j++) tmp[j] = (uint64_t)0 - tmp[j];
    for (uint32_t j = 0; j < N; j++) body[j] = tmp[(j + half) % N]; /* rotate_left(half) */
    free(tmp);
    return max_value;
}

/* shortint/server_key/mod.rs:763-781 (trivial_pbs_assign): returns the new body */
uint64_t orc_trivial_pbs(const orc_params *p, const uint64_t *lut_glwe, uint64_t body)
{
    const uint32_t N = p->poly_size, k = p->glwe_dim;
    const uint64_t modulus_sup = (uint64_t)p->message_modulus * p->carry_modulus;
    const uint64_t delta = ((uint64_t)1 << 63) / modulus_sup;
    const uint64_t box_size = N / modulus_sup;
    uint64_t ct_value = body / delta;
    const uint64_t *lut_body = lut_glwe + (size_t)k * N;
    if (ct_value >= modulus_sup) {
        ct_value %= modulus_sup;
        return (uint64_t)0 - lut_body[ct_value * box_size];
    }
    return lut_body[ct_value * box_size];
}

/* core_crypto/algorithms/lwe_linear_algebra.rs:68 (lwe_ciphertext_add_assign) */
void orc_lwe_add_assign(uint64_t *lhs, const uint64_t *rhs, size_t len)
{
    for (size_t i = 0; i < len; i++) lhs[i] += rhs[i];
}

/* lwe_linear_algebra.rs:556 (lwe_ciphertext_cleartext_mul_assign) */
void orc_lwe_scalar_mul_assign(uint64_t *lhs, uint64_t scalar, size_t len)
{
    for (size_t i = 0; i < len; i++) lhs[i] *= scalar;
}

/* fft64/math/fft/x86.rs:25-81 (mm256_cvtpd_epi64), scalar model fft64/math/fft/tests.rs:243-297:
 * f64 -> i64 by manipulating the IEEE-754 bits; exact for |x| < 2^63, +-2^63 -> i64::MIN. */
uint64_t orc_f64_to_u64_wrap(double x)
{
    uint64_t bits;
    memcpy(&bits, &x, 8);
    uint64_t mantissa = (bits & 0xFFFFFFFFFFFFFull) | 0x10000000000000ull;
    uint64_t biased_exp = (bits >> 52) & 0x7FF;
    uint64_t sign = bits >> 63;
    uint64_t lshift = mantissa << 11;
    uint64_t rshift = 1086 - biased_exp; /* wraps for huge exponents, like the vector shift count */
    uint64_t v = rshift < 64 ? lshift >> rshift : 0;
    return sign ? (uint64_t)0 - v : v;
}

/* commons/math/torus/mod.rs:72-78 (from_torus) in the AVX2 formulation the x86 build executes
 * (fft64/math/fft/x86.rs:864-871: x - round_nearest_even(x), *2^64, round_nearest_even,
 * then f64->i64 by the bit-twiddle of x86.rs:25-81, which wraps +2^63 to i64::MIN). */
uint64_t orc_torus_from_f64(double x)
{
    double fract = x - rint(x);
    fract = rint(fract * 18446744073709551616.0);
    return orc_f64_to_u64_wrap(fract);
}

/* ======================================================================================
 * T1: negacyclic f64 FFT -- fixed schedule
 * ====================================================================================== */

struct orc_fft {
    uint32_t N;      /* polynomial size */
    uint32_t n;      /* N/2 complex points */
    uint32_t stages; /* radix-4 stages: log4(n), or log4(n / 2) behind one radix-2 stage when n = 2 * 4^k */
    uint32_t radix2; /* 1: n = 2 * 4^k */
    double *tw_re, *tw_im; /* twisties e^{i pi j / N}, j<n (fft/mod.rs:58-69) */
    double *w_re, *w_im;   /* W[k] = e^{-2 pi i k / n}, k<n */
};

/* sin/cos of a binary128 angle in [0, pi/4] by Taylor series (libgcc soft-float only: no libm,
 * so the tables are identical on every machine). */
static void sincos_q(__float128 x, __float128 *s, __float128 *c)
{
    __float128 x2 = x * x, term = x, sum = x;
    for (int i = 1; i < 40; i++) {
        term = -term * x2 / (__float128)((2 * i) * (2 * i + 1));
        sum += term;
    }
    *s = sum;
    term = 1; sum = 1;
    for (int i = 1; i < 40; i++) {
        term = -term * x2 / (__float128)((2 * i - 1) * (2 * i));
        sum += term;
    }
    *c = sum;
}

static __float128 pi_q(void)
{
    /* pi = 3.243F6A8885A3 08D313198A2E 03707344A409 3822299F31D0 ... (hex) */
    __float128 p1 = 0x3.243F6A8885A3p0;
    __float128 p2 = 0x08D313198A2Ep-96;
    __float128 p3 = 0x03707344A409p-144;
    __float128 p4 = 0x3822299F31D0p-192;
    return ((p4 + p3) + p2) + p1;
}

/* correctly rounded (sin, cos) of the angle `x` in [0, pi/2] given in binary128 */
static void sincos_round(__float128 x, double *s, double *c)
{
    __float128 pi = pi_q(), sq, cq;
    if (x <= pi / 4) {
        sincos_q(x, &sq, &cq);
    } else {
        sincos_q(pi / 2 - x, &cq, &sq);
    }
    *s = (double)sq;
    *c = (double)cq;
}

orc_fft *orc_fft_new(uint32_t poly_size)
{
    uint32_t n = poly_size / 2, stages = 0, radix2 = 0, lg = 0;
    if (poly_size < 8 || (poly_size & (poly_size - 1))) return NULL; /* a power of two */
    while ((1u << lg) < n) lg++;
    if (lg & 1u) radix2 = 1; /* n = 2 * 4^k: one radix-2 stage in front of the radix-4 ones */
    stages = lg / 2;
    orc_fft *f = (orc_fft *)malloc(sizeof(*f));
    f->N = poly_size; f->n = n; f->stages = stages; f->radix2 = radix2;
    f->tw_re = (double *)malloc(sizeof(double) * n);
    f->tw_im = (double *)malloc(sizeof(double) * n);
    f->w_re = (double *)malloc(sizeof(double) * n);
    f->w_im = (double *)malloc(sizeof(double) * n);
    /* Twisties::new(n): unit = PI / (2 n) in f64; (im, re) = (i as f64 * unit).sin_cos()
     * (fft/mod.rs:58-69).  sin/cos are evaluated at that f64 argument, correctly rounded. */
    const double unit = 3.14159265358979323846 / (2.0 * (double)n);
    for (uint32_t i = 0; i < n; i++) {
        double arg = (double)i * unit;
        sincos_round((__float128)arg, &f->tw_im[i], &f->tw_re[i]);
    }
    /* FFT twiddles: W[k] = exp(-2 pi i k / n) evaluated at the exact angle, by octant symmetry. */
    const __float128 pi = pi_q();
    for (uint32_t k = 0; k < n; k++) {
        /* angle = 2 pi k / n ; reduce to first quadrant */
        uint32_t quad = (4 * k) / n;            /* 0..3 */
        uint32_t r = k - quad * (n / 4);        /* 0 .. n/4-1 */
        double s, c;
        sincos_round(pi * (__float128)(2 * r) / (__float128)n, &s, &c);
        double cr, sr; /* cos, sin of the full angle */
        switch (quad) {
        case 0: cr = c; sr = s; break;
        case 1: cr = -s; sr = c; break;
        case 2: cr = -c; sr = -s; break;
        default: cr = s; sr = -c; break;
        }
        f->w_re[k] = cr;
        f->w_im[k] = -sr;
    }
    return f;
}

void orc_fft_free(orc_fft *f)
{
    if (!f) return;
    free(f->tw_re); free(f->tw_im); free(f->w_re); free(f->w_im); free(f);
}
const double *orc_fft_twist_re(const orc_fft *f) { return f->tw_re; }
const double *orc_fft_twist_im(const orc_fft *f) { return f->tw_im; }
const double *orc_fft_w_re(const orc_fft *f) { return f->w_re; }
const double *orc_fft_w_im(const orc_fft *f) { return f->w_im; }

/* Forward transform core: in-place radix-4 DIF on n complex points, output left in DIF position
 * order (no digit reversal).  Stands in for concrete-fft's `plan.fwd` (fft/mod.rs:513), whose
 * source is not in the tree.  Stage s: L = n >> 2s, Q = L/4, e = j * (n/L):
 *   t0=a+c t1=a-c t2=b+d t3=b-d ; y0=t0+t2 ; y2=t0-t2 ; y1=t1-i*t3 ; y3=t1+i*t3
 *   x[j]=y0 ; x[j+Q]=y1*W[e] ; x[j+2Q]=y2*W[2e] ; x[j+3Q]=y3*W[3e]   (no multiply when e==0)
 * complex product y*w: re = fma(y.re, w.re, -(y.im*w.im)) ; im = fma(y.re, w.im, y.im*w.re). */
static void fft_fwd(const orc_fft *f, double *re, double *im)
{
    const uint32_t n = f->n, n4 = f->radix2 ? n / 2 : n;
    if (f->radix2) { /* n = 2 * 4^k: radix-2 DIF stage: x[j] = a + b ; x[j + n/2] = (a - b) * W[j] (no multiply when j == 0) */
        const uint32_t h = n / 2;
        for (uint32_t j = 0; j < h; j++) {
            const double ar = re[j], ai = im[j], br = re[j + h], bi = im[j + h];
            const double dr = ar - br, di = ai - bi;
            re[j] = ar + br; im[j] = ai + bi;
            if (j == 0) { re[h] = dr; im[h] = di; }
            else {
                re[j + h] = fma(dr, f->w_re[j], -(di * f->w_im[j]));
                im[j + h] = fma(dr, f->w_im[j], di * f->w_re[j]);
            }
        }
    }
    for (uint32_t s = 0; s < f->stages; s++) {
        const uint32_t L = n4 >> (2 * s), Q = L / 4, step = n / L;
        for (uint32_t b = 0; b < n; b += L) {
            for (uint32_t j = 0; j < Q; j++) {
                const uint32_t i0 = b + j, i1 = i0 + Q, i2 = i1 + Q, i3 = i2 + Q;
                double ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
                double cr = re[i2], ci = im[i2], dr = re[i3], di = im[i3];
                double t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;
                double t2r = br + dr, t2i = bi + di, t3r = br - dr, t3i = bi - di;
                double y0r = t0r + t2r, y0i = t0i + t2i;
                double y2r = t0r - t2r, y2i = t0i - t2i;
                double y1r = t1r + t3i, y1i = t1i - t3r;
                double y3r = t1r - t3i, y3i = t1i + t3r;
                re[i0] = y0r; im[i0] = y0i;
                if (j == 0) {
                    re[i1] = y1r; im[i1] = y1i;
                    re[i2] = y2r; im[i2] = y2i;
                    re[i3] = y3r; im[i3] = y3i;
                } else {
                    const uint32_t e = j * step;
                    double w1r = f->w_re[e], w1i = f->w_im[e];
                    double w2r = f->w_re[2 * e], w2i = f->w_im[2 * e];
                    double w3r = f->w_re[3 * e], w3i = f->w_im[3 * e];
                    re[i1] = fma(y1r, w1r, -(y1i * w1i)); im[i1] = fma(y1r, w1i, y1i * w1r);
                    re[i2] = fma(y2r, w2r, -(y2i * w2i)); im[i2] = fma(y2r, w2i, y2i * w2r);
                    re[i3] = fma(y3r, w3r, -(y3i * w3i)); im[i3] = fma(y3r, w3i, y3i * w3r);
                }
            }
        }
    }
}

/* Inverse core: exact mirror (radix-4 DIT, unnormalised), stands in for `plan.inv`
 * (fft/mod.rs:533,553).  Stage s from last to first:
 *   a=x[j] ; b=x[j+Q]*conj(W[e]) ; c=x[j+2Q]*conj(W[2e]) ; d=x[j+3Q]*conj(W[3e])
 *   t0=a+c t1=a-c t2=b+d t3=b-d ; x[j]=t0+t2 ; x[j+2Q]=t0-t2 ; x[j+Q]=t1+i*t3 ; x[j+3Q]=t1-i*t3
 * x*conj(w): re = fma(x.re, w.re, x.im*w.im) ; im = fma(-x.re, w.im, x.im*w.re). */
static void fft_inv(const orc_fft *f, double *re, double *im)
{
    const uint32_t n = f->n, n4 = f->radix2 ? n / 2 : n;
    for (int32_t s = (int32_t)f->stages - 1; s >= 0; s--) {
        const uint32_t L = n4 >> (2 * s), Q = L / 4, step = n / L;
        for (uint32_t b = 0; b < n; b += L) {
            for (uint32_t j = 0; j < Q; j++) {
                const uint32_t i0 = b + j, i1 = i0 + Q, i2 = i1 + Q, i3 = i2 + Q;
                double ar = re[i0], ai = im[i0];
                double br, bi, cr, ci, dr, di;
                if (j == 0) {
                    br = re[i1]; bi = im[i1]; cr = re[i2]; ci = im[i2]; dr = re[i3]; di = im[i3];
                } else {
                    const uint32_t e = j * step;
                    double w1r = f->w_re[e], w1i = f->w_im[e];
                    double w2r = f->w_re[2 * e], w2i = f->w_im[2 * e];
                    double w3r = f->w_re[3 * e], w3i = f->w_im[3 * e];
                    br = fma(re[i1], w1r, im[i1] * w1i); bi = fma(-re[i1], w1i, im[i1] * w1r);
                    cr = fma(re[i2], w2r, im[i2] * w2i); ci = fma(-re[i2], w2i, im[i2] * w2r);
                    dr = fma(re[i3], w3r, im[i3] * w3i); di = fma(-re[i3], w3i, im[i3] * w3r);
                }
                double t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;
                double t2r = br + dr, t2i = bi + di, t3r = br - dr, t3i = bi - di;
                re[i0] = t0r + t2r; im[i0] = t0i + t2i;
                re[i2] = t0r - t2r; im[i2] = t0i - t2i;
                re[i1] = t1r - t3i; im[i1] = t1i + t3r;
                re[i3] = t1r + t3i; im[i3] = t1i - t3r;
            }
        }
    }
    if (f->radix2) { /* the mirror of the radix-2 stage: b' = x[j + n/2] * conj(W[j]) ; x[j] = a + b' ; x[j + n/2] = a - b' */
        const uint32_t h = n / 2;
        for (uint32_t j = 0; j < h; j++) {
            const double ar = re[j], ai = im[j];
            double br = re[j + h], bi = im[j + h];
            if (j != 0) {
                const double xr = br, xi = bi;
                br = fma(xr, f->w_re[j], xi * f->w_im[j]);
                bi = fma(-xr, f->w_im[j], xi * f->w_re[j]);
            }
            re[j] = ar + br; im[j] = ai + bi;
            re[j + h] = ar - br; im[j + h] = ai - bi;
        }
    }
}

/* fft/mod.rs:398-405 forward_as_integer -> forward_with_conv (:496-515) with convert_forward_integer
 * in its AVX2+FMA form (x86.rs:505-596): fold re=coef[j], im=coef[j+N/2] as signed -> f64,
 * twist: re = fmsub(in_re,w_re,in_im*w_im), im = fmadd(in_re,w_im,in_im*w_re); then plan.fwd. */
void orc_fft_forward_integer(const orc_fft *f, const int64_t *poly, double *out_re, double *out_im)
{
    const uint32_t n = f->n;
    for (uint32_t j = 0; j < n; j++) {
        double in_re = (double)poly[j], in_im = (double)poly[j + n];
        out_re[j] = fma(in_re, f->tw_re[j], -(in_im * f->tw_im[j]));
        out_im[j] = fma(in_re, f->tw_im[j], in_im * f->tw_re[j]);
    }
    fft_fwd(f, out_re, out_im);
}

/* fft/mod.rs:378-386 forward_as_torus with convert_forward_torus (:197-219): the signed value is
 * scaled by 2^-64 before the twist.  (The reference's scalar code does a plain c64*c64 here; the
 * oracle uses the same FMA twist as forward_as_integer -- one documented convention.) */
void orc_fft_forward_torus(const orc_fft *f, const uint64_t *poly, double *out_re, double *out_im)
{
    const uint32_t n = f->n;
    const double norm = 5.421010862427522170037264004349708557128906250e-20; /* 2^-64 */
    for (uint32_t j = 0; j < n; j++) {
        double in_re = (double)(int64_t)poly[j] * norm, in_im = (double)(int64_t)poly[j + n] * norm;
        out_re[j] = fma(in_re, f->tw_re[j], -(in_im * f->tw_im[j]));
        out_im[j] = fma(in_re, f->tw_im[j], in_im * f->tw_re[j]);
    }
    fft_fwd(f, out_re, out_im);
}

/* fft/mod.rs:487-494 add_backward_in_place_as_torus -> :539-557 with convert_add_backward_torus in
 * its AVX2+FMA form (x86.rs:823-874, 961-1044): plan.inv, then w' = w/n,
 * re = fmadd(in_re,w_re',in_im*w_im'), im = fnmadd(in_re,w_im',in_im*w_re'), from_torus, wrapping add. */
void orc_fft_add_backward_torus(const orc_fft *f, uint64_t *poly, double *four_re, double *four_im)
{
    const uint32_t n = f->n;
    const double norm = 1.0 / (double)n;
    fft_inv(f, four_re, four_im);
    for (uint32_t j = 0; j < n; j++) {
        double wr = norm * f->tw_re[j], wi = norm * f->tw_im[j];
        double mr = fma(four_re[j], wr, four_im[j] * wi);
        double mi = fma(-four_re[j], wi, four_im[j] * wr);
        poly[j] += orc_torus_from_f64(mr);
        poly[j + n] += orc_torus_from_f64(mi);
    }
}

/* core_crypto/algorithms/lwe_bootstrap_key_conversion.rs:99-152 -> fft64/crypto/bootstrap.rs:182-208
 * -> ggsw.rs:252-271: forward_as_torus on every polynomial, same container order. */
void orc_bsk_to_fourier(const orc_params *p, const uint64_t *bsk_std, double *bsk_re, double *bsk_im)
{
    orc_fft *f = orc_fft_new(p->poly_size);
    const size_t polys = (size_t)p->lwe_dim * p->pbs_level * (p->glwe_dim + 1) * (p->glwe_dim + 1);
    const uint32_t N = p->poly_size, n = N / 2;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < polys; i++)
        orc_fft_forward_torus(f, bsk_std + i * N, bsk_re + i * n, bsk_im + i * n);
    orc_fft_free(f);
}

/* fft64/crypto/ggsw.rs:477-598 (add_external_product_assign) + update_with_fmadd (:616-697)
 * + TensorSignedDecompositionLendingIter (fft64/math/decomposition.rs:11-86).
 * Levels are visited highest index first (ggsw.rs:524 `.rev()`), rows in order; the very first
 * (level,row) initialises the Fourier accumulator with a complex multiply, later ones use
 * multiply-add.  Complex kernels (pulp's are not in the tree; these are the oracle's convention):
 *   mul:     re = fma(a.re,b.re, -(a.im*b.im))          im = fma(a.re,b.im, a.im*b.re)
 *   mul_add: re = fma(a.re,b.re, -fma(a.im,b.im,-c.re)) im = fma(a.re,b.im, fma(a.im,b.re,c.im))
 * with a = ggsw element, b = fourier(decomposed glwe poly), c = accumulator. */
void orc_add_external_product(const orc_params *p, const orc_fft *f, uint64_t *glwe_out,
                              const double *ggsw_re, const double *ggsw_im, const uint64_t *glwe_in)
{
    const uint32_t N = p->poly_size, n = N / 2, gs = p->glwe_dim + 1;
    const uint32_t bl = p->pbs_base_log, lv = p->pbs_level;
    const uint64_t mask = ((uint64_t)1 << bl) - 1;
    uint64_t *states = (uint64_t *)malloc(sizeof(uint64_t) * gs * N);
    int64_t *term = (int64_t *)malloc(sizeof(int64_t) * N);
    double *fr = (double *)malloc(sizeof(double) * n), *fi = (double *)malloc(sizeof(double) * n);
    double *outr = (double *)malloc(sizeof(double) * gs * n), *outi = (double *)malloc(sizeof(double) * gs * n);
    for (size_t i = 0; i < (size_t)gs * N; i++)
        states[i] = orc_closest_representable(glwe_in[i], bl, lv) >> (64 - bl * lv);
    int uninit = 1;
    for (int32_t level = (int32_t)lv; level >= 1; level--) {
        const double *lvl_re = ggsw_re + (size_t)(level - 1) * gs * gs * n;
        const double *lvl_im = ggsw_im + (size_t)(level - 1) * gs * gs * n;
        for (uint32_t row = 0; row < gs; row++) {
            for (uint32_t j = 0; j < N; j++)
                term[j] = (int64_t)decompose_one_level(bl, &states[(size_t)row * N + j], mask);
            orc_fft_forward_integer(f, term, fr, fi);
            for (uint32_t col = 0; col < gs; col++) {
                const double *ar = lvl_re + ((size_t)row * gs + col) * n;
                const double *ai = lvl_im + ((size_t)row * gs + col) * n;
                double *cr = outr + (size_t)col * n, *ci = outi + (size_t)col * n;
                if (uninit) {
                    for (uint32_t q = 0; q < n; q++) {
                        cr[q] = fma(ar[q], fr[q], -(ai[q] * fi[q]));
                        ci[q] = fma(ar[q], fi[q], ai[q] * fr[q]);
                    }
                } else {
                    for (uint32_t q = 0; q < n; q++) {
                        double r = fma(ar[q], fr[q], -fma(ai[q], fi[q], -cr[q]));
                        double i = fma(ar[q], fi[q], fma(ai[q], fr[q], ci[q]));
                        cr[q] = r; ci[q] = i;
                    }
                }
            }
            uninit = 0;
        }
    }
    for (uint32_t col = 0; col < gs; col++)
        orc_fft_add_backward_torus(f, glwe_out + (size_t)col * N, outr + (size_t)col * n, outi + (size_t)col * n);
    free(states); free(term); free(fr); free(fi); free(outr); free(outi);
}

/* fft64/crypto/bootstrap.rs:242-331 (blind_rotate_assign), native 2^64 modulus. */
void orc_blind_rotate(const orc_params *p, const orc_fft *f, const double *bsk_re, const double *bsk_im,
                      uint64_t *acc, const uint64_t *lwe_small)
{
    const uint32_t N = p->poly_size, gs = p->glwe_dim + 1, n = N / 2;
    uint32_t log2N = 0;
    while ((1u << log2N) < N) log2N++;
    const size_t ggsw_len = (size_t)p->pbs_level * gs * gs * n;
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * gs * N);
    uint64_t *ct1 = (uint64_t *)malloc(sizeof(uint64_t) * gs * N);
    const uint64_t b_hat = orc_modswitch(lwe_small[p->lwe_dim], log2N);
    memcpy(tmp, acc, sizeof(uint64_t) * gs * N);
    for (uint32_t q = 0; q < gs; q++) orc_monomial_div(acc + (size_t)q * N, tmp + (size_t)q * N, b_hat, N);
    for (uint32_t i = 0; i < p->lwe_dim; i++) {
        if (lwe_small[i] == 0) continue; /* bootstrap.rs:281 */
        const uint64_t a_hat = orc_modswitch(lwe_small[i], log2N);
        for (uint32_t q = 0; q < gs; q++)
            orc_monomial_mul_and_subtract(ct1 + (size_t)q * N, acc + (size_t)q * N, a_hat, N);
        orc_add_external_product(p, f, acc, bsk_re + i * ggsw_len, bsk_im + i * ggsw_len, ct1);
    }
    free(tmp); free(ct1);
}

/* bootstrap.rs:333-364 (bootstrap) == lwe_programmable_bootstrapping.rs:1067-1111 */
void orc_pbs(const orc_params *p, const orc_fft *f, const double *bsk_re, const double *bsk_im,
             const uint64_t *lwe_small, const uint64_t *lut_glwe, uint64_t *lwe_big_out)
{
    const uint32_t N = p->poly_size, gs = p->glwe_dim + 1;
    uint64_t *acc = (uint64_t *)malloc(sizeof(uint64_t) * gs * N);
    memcpy(acc, lut_glwe, sizeof(uint64_t) * gs * N);
    orc_blind_rotate(p, f, bsk_re, bsk_im, acc, lwe_small);
    orc_sample_extract(acc, p->glwe_dim, N, lwe_big_out);
    free(acc);
}

/* shortint/server_key/mod.rs:783-857 (keyswitch_programmable_bootstrap_assign), crypto part */
void orc_ks_pbs(const orc_params *p, const orc_fft *f, const uint64_t *ksk, const double *bsk_re,
                const double *bsk_im, const uint64_t *lwe_big_in, const uint64_t *lut_glwe,
                uint64_t *lwe_big_out)
{
    uint64_t *small = (uint64_t *)malloc(sizeof(uint64_t) * (p->lwe_dim + 1));
    orc_keyswitch(p, ksk, lwe_big_in, small);
    orc_pbs(p, f, bsk_re, bsk_im, small, lut_glwe, lwe_big_out);
    free(small);
}

void orc_ks_pbs_batch(const orc_params *p, const orc_fft *f, const uint64_t *ksk, const double *bsk_re,
                      const double *bsk_im, const uint64_t *lwe_in, const uint64_t *luts,
                      const uint32_t *lut_idx, uint64_t *lwe_out, size_t B, int threads)
{
    const size_t big = (size_t)p->glwe_dim * p->poly_size + 1;
    const size_t lut_len = (size_t)(p->glwe_dim + 1) * p->poly_size;
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (size_t i = 0; i < B; i++)
        orc_ks_pbs(p, f, ksk, bsk_re, bsk_im, lwe_in + i * big, luts + lut_idx[i] * lut_len, lwe_out + i * big);
}

void orc_keyswitch_batch(const orc_params *p, const uint64_t *ksk, const uint64_t *lwe_in,
                         uint64_t *lwe_out, size_t B, int threads)
{
    const size_t big = (size_t)p->glwe_dim * p->poly_size + 1, small = p->lwe_dim + 1;
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (size_t i = 0; i < B; i++) orc_keyswitch(p, ksk, lwe_in + i * big, lwe_out + i * small);
}

void orc_pbs_batch(const orc_params *p, const orc_fft *f, const double *bsk_re, const double *bsk_im,
                   const uint64_t *lwe_in, const uint64_t *luts, const uint32_t *lut_idx,
                   uint64_t *lwe_out, size_t B, int threads)
{
    const size_t big = (size_t)p->glwe_dim * p->poly_size + 1, small = p->lwe_dim + 1;
    const size_t lut_len = (size_t)(p->glwe_dim + 1) * p->poly_size;
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (size_t i = 0; i < B; i++)
        orc_pbs(p, f, bsk_re, bsk_im, lwe_in + i * small, luts + lut_idx[i] * lut_len, lwe_out + i * big);
}

/* ======================================================================================
 * Test-data side: seeded RNG, key generation, encryption (distribution-compatible with the
 * reference, not stream-compatible with its AES-CTR CSPRNG -- concrete-csprng is out of scope)
 * ====================================================================================== */

static uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void orc_rng_seed(orc_rng *r, uint64_t seed)
{
    for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&seed);
}
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t orc_rng_u64(orc_rng *r) /* xoshiro256** */
{
    uint64_t *s = r->s;
    const uint64_t result = rotl64(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl64(s[3], 45);
    return result;
}
/* Gaussian torus noise: commons/math/random/gaussian.rs:13-42 draws a normal of the given std and
 * maps it through from_torus; here Box-Muller. */
uint64_t orc_rng_noise(orc_rng *r, double std)
{
    double u1 = ((double)(orc_rng_u64(r) >> 11) + 1.0) * (1.0 / 9007199254740993.0);
    double u2 = (double)(orc_rng_u64(r) >> 11) * (1.0 / 9007199254740992.0);
    double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
    return orc_torus_from_f64(z * std);
}
/* lwe_secret_key_generation.rs / glwe_secret_key_generation.rs: uniform binary */
void orc_gen_binary_key(orc_rng *r, uint64_t *key, size_t len)
{
    for (size_t i = 0; i < len; i++) key[i] = orc_rng_u64(r) >> 63;
}
/* lwe_encryption.rs:61-117: b = <a,s> + plaintext + e */
void orc_lwe_encrypt(orc_rng *r, const uint64_t *sk, uint32_t dim, double std, uint64_t plaintext, uint64_t *ct)
{
    uint64_t dot = 0;
    for (uint32_t i = 0; i < dim; i++) {
        ct[i] = orc_rng_u64(r);
        dot += ct[i] * sk[i];
    }
    ct[dim] = dot + plaintext + orc_rng_noise(r, std);
}
/* lwe_encryption.rs:502-547 */
uint64_t orc_lwe_decrypt(const uint64_t *sk, uint32_t dim, const uint64_t *ct)
{
    uint64_t dot = 0;
    for (uint32_t i = 0; i < dim; i++) dot += ct[i] * sk[i];
    return ct[dim] - dot;
}

/* body += mask * S (negacyclic, binary S) */
static void negacyclic_mul_binary_add(uint64_t *body, const uint64_t *mask, const uint64_t *s, uint32_t N)
{
    for (uint32_t d = 0; d < N; d++) {
        if (!s[d]) continue;
        /* mask * X^d */
        for (uint32_t i = 0; i < N - d; i++) body[i + d] += mask[i];
        for (uint32_t i = N - d; i < N; i++) body[i + d - N] -= mask[i];
    }
}
/* glwe_encryption.rs:17-60 (encrypt_glwe_ciphertext_assign): body += sum mask_i*S_i + e */
static void glwe_encrypt_assign(orc_rng *r, const orc_params *p, const uint64_t *glwe_sk, uint64_t *glwe)
{
    const uint32_t N = p->poly_size, k = p->glwe_dim;
    uint64_t *body = glwe + (size_t)k * N;
    for (size_t i = 0; i < (size_t)k * N; i++) glwe[i] = orc_rng_u64(r);
    for (uint32_t j = 0; j < N; j++) body[j] += orc_rng_noise(r, p->glwe_std);
    for (uint32_t q = 0; q < k; q++)
        negacyclic_mul_binary_add(body, glwe + (size_t)q * N, glwe_sk + (size_t)q * N, N);
}
size_t orc_bsk_len(const orc_params *p)
{
    return (size_t)p->lwe_dim * p->pbs_level * (p->glwe_dim + 1) * (p->glwe_dim + 1) * p->poly_size;
}
size_t orc_ksk_len(const orc_params *p)
{
    return (size_t)p->glwe_dim * p->poly_size * p->ks_level * (p->lwe_dim + 1);
}
/* lwe_bootstrap_key_generation.rs:113-134 + ggsw_encryption.rs:72-151,300-331.
 * Layout [n][level 1..l][row 0..k][poly 0..k][N] (entities/lwe_bootstrap_key.rs:99-106). */
void orc_gen_bsk(orc_rng *r, const orc_params *p, const uint64_t *lwe_sk, const uint64_t *glwe_sk, uint64_t *bsk_std)
{
    const uint32_t N = p->poly_size, k = p->glwe_dim, gs = k + 1;
    for (uint32_t i = 0; i < p->lwe_dim; i++) {
        for (uint32_t li = 0; li < p->pbs_level; li++) {
            const uint64_t factor = ((uint64_t)0 - lwe_sk[i]) * ((uint64_t)1 << (64 - p->pbs_base_log * (li + 1)));
            for (uint32_t row = 0; row < gs; row++) {
                uint64_t *glwe = bsk_std + ((((size_t)i * p->pbs_level + li) * gs + row) * gs) * N;
                uint64_t *body = glwe + (size_t)k * N;
                if (row < k) {
                    for (uint32_t j = 0; j < N; j++) body[j] = glwe_sk[(size_t)row * N + j] * factor;
                } else {
                    memset(body, 0, sizeof(uint64_t) * N);
                    body[0] = (uint64_t)0 - factor;
                }
                glwe_encrypt_assign(r, p, glwe_sk, glwe);
            }
        }
    }
}
/* lwe_keyswitch_key_generation.rs:102-129: block j = encryptions of s_j * 2^(64 - b*level),
 * level = l .. 1 */
void orc_gen_ksk(orc_rng *r, const orc_params *p, const uint64_t *big_sk, const uint64_t *small_sk, uint64_t *ksk)
{
    const uint32_t in_dim = p->glwe_dim * p->poly_size, out_size = p->lwe_dim + 1;
    for (uint32_t j = 0; j < in_dim; j++) {
        for (uint32_t b = 0; b < p->ks_level; b++) {
            const uint32_t level = p->ks_level - b;
            const uint64_t pt = big_sk[j] << (64 - p->ks_base_log * level);
            orc_lwe_encrypt(r, small_sk, p->lwe_dim, p->lwe_std, pt, ksk + ((size_t)j * p->ks_level + b) * out_size);
        }
    }
}
/* shortint/engine/client_side.rs:58-85 */
uint64_t orc_shortint_encode(const orc_params *p, uint64_t message)
{
    const uint64_t delta = ((uint64_t)1 << 63) / ((uint64_t)p->message_modulus * p->carry_modulus);
    return message * delta;
}
/* shortint/client_key/mod.rs:281-302 (decrypt_message_and_carry, decode part) */
uint64_t orc_shortint_decode(const orc_params *p, uint64_t plaintext)
{
    const uint64_t delta = ((uint64_t)1 << 63) / ((uint64_t)p->message_modulus * p->carry_modulus);
    const uint64_t rounding_bit = delta >> 1;
    const uint64_t rounding = (plaintext & rounding_bit) << 1;
    return (plaintext + rounding) / delta;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ======================================================================================
 * Multi-bit PBS (core_crypto/algorithms/lwe_multi_bit_programmable_bootstrapping.rs:18-545,
 * lwe_multi_bit_bootstrap_key_generation.rs:87-170, 401-427).  grouping factor g: the LWE mask is
 * taken g elements at a time; a group's 2^g GGSWs encrypt the products prod_i (s_i or 1 - s_i)
 * (ggsw 0 = all keys inverted: the constant term).  Per group:
 *   M = ggsw_0 + sum_{idx >= 1} factor_idx * (ggsw_idx * mono_idx)     in the Fourier domain
 *   acc <- 0 + external_product(M, acc)
 * where X^d (d = modswitch(sum of the selected mask elements)) has the Fourier image
 * factor * mono:  mono[pos] = W[(d' * freq(pos)) mod n] (the transform of the unit vector e_{d'},
 * d' = d mod n the folded degree: `plan.fwd_monomial`, fft/mod.rs:432) and
 * factor = +-(1 or i) * twist[d'] (fft/mod.rs:416-444).  concrete-fft's fwd_monomial is not in the
 * tree; the oracle takes the table entry W[...] (the exact root of unity, correctly rounded).
 * Complex kernels: lhs*rhs then c64s_mul_adde(factor, ., out) (ggsw.rs:699-754) in the oracle's
 * mul / mul_add convention.  The reference's default multi-bit execution applies the groups in a
 * thread-dependent order (work queue); the oracle (like the `deterministic` variant, :548-940)
 * applies them in key order.
 * ====================================================================================== */

size_t orc_mb_bsk_len(const orc_params *p, uint32_t g)
{
    return (size_t)(p->lwe_dim / g) * ((size_t)1 << g) * p->pbs_level * (p->glwe_dim + 1) * (p->glwe_dim + 1) * p->poly_size;
}

/* Layout [n/g groups][2^g ggsw][level][row][col][N] (entities/lwe_multi_bit_bootstrap_key.rs) */
void orc_mb_gen_bsk(orc_rng *r, const orc_params *p, uint32_t g, const uint64_t *lwe_sk, const uint64_t *glwe_sk, uint64_t *bsk_std)
{
    const uint32_t N = p->poly_size, k = p->glwe_dim, gs = k + 1, per = 1u << g;
    const size_t ggsw_len = (size_t)p->pbs_level * gs * gs * N;
    for (uint32_t grp = 0; grp < p->lwe_dim / g; grp++) {
        for (uint32_t idx = 0; idx < per; idx++) {
            uint64_t pt = 1; /* combine_key_bits */
            for (uint32_t b = 0; b < g; b++) {
                const uint32_t pos = g - (b + 1);
                const uint64_t inv = ((idx >> pos) & 1u) ^ 1u;
                pt *= lwe_sk[(size_t)grp * g + b] ^ inv;
            }
            uint64_t *ggsw = bsk_std + ((size_t)grp * per + idx) * ggsw_len;
            for (uint32_t li = 0; li < p->pbs_level; li++) {
                const uint64_t factor = ((uint64_t)0 - pt) * ((uint64_t)1 << (64 - p->pbs_base_log * (li + 1)));
                for (uint32_t row = 0; row < gs; row++) {
                    uint64_t *glwe = ggsw + (((size_t)li * gs + row) * gs) * N;
                    uint64_t *body = glwe + (size_t)k * N;
                    if (row < k) {
                        for (uint32_t j = 0; j < N; j++) body[j] = glwe_sk[(size_t)row * N + j] * factor;
                    } else {
                        memset(body, 0, sizeof(uint64_t) * N);
                        body[0] = (uint64_t)0 - factor;
                    }
                    glwe_encrypt_assign(r, p, glwe_sk, glwe);
                }
            }
        }
    }
}

void orc_polys_to_fourier(uint32_t poly_size, size_t polys, const uint64_t *std, double *re, double *im)
{
    orc_fft *f = orc_fft_new(poly_size);
    const uint32_t N = poly_size, n = N / 2;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < polys; i++) orc_fft_forward_torus(f, std + i * N, re + i * n, im + i * n);
    orc_fft_free(f);
}

/* natural frequency index of DIF position `pos` (base-4 digit reversal over `stages` digits) */
uint32_t orc_fft_freq_of_pos(uint32_t pos, uint32_t stages)
{
    uint32_t f = 0;
    for (uint32_t s = 0; s < stages; s++) {
        f = (f << 2) | (pos & 3u);
        pos >>= 2;
    }
    return f;
}

/* factor and folded degree of X^degree (fft/mod.rs:408-445) */
static void monomial_factor(const orc_fft *f, uint64_t degree, uint32_t *folded, double *fr, double *fi)
{
    const uint32_t N = f->N, n = f->n;
    const int negate = (degree / N) % 2 == 1;
    uint32_t d = (uint32_t)(degree % N);
    const int imag = d >= n;
    if (imag) d -= n;
    double tr = f->tw_re[d], ti = f->tw_im[d];
    if (imag) { /* i * (tr + i ti) = -ti + i tr */
        const double t = tr;
        tr = -ti;
        ti = t;
    }
    if (negate) { tr = -tr; ti = -ti; }
    *folded = d; *fr = tr; *fi = ti;
}

/* prepare_multi_bit_ggsw_mem_optimized (:18-81): out (one Fourier GGSW) from the group's 2^g GGSWs */
void orc_mb_prepare_ggsw(const orc_params *p, uint32_t g, const orc_fft *f, const double *grp_re, const double *grp_im,
                         const uint64_t *mask_elems, double *out_re, double *out_im)
{
    const uint32_t N = p->poly_size, n = N / 2, gs = p->glwe_dim + 1, per = 1u << g;
    const size_t len = (size_t)p->pbs_level * gs * gs * n;
    uint32_t log2N = 0;
    while ((1u << log2N) < N) log2N++;
    memcpy(out_re, grp_re, sizeof(double) * len);
    memcpy(out_im, grp_im, sizeof(double) * len);
    for (uint32_t idx = 1; idx < per; idx++) {
        uint64_t deg = 0;
        for (uint32_t m = 0; m < g; m++) {
            const uint32_t pos = g - (m + 1);
            deg += (uint64_t)((idx >> pos) & 1u) * mask_elems[m];
        }
        const uint64_t switched = orc_modswitch(deg, log2N);
        uint32_t d;
        double fr, fi;
        monomial_factor(f, switched, &d, &fr, &fi);
        const double *gr = grp_re + (size_t)idx * len, *gi = grp_im + (size_t)idx * len;
        for (size_t e = 0; e < len; e++) {
            const uint32_t pos = (uint32_t)(e % n);
            const uint32_t w = (uint32_t)(((uint64_t)d * orc_fft_freq_of_pos(pos, f->stages)) % n);
            const double mr = f->w_re[w], mi = f->w_im[w];
            /* t = lhs * rhs */
            const double tr = fma(gr[e], mr, -(gi[e] * mi));
            const double ti = fma(gr[e], mi, gi[e] * mr);
            /* out = factor * t + out */
            const double r = fma(fr, tr, -fma(fi, ti, -out_re[e]));
            const double i = fma(fr, ti, fma(fi, tr, out_im[e]));
            out_re[e] = r; out_im[e] = i;
        }
    }
}

void orc_mb_blind_rotate(const orc_params *p, uint32_t g, const orc_fft *f, const double *bsk_re, const double *bsk_im,
                         uint64_t *acc, const uint64_t *lwe_small)
{
    const uint32_t N = p->poly_size, gs = p->glwe_dim + 1, n = N / 2, per = 1u << g;
    uint32_t log2N = 0;
    while ((1u << log2N) < N) log2N++;
    const size_t ggsw_len = (size_t)p->pbs_level * gs * gs * n;
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * gs * N);
    double *mr = (double *)malloc(sizeof(double) * ggsw_len), *mi = (double *)malloc(sizeof(double) * ggsw_len);
    const uint64_t b_hat = orc_modswitch(lwe_small[p->lwe_dim], log2N);
    memcpy(tmp, acc, sizeof(uint64_t) * gs * N);
    for (uint32_t q = 0; q < gs; q++) orc_monomial_div(acc + (size_t)q * N, tmp + (size_t)q * N, b_hat, N);
    for (uint32_t grp = 0; grp < p->lwe_dim / g; grp++) {
        orc_mb_prepare_ggsw(p, g, f, bsk_re + (size_t)grp * per * ggsw_len, bsk_im + (size_t)grp * per * ggsw_len,
                            lwe_small + (size_t)grp * g, mr, mi);
        memcpy(tmp, acc, sizeof(uint64_t) * gs * N);
        memset(acc, 0, sizeof(uint64_t) * gs * N);
        orc_add_external_product(p, f, acc, mr, mi, tmp);
    }
    free(tmp); free(mr); free(mi);
}

void orc_mb_pbs(const orc_params *p, uint32_t g, const orc_fft *f, const double *bsk_re, const double *bsk_im,
                const uint64_t *lwe_small, const uint64_t *lut_glwe, uint64_t *lwe_big_out)
{
    const uint32_t N = p->poly_size, gs = p->glwe_dim + 1;
    uint64_t *acc = (uint64_t *)malloc(sizeof(uint64_t) * gs * N);
    memcpy(acc, lut_glwe, sizeof(uint64_t) * gs * N);
    orc_mb_blind_rotate(p, g, f, bsk_re, bsk_im, acc, lwe_small);
    orc_sample_extract(acc, p->glwe_dim, N, lwe_big_out);
    free(acc);
}

void orc_mb_pbs_batch(const orc_params *p, uint32_t g, const orc_fft *f, const double *bsk_re, const double *bsk_im,
                      const uint64_t *lwe_small, const uint64_t *luts, const uint32_t *lut_idx, uint64_t *lwe_out,
                      size_t B, int threads)
{
    const size_t small = p->lwe_dim + 1, big = (size_t)p->glwe_dim * p->poly_size + 1;
    const size_t lut_len = (size_t)(p->glwe_dim + 1) * p->poly_size;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic) num_threads(threads)
    for (size_t i = 0; i < B; i++)
        orc_mb_pbs(p, g, f, bsk_re, bsk_im, lwe_small + i * small, luts + lut_idx[i] * lut_len, lwe_out + i * big);
}

void orc_mb_ks_pbs_batch(const orc_params *p, uint32_t g, const orc_fft *f, const uint64_t *ksk, const double *bsk_re,
                         const double *bsk_im, const uint64_t *lwe_in, const uint64_t *luts, const uint32_t *lut_idx,
                         uint64_t *lwe_out, size_t B, int threads)
{
    const size_t small = p->lwe_dim + 1, big = (size_t)p->glwe_dim * p->poly_size + 1;
    const size_t lut_len = (size_t)(p->glwe_dim + 1) * p->poly_size;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic) num_threads(threads)
    for (size_t i = 0; i < B; i++) {
        uint64_t *s = (uint64_t *)malloc(sizeof(uint64_t) * small);
        orc_keyswitch(p, ksk, lwe_in + i * big, s);
        orc_mb_pbs(p, g, f, bsk_re, bsk_im, s, luts + lut_idx[i] * lut_len, lwe_out + i * big);
        free(s);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Seeded (compressed) keys: the mask of every ciphertext of a key is the output of the CSPRNG
 * seeded with the key's compression seed, only the bodies are stored.
 *   concrete-csprng/src/generators/aes_ctr/{mod,generic,states,index}.rs, implem/soft/block_cipher.rs:
 *     AES-128 in counter mode, key = the 128-bit seed (native = little-endian bytes), block i =
 *     AES_key(i as 16 little-endian bytes); a new generator starts at TableIndex::SECOND (generic.rs:38),
 *     i.e. its first output byte is byte 1 of block 0; forks hand out consecutive byte ranges
 *     (generic.rs:84-121), so a key's masks are consecutive stretches of ONE stream, in memory order.
 *   core_crypto/commons/math/random/uniform.rs:15-24: a u64 is 8 consecutive bytes, little-endian.
 *   core_crypto/algorithms/seeded_lwe_bootstrap_key_decompression.rs, seeded_ggsw_ciphertext(_list)_decompression.rs,
 *   seeded_glwe_ciphertext_decompression.rs, seeded_lwe_keyswitch_key_decompression.rs,
 *   seeded_lwe_ciphertext_list_decompression.rs: per GLWE row k*N mask words then the stored body polynomial;
 *   per keyswitch-key ciphertext n mask words then the stored body.
 * The block cipher is pinned on the FIPS-197 appendix vectors the reference's own tests hold
 * (implem/soft/block_cipher.rs:89-113, implem/aesni/block_cipher.rs:189-229: key, key schedule, plaintext, ciphertext).
 * ------------------------------------------------------------------------------------------------ */
static uint8_t aes_sbox[256];
static int aes_sbox_ready = 0;
static uint8_t gf_mul(uint8_t a, uint8_t b)
{
    uint8_t r = 0;
    while (b) {
        if (b & 1) r ^= a;
        a = (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1b : 0));
        b >>= 1;
    }
    return r;
}
static void aes_init_sbox(void)
{
    if (aes_sbox_ready) return;
    for (int x = 0; x < 256; x++) {
        uint8_t inv = 0;
        if (x)
            for (int y = 1; y < 256; y++)
                if (gf_mul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
        uint8_t s = inv;
        for (int i = 1; i < 5; i++) s ^= (uint8_t)((inv << i) | (inv >> (8 - i)));
        aes_sbox[x] = (uint8_t)(s ^ 0x63);
    }
    aes_sbox_ready = 1;
}
const uint8_t *orc_aes_sbox(void)
{
    aes_init_sbox();
    return aes_sbox;
}
/* FIPS-197 5.2: 11 round keys of 16 bytes */
void orc_aes128_key_schedule(const uint8_t key[16], uint8_t rk[176])
{
    aes_init_sbox();
    memcpy(rk, key, 16);
    uint8_t rcon = 1;
    for (int i = 16; i < 176; i += 4) {
        uint8_t t[4] = {rk[i - 4], rk[i - 3], rk[i - 2], rk[i - 1]};
        if (i % 16 == 0) {
            const uint8_t t0 = t[0];
            t[0] = (uint8_t)(aes_sbox[t[1]] ^ rcon);
            t[1] = aes_sbox[t[2]];
            t[2] = aes_sbox[t[3]];
            t[3] = aes_sbox[t0];
            rcon = gf_mul(rcon, 2);
        }
        for (int j = 0; j < 4; j++) rk[i + j] = (uint8_t)(rk[i - 16 + j] ^ t[j]);
    }
}
/* FIPS-197 5.1: state byte (row r, column c) = in[r + 4 c] */
void orc_aes128_encrypt(const uint8_t rk[176], const uint8_t in[16], uint8_t out[16])
{
    uint8_t s[16], t[16];
    for (int i = 0; i < 16; i++) s[i] = (uint8_t)(in[i] ^ rk[i]);
    for (int round = 1; round <= 10; round++) {
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) t[r + 4 * c] = aes_sbox[s[r + 4 * ((c + r) & 3)]]; /* SubBytes + ShiftRows */
        if (round < 10) {
            for (int c = 0; c < 4; c++) { /* MixColumns */
                const uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                s[4 * c] = (uint8_t)(gf_mul(a0, 2) ^ gf_mul(a1, 3) ^ a2 ^ a3);
                s[4 * c + 1] = (uint8_t)(a0 ^ gf_mul(a1, 2) ^ gf_mul(a2, 3) ^ a3);
                s[4 * c + 2] = (uint8_t)(a0 ^ a1 ^ gf_mul(a2, 2) ^ gf_mul(a3, 3));
                s[4 * c + 3] = (uint8_t)(gf_mul(a0, 3) ^ a1 ^ a2 ^ gf_mul(a3, 2));
            }
        } else {
            memcpy(s, t, 16);
        }
        for (int i = 0; i < 16; i++) s[i] ^= rk[16 * round + i];
    }
    memcpy(out, s, 16);
}
/* bytes [offset, offset + len) of the output of a generator created from `seed` */
void orc_csprng_bytes(const uint8_t seed[16], uint64_t offset, uint8_t *out, size_t len)
{
    uint8_t rk[176], ctr[16], blk[16];
    orc_aes128_key_schedule(seed, rk);
    uint64_t pos = offset + 1; /* TableIndex::SECOND */
    uint64_t have = (uint64_t)-1;
    for (size_t i = 0; i < len; i++, pos++) {
        const uint64_t b = pos >> 4;
        if (b != have) {
            memset(ctr, 0, 16);
            for (int j = 0; j < 8; j++) ctr[j] = (uint8_t)(b >> (8 * j)); /* AesIndex(u128).to_ne_bytes() */
            orc_aes128_encrypt(rk, ctr, blk);
            have = b;
        }
        out[i] = blk[pos & 15];
    }
}
/* words [first, first + count) of the generator's u64 output (uniform.rs: from_le_bytes) */
void orc_csprng_words(const uint8_t seed[16], uint64_t first, uint64_t *out, size_t count)
{
    uint8_t *bytes = (uint8_t *)malloc(count * 8 + 8);
    orc_csprng_bytes(seed, first * 8, bytes, count * 8);
    for (size_t i = 0; i < count; i++) {
        uint64_t v = 0;
        for (int j = 0; j < 8; j++) v |= (uint64_t)bytes[8 * i + j] << (8 * j);
        out[i] = v;
    }
    free(bytes);
}
/* decompression: `units` ciphertexts of mask_words + body_words words; masks from the stream, bodies as stored */
void orc_seeded_expand(const uint8_t seed[16], const uint64_t *bodies, size_t units, size_t mask_words, size_t body_words,
                       uint64_t *out)
{
    uint64_t *mask = (uint64_t *)malloc(units * mask_words * 8 + 8);
    orc_csprng_words(seed, 0, mask, units * mask_words);
    for (size_t u = 0; u < units; u++) {
        memcpy(out + u * (mask_words + body_words), mask + u * mask_words, mask_words * 8);
        memcpy(out + u * (mask_words + body_words) + mask_words, bodies + u * body_words, body_words * 8);
    }
    free(mask);
}
/* TEST DATA: turns an ordinary key into the seeded key with the same secret, noise and plaintexts: every mask
 * is replaced by the stream's, the body follows (b' = b - <a, s> + <a', s>).  GLWE rows (bootstrap keys) ... */
void orc_seeded_reseed_glwe_rows(const orc_params *p, const uint8_t seed[16], const uint64_t *glwe_sk, uint64_t *rows, size_t count,
                                 uint64_t *bodies_out)
{
    const uint32_t N = p->poly_size, k = p->glwe_dim;
    const size_t mw = (size_t)k * N, tw = mw + N;
    uint64_t *mask = (uint64_t *)malloc(count * mw * 8 + 8);
    uint64_t *tmp = (uint64_t *)malloc((size_t)N * 8);
    orc_csprng_words(seed, 0, mask, count * mw);
    for (size_t u = 0; u < count; u++) {
        uint64_t *glwe = rows + u * tw, *body = glwe + mw;
        memset(tmp, 0, (size_t)N * 8);
        for (uint32_t q = 0; q < k; q++) negacyclic_mul_binary_add(tmp, glwe + (size_t)q * N, glwe_sk + (size_t)q * N, N);
        for (uint32_t j = 0; j < N; j++) body[j] -= tmp[j];
        memcpy(glwe, mask + u * mw, mw * 8);
        for (uint32_t q = 0; q < k; q++) negacyclic_mul_binary_add(body, glwe + (size_t)q * N, glwe_sk + (size_t)q * N, N);
        memcpy(bodies_out + u * N, body, (size_t)N * 8);
    }
    free(mask);
    free(tmp);
}
/* ... and LWE ciphertexts (keyswitch keys) */
void orc_seeded_reseed_lwe_list(const uint8_t seed[16], const uint64_t *sk, uint32_t dim, uint64_t *list, size_t count,
                                uint64_t *bodies_out)
{
    uint64_t *mask = (uint64_t *)malloc(count * dim * 8 + 8);
    orc_csprng_words(seed, 0, mask, count * dim);
    for (size_t u = 0; u < count; u++) {
        uint64_t *ct = list + u * (dim + 1);
        uint64_t dot_old = 0, dot_new = 0;
        for (uint32_t i = 0; i < dim; i++) {
            dot_old += ct[i] * sk[i];
            ct[i] = mask[u * dim + i];
            dot_new += ct[i] * sk[i];
        }
        ct[dim] += dot_new - dot_old;
        bodies_out[u] = ct[dim];
    }
    free(mask);
}
