/* tfhe_oracle_ntt.c -- EXPERIMENT: the external product as EXACT integer arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  The reference multiplies polynomials through an f64 FFT (fft64/crypto/ggsw.rs:477-598): the
 * result carries rounding noise and its low bits depend on the transform.  Here the same
 * add_external_product -- same decomposition (ggsw.rs:515-517, fft64/math/decomposition.rs:11-86), same rows and columns --
 * is computed exactly in Z/2^64 with a number-theoretic transform:
 *
 *   modulus  p = 2^64 - 2^32 + 1 ("Goldilocks"; p - 1 = 2^32 (2^32 - 1), so 2^32-th roots of unity exist; 7 generates Z_p^*)
 *   negacyclic transform of size N: psi a primitive 2N-th root of unity; x_j psi^j, then a cyclic radix-2 NTT
 *   the u64 GGSW coefficients are split into three limbs of 22 / 22 / 20 bits, so that every convolution coefficient
 *       sum over (k+1) N terms of |digit| < 2^(B-1) times limb < 2^22     is  < 2^(B + 33) <= 2^56 in magnitude
 *   and is recovered exactly from its residue (centred lift); result = c0 + 2^22 c1 + 2^44 c2  (mod 2^64).
 *
 * orc_ntt_add_external_product therefore equals the schoolbook product (orc_poly_add_mul on the digit polynomials) BIT FOR
 * BIT -- T0, deterministic on every machine -- and its distance to orc_add_external_product is the noise of the f64
 * transform, which tests/test_ntt_experiment.py measures.  Cost: 2 forward + 6 inverse transforms of N points with 64-bit
 * modular multiplications per external product, against 2 + 2 complex f64 transforms of N/2 points.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "tfhe_oracle.h"

#define GP 0xFFFFFFFF00000001ull

static inline uint64_t gl_add(uint64_t a, uint64_t b)
{
    uint64_t s = a + b;
    if (s < a || s >= GP) s -= GP;
    return s;
}
static inline uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + GP - b; }
static inline uint64_t gl_mul(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % GP); }
static uint64_t gl_pow(uint64_t b, uint64_t e)
{
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_mul(b, b);
        e >>= 1;
    }
    return r;
}

struct orc_ntt {
    uint32_t N, logN;
    uint64_t *psi, *psi_inv; /* psi^j, psi^-j / N */
    uint64_t *w, *w_inv;     /* omega^k, omega^-k, k < N/2, omega = psi^2 */
};

orc_ntt *orc_ntt_new(uint32_t N)
{
    uint32_t logN = 0;
    while ((1u << logN) < N) logN++;
    if ((1u << logN) != N || logN > 31) return NULL;
    orc_ntt *t = (orc_ntt *)calloc(1, sizeof(orc_ntt));
    t->N = N;
    t->logN = logN;
    const uint64_t psi = gl_pow(7, (GP - 1) / (2ull * N)); /* primitive 2N-th root of unity */
    const uint64_t psi_inv = gl_pow(psi, GP - 2), n_inv = gl_pow(N, GP - 2);
    t->psi = (uint64_t *)malloc(8 * N); t->psi_inv = (uint64_t *)malloc(8 * N);
    t->w = (uint64_t *)malloc(8 * (N / 2)); t->w_inv = (uint64_t *)malloc(8 * (N / 2));
    uint64_t a = 1, b = n_inv;
    for (uint32_t j = 0; j < N; j++) {
        t->psi[j] = a; t->psi_inv[j] = b;
        a = gl_mul(a, psi); b = gl_mul(b, psi_inv);
    }
    const uint64_t om = gl_mul(psi, psi), om_inv = gl_mul(psi_inv, psi_inv);
    a = 1; b = 1;
    for (uint32_t k = 0; k < N / 2; k++) {
        t->w[k] = a; t->w_inv[k] = b;
        a = gl_mul(a, om); b = gl_mul(b, om_inv);
    }
    return t;
}
void orc_ntt_free(orc_ntt *t)
{
    if (!t) return;
    free(t->psi); free(t->psi_inv); free(t->w); free(t->w_inv); free(t);
}

/* cyclic radix-2 transforms: forward decimation in frequency (natural in, bit-reversed out), inverse decimation in time
 * (bit-reversed in, natural out): no permutation needed for pointwise products in between */
static void ntt_fwd(const orc_ntt *t, uint64_t *x)
{
    const uint32_t N = t->N;
    for (uint32_t len = N / 2, step = 1; len >= 1; len >>= 1, step <<= 1)
        for (uint32_t b = 0; b < N; b += 2 * len)
            for (uint32_t j = 0; j < len; j++) {
                const uint64_t u = x[b + j], v = x[b + j + len];
                x[b + j] = gl_add(u, v);
                x[b + j + len] = gl_mul(gl_sub(u, v), t->w[j * step]);
            }
}
static void ntt_inv(const orc_ntt *t, uint64_t *x)
{
    const uint32_t N = t->N;
    for (uint32_t len = 1, step = N / 2; len < N; len <<= 1, step >>= 1)
        for (uint32_t b = 0; b < N; b += 2 * len)
            for (uint32_t j = 0; j < len; j++) {
                const uint64_t u = x[b + j], v = gl_mul(x[b + j + len], t->w_inv[j * step]);
                x[b + j] = gl_add(u, v);
                x[b + j + len] = gl_sub(u, v);
            }
}
/* negacyclic forward transform of a polynomial given by residues */
static void nega_fwd(const orc_ntt *t, uint64_t *x)
{
    for (uint32_t j = 0; j < t->N; j++) x[j] = gl_mul(x[j], t->psi[j]);
    ntt_fwd(t, x);
}
static void nega_inv(const orc_ntt *t, uint64_t *x)
{
    ntt_inv(t, x);
    for (uint32_t j = 0; j < t->N; j++) x[j] = gl_mul(x[j], t->psi_inv[j]);
}

#define LIMBS 3
static const uint32_t limb_shift[LIMBS] = {0, 22, 44};
static const uint64_t limb_mask[LIMBS] = {0x3FFFFF, 0x3FFFFF, 0xFFFFF};

/* GGSW (standard domain, [level][row][col][N] u64) -> transformed limbs [level][row][col][limb][N] */
void orc_ntt_ggsw_prepare(const orc_params *p, const orc_ntt *t, const uint64_t *ggsw_std, uint64_t *ggsw_ntt)
{
    const uint32_t N = p->poly_size, gs = p->glwe_dim + 1;
    const size_t polys = (size_t)p->pbs_level * gs * gs;
    for (size_t q = 0; q < polys; q++)
        for (int l = 0; l < LIMBS; l++) {
            uint64_t *dst = ggsw_ntt + (q * LIMBS + l) * N;
            for (uint32_t j = 0; j < N; j++) dst[j] = (ggsw_std[q * N + j] >> limb_shift[l]) & limb_mask[l];
            nega_fwd(t, dst);
        }
}
size_t orc_ntt_ggsw_len(const orc_params *p)
{
    const size_t gs = p->glwe_dim + 1;
    return (size_t)p->pbs_level * gs * gs * LIMBS * p->poly_size;
}

/* glwe_out += ggsw (x) glwe_in exactly: the decomposition of orc_add_external_product, the products in Z/2^64 */
void orc_ntt_add_external_product(const orc_params *p, const orc_ntt *t, uint64_t *glwe_out, const uint64_t *ggsw_ntt,
                                  const uint64_t *glwe_in)
{
    const uint32_t N = p->poly_size, gs = p->glwe_dim + 1, bl = p->pbs_base_log, lv = p->pbs_level;
    const uint64_t mask = ((uint64_t)1 << bl) - 1;
    uint64_t *states = (uint64_t *)malloc(8 * (size_t)gs * N);
    uint64_t *dig = (uint64_t *)malloc(8 * (size_t)N);
    uint64_t *accl = (uint64_t *)calloc((size_t)gs * LIMBS * N, 8); /* [col][limb][N], transformed domain */
    for (size_t i = 0; i < (size_t)gs * N; i++) states[i] = orc_closest_representable(glwe_in[i], bl, lv) >> (64 - bl * lv);
    for (int32_t level = (int32_t)lv; level >= 1; level--)
        for (uint32_t row = 0; row < gs; row++) {
            for (uint32_t j = 0; j < N; j++) { /* decompose_one_level (fft64/math/decomposition.rs:78-86), as a residue */
                uint64_t *st = &states[(size_t)row * N + j];
                uint64_t res = *st & mask;
                *st >>= bl;
                uint64_t carry = ((res - 1) | *st) & res;
                carry >>= bl - 1;
                *st += carry;
                const int64_t d = (int64_t)(res - (carry << bl));
                dig[j] = d >= 0 ? (uint64_t)d : GP - (uint64_t)(-d);
            }
            nega_fwd(t, dig);
            for (uint32_t col = 0; col < gs; col++)
                for (int l = 0; l < LIMBS; l++) {
                    const uint64_t *g = ggsw_ntt + (((((size_t)(level - 1) * gs + row) * gs + col) * LIMBS) + l) * N;
                    uint64_t *a = accl + ((size_t)col * LIMBS + l) * N;
                    for (uint32_t j = 0; j < N; j++) a[j] = gl_add(a[j], gl_mul(dig[j], g[j]));
                }
        }
    for (uint32_t col = 0; col < gs; col++)
        for (int l = 0; l < LIMBS; l++) {
            uint64_t *a = accl + ((size_t)col * LIMBS + l) * N;
            nega_inv(t, a);
            for (uint32_t j = 0; j < N; j++) { /* centred lift: |value| < 2^56 */
                const uint64_t v = a[j] > GP / 2 ? (uint64_t)0 - (GP - a[j]) : a[j];
                glwe_out[(size_t)col * N + j] += v << limb_shift[l];
            }
        }
    free(states); free(dig); free(accl);
}

/* the same product by the schoolbook convolution of the reference's integer code (polynomial_algorithms.rs:130-199):
 * the ground truth the transform must reproduce */
void orc_schoolbook_add_external_product(const orc_params *p, uint64_t *glwe_out, const uint64_t *ggsw_std, const uint64_t *glwe_in)
{
    const uint32_t N = p->poly_size, gs = p->glwe_dim + 1, bl = p->pbs_base_log, lv = p->pbs_level;
    const uint64_t mask = ((uint64_t)1 << bl) - 1;
    uint64_t *states = (uint64_t *)malloc(8 * (size_t)gs * N);
    uint64_t *dig = (uint64_t *)malloc(8 * (size_t)N);
    for (size_t i = 0; i < (size_t)gs * N; i++) states[i] = orc_closest_representable(glwe_in[i], bl, lv) >> (64 - bl * lv);
    for (int32_t level = (int32_t)lv; level >= 1; level--)
        for (uint32_t row = 0; row < gs; row++) {
            for (uint32_t j = 0; j < N; j++) {
                uint64_t *st = &states[(size_t)row * N + j];
                uint64_t res = *st & mask;
                *st >>= bl;
                uint64_t carry = ((res - 1) | *st) & res;
                carry >>= bl - 1;
                *st += carry;
                dig[j] = res - (carry << bl);
            }
            for (uint32_t col = 0; col < gs; col++)
                orc_poly_add_mul(glwe_out + (size_t)col * N, dig, ggsw_std + ((((size_t)(level - 1) * gs + row) * gs + col)) * N, N);
        }
    free(states); free(dig);
}

/* blind rotation + sample extraction with EXACT external products: bootstrap.rs:242-364 with the product swapped */
void orc_ntt_pbs(const orc_params *p, const orc_ntt *t, const uint64_t *bsk_ntt, const uint64_t *lwe_small, const uint64_t *lut_glwe,
                 uint64_t *lwe_big_out)
{
    const uint32_t N = p->poly_size, gs = p->glwe_dim + 1;
    uint32_t log2N = 0;
    while ((1u << log2N) < N) log2N++;
    const size_t ggsw_len = orc_ntt_ggsw_len(p);
    uint64_t *acc = (uint64_t *)malloc(8 * (size_t)gs * N), *tmp = (uint64_t *)malloc(8 * (size_t)gs * N);
    uint64_t *ct1 = (uint64_t *)malloc(8 * (size_t)gs * N);
    memcpy(tmp, lut_glwe, 8 * (size_t)gs * N);
    const uint64_t b_hat = orc_modswitch(lwe_small[p->lwe_dim], log2N);
    for (uint32_t q = 0; q < gs; q++) orc_monomial_div(acc + (size_t)q * N, tmp + (size_t)q * N, b_hat, N);
    for (uint32_t i = 0; i < p->lwe_dim; i++) {
        if (lwe_small[i] == 0) continue;
        const uint64_t a_hat = orc_modswitch(lwe_small[i], log2N);
        for (uint32_t q = 0; q < gs; q++) orc_monomial_mul_and_subtract(ct1 + (size_t)q * N, acc + (size_t)q * N, a_hat, N);
        orc_ntt_add_external_product(p, t, acc, bsk_ntt + (size_t)i * ggsw_len, ct1);
    }
    orc_sample_extract(acc, p->glwe_dim, N, lwe_big_out);
    free(acc); free(tmp); free(ct1);
}

void orc_ntt_bsk_prepare(const orc_params *p, const orc_ntt *t, const uint64_t *bsk_std, uint64_t *bsk_ntt)
{
    const size_t gs = p->glwe_dim + 1, std_len = (size_t)p->pbs_level * gs * gs * p->poly_size, ntt_len = orc_ntt_ggsw_len(p);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < p->lwe_dim; i++) orc_ntt_ggsw_prepare(p, t, bsk_std + i * std_len, bsk_ntt + i * ntt_len);
}

void orc_ntt_pbs_batch(const orc_params *p, const orc_ntt *t, const uint64_t *bsk_ntt, const uint64_t *lwe_in, const uint64_t *luts,
                       const uint32_t *lut_idx, uint64_t *lwe_out, size_t B, int threads)
{
    const size_t big = (size_t)p->glwe_dim * p->poly_size + 1, small = p->lwe_dim + 1;
    const size_t lut_len = (size_t)(p->glwe_dim + 1) * p->poly_size;
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (size_t i = 0; i < B; i++) orc_ntt_pbs(p, t, bsk_ntt, lwe_in + i * small, luts + lut_idx[i] * lut_len, lwe_out + i * big);
}
