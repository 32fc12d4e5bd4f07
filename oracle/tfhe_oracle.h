/*
 * tfhe_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's shortint keyswitch + programmable
 * bootstrap path (M-Bln/tfhe-rs-string, a tfhe-rs 0.5.0 fork).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (tfhe_rs_string_amd/csrc) never includes, links or calls it.
 *
 * Parity status (see DESIGN.md "Parity tiers"):
 *   T0  integer functions: follow in-tree Rust line by line and are pinned by the
 *       reference's doc-test known answers (tests/test_oracle_golden.py).
 *   T1  f64 negacyclic FFT: the reference delegates to the un-vendored crate
 *       concrete-fft 0.3 (plan chosen by run-time timing) -> bit-level parity with the
 *       reference at that boundary is UNPINNED; the oracle fixes its own documented
 *       schedule ("radix-4 DIF/DIT, FMA twist") and the HIP path must
 *       match the oracle bit for bit.
 *   T2  decrypt-level agreement with the reference semantics.
 *
 * All reference paths are relative to /root/reference/tfhe/src.
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* shortint/parameters/mod.rs:703-717 (ClassicPBSParameters), reduced to what the path uses. */
typedef struct orc_params {
    uint32_t lwe_dim;       /* n   (742) */
    uint32_t glwe_dim;      /* k   (1)   */
    uint32_t poly_size;     /* N   (2048), N/2 must be a power of 4 for the fixed FFT schedule */
    uint32_t pbs_base_log;  /* 23 */
    uint32_t pbs_level;     /* 1  */
    uint32_t ks_base_log;   /* 3  */
    uint32_t ks_level;      /* 5  */
    uint32_t message_modulus; /* 4 */
    uint32_t carry_modulus;   /* 4 */
    double lwe_std;         /* 7.069849454709433e-6  */
    double glwe_std;        /* 2.9403601535432533e-16 */
} orc_params;

/* ------------------------------------------------------------------ T0: integer pieces */
uint64_t orc_closest_representable(uint64_t x, uint32_t base_log, uint32_t level);
/* terms[0] is level `level` (least significant digit) ... terms[level-1] is level 1 */
void orc_decompose(uint64_t x, uint32_t base_log, uint32_t level, int64_t *terms);
uint64_t orc_modswitch(uint64_t x, uint32_t log2_poly_size);
void orc_monomial_div(uint64_t *out, const uint64_t *in, uint64_t degree, uint32_t N);
void orc_monomial_mul(uint64_t *out, const uint64_t *in, uint64_t degree, uint32_t N);
void orc_monomial_mul_and_subtract(uint64_t *out, const uint64_t *in, uint64_t degree, uint32_t N);
void orc_sample_extract(const uint64_t *glwe, uint32_t k, uint32_t N, uint64_t *lwe_out);
void orc_keyswitch(const orc_params *p, const uint64_t *ksk, const uint64_t *lwe_in, uint64_t *lwe_out);
uint64_t orc_lut_generate(const orc_params *p, const uint64_t *table, uint64_t *glwe_out);
uint64_t orc_trivial_pbs(const orc_params *p, const uint64_t *lut_glwe, uint64_t body);
void orc_lwe_add_assign(uint64_t *lhs, const uint64_t *rhs, size_t len);
void orc_lwe_scalar_mul_assign(uint64_t *lhs, uint64_t scalar, size_t len);
uint64_t orc_torus_from_f64(double x);
uint64_t orc_f64_to_u64_wrap(double x);
void orc_slice_sub_scalar_mul(uint64_t *lhs, const uint64_t *rhs, uint64_t scalar, size_t len);
void orc_poly_add_mul(uint64_t *out, const uint64_t *a, const uint64_t *b, uint32_t N);

/* ------------------------------------------------------------------ T1: f64 FFT schedule */
typedef struct orc_fft orc_fft;
orc_fft *orc_fft_new(uint32_t poly_size);
void orc_fft_free(orc_fft *f);
const double *orc_fft_twist_re(const orc_fft *f);   /* N/2 values */
const double *orc_fft_twist_im(const orc_fft *f);
const double *orc_fft_w_re(const orc_fft *f);       /* N/2 values: cos(2 pi k/(N/2)) */
const double *orc_fft_w_im(const orc_fft *f);       /* -sin(2 pi k/(N/2)) */
/* out_re/out_im: N/2 doubles each, in DIF position order */
void orc_fft_forward_integer(const orc_fft *f, const int64_t *poly, double *out_re, double *out_im);
void orc_fft_forward_torus(const orc_fft *f, const uint64_t *poly, double *out_re, double *out_im);
/* destroys four_re/four_im */
void orc_fft_add_backward_torus(const orc_fft *f, uint64_t *poly, double *four_re, double *four_im);
/* bsk_std: [n][l][k+1][k+1][N] -> re/im planes [n][l][k+1][k+1][N/2] */
void orc_bsk_to_fourier(const orc_params *p, const uint64_t *bsk_std, double *bsk_re, double *bsk_im);
/* glwe_out += ggsw (x) glwe_in ; ggsw_re/im: one GGSW [l][k+1][k+1][N/2] */
void orc_add_external_product(const orc_params *p, const orc_fft *f, uint64_t *glwe_out,
                              const double *ggsw_re, const double *ggsw_im, const uint64_t *glwe_in);
void orc_blind_rotate(const orc_params *p, const orc_fft *f, const double *bsk_re, const double *bsk_im,
                      uint64_t *acc /* (k+1)N, in: LUT, out: rotated */, const uint64_t *lwe_small);
void orc_pbs(const orc_params *p, const orc_fft *f, const double *bsk_re, const double *bsk_im,
             const uint64_t *lwe_small, const uint64_t *lut_glwe, uint64_t *lwe_big_out);
void orc_ks_pbs(const orc_params *p, const orc_fft *f, const uint64_t *ksk, const double *bsk_re,
                const double *bsk_im, const uint64_t *lwe_big_in, const uint64_t *lut_glwe,
                uint64_t *lwe_big_out);
/* B ciphertexts, luts: L x (k+1)N, lut_idx: B ; OpenMP over the batch with `threads` threads */
void orc_ks_pbs_batch(const orc_params *p, const orc_fft *f, const uint64_t *ksk, const double *bsk_re,
                      const double *bsk_im, const uint64_t *lwe_in, const uint64_t *luts,
                      const uint32_t *lut_idx, uint64_t *lwe_out, size_t B, int threads);
void orc_keyswitch_batch(const orc_params *p, const uint64_t *ksk, const uint64_t *lwe_in,
                         uint64_t *lwe_out, size_t B, int threads);
void orc_pbs_batch(const orc_params *p, const orc_fft *f, const double *bsk_re, const double *bsk_im,
                   const uint64_t *lwe_in, const uint64_t *luts, const uint32_t *lut_idx,
                   uint64_t *lwe_out, size_t B, int threads);

/* ------------------------------------------------------------------ test-data side (keygen etc.) */
typedef struct orc_rng { uint64_t s[4]; } orc_rng;
void orc_rng_seed(orc_rng *r, uint64_t seed);
uint64_t orc_rng_u64(orc_rng *r);
uint64_t orc_rng_noise(orc_rng *r, double std);
void orc_gen_binary_key(orc_rng *r, uint64_t *key, size_t len);
void orc_lwe_encrypt(orc_rng *r, const uint64_t *sk, uint32_t dim, double std, uint64_t plaintext, uint64_t *ct);
uint64_t orc_lwe_decrypt(const uint64_t *sk, uint32_t dim, const uint64_t *ct);
void orc_gen_bsk(orc_rng *r, const orc_params *p, const uint64_t *lwe_sk, const uint64_t *glwe_sk, uint64_t *bsk_std);
void orc_gen_ksk(orc_rng *r, const orc_params *p, const uint64_t *big_sk, const uint64_t *small_sk, uint64_t *ksk);
uint64_t orc_shortint_encode(const orc_params *p, uint64_t message);
uint64_t orc_shortint_decode(const orc_params *p, uint64_t plaintext);

size_t orc_bsk_len(const orc_params *p);   /* u64 count of the standard BSK */
size_t orc_ksk_len(const orc_params *p);   /* u64 count of the KSK */
int orc_max_threads(void);


/* ------------------------------------------------------------------ multi-bit PBS (grouping factor g) */
size_t orc_mb_bsk_len(const orc_params *p, uint32_t g);
void orc_mb_gen_bsk(orc_rng *r, const orc_params *p, uint32_t g, const uint64_t *lwe_sk, const uint64_t *glwe_sk, uint64_t *bsk_std);
void orc_polys_to_fourier(uint32_t poly_size, size_t polys, const uint64_t *std, double *re, double *im);
uint32_t orc_fft_freq_of_pos(uint32_t pos, uint32_t stages);
void orc_mb_prepare_ggsw(const orc_params *p, uint32_t g, const orc_fft *f, const double *grp_re, const double *grp_im,
                         const uint64_t *mask_elems, double *out_re, double *out_im);
void orc_mb_blind_rotate(const orc_params *p, uint32_t g, const orc_fft *f, const double *bsk_re, const double *bsk_im,
                         uint64_t *acc, const uint64_t *lwe_small);
void orc_mb_pbs(const orc_params *p, uint32_t g, const orc_fft *f, const double *bsk_re, const double *bsk_im,
                const uint64_t *lwe_small, const uint64_t *lut_glwe, uint64_t *lwe_big_out);
void orc_mb_pbs_batch(const orc_params *p, uint32_t g, const orc_fft *f, const double *bsk_re, const double *bsk_im,
                      const uint64_t *lwe_small, const uint64_t *luts, const uint32_t *lut_idx, uint64_t *lwe_out,
                      size_t B, int threads);
void orc_mb_ks_pbs_batch(const orc_params *p, uint32_t g, const orc_fft *f, const uint64_t *ksk, const double *bsk_re,
                         const double *bsk_im, const uint64_t *lwe_in, const uint64_t *luts, const uint32_t *lut_idx,
                         uint64_t *lwe_out, size_t B, int threads);

/* ------------------------------------------------------------------ seeded (compressed) keys: AES-128-CTR CSPRNG */
const uint8_t *orc_aes_sbox(void);
void orc_aes128_key_schedule(const uint8_t key[16], uint8_t rk[176]);
void orc_aes128_encrypt(const uint8_t rk[176], const uint8_t in[16], uint8_t out[16]);
void orc_csprng_bytes(const uint8_t seed[16], uint64_t offset, uint8_t *out, size_t len);
void orc_csprng_words(const uint8_t seed[16], uint64_t first, uint64_t *out, size_t count);
void orc_seeded_expand(const uint8_t seed[16], const uint64_t *bodies, size_t units, size_t mask_words, size_t body_words,
                       uint64_t *out);
void orc_seeded_reseed_glwe_rows(const orc_params *p, const uint8_t seed[16], const uint64_t *glwe_sk, uint64_t *rows, size_t count,
                                 uint64_t *bodies_out);
void orc_seeded_reseed_lwe_list(const uint8_t seed[16], const uint64_t *sk, uint32_t dim, uint64_t *list, size_t count,
                                uint64_t *bodies_out);

/* tfhe_oracle_fast.c: tuned port of orc_ks_pbs_batch (k = 1, one PBS level; anything else falls through to the plain one);
 * bit-identical results.  bench.py's cpu_baseline leg only. */
void orcf_ks_pbs_batch(const orc_params *p, const orc_fft *f, const uint64_t *ksk, const double *bsk_re, const double *bsk_im,
                       const uint64_t *lwe_in, const uint64_t *luts, const uint32_t *lut_idx, uint64_t *lwe_out, size_t B,
                       int threads);

/* tfhe_oracle_ntt.c: EXPERIMENT -- the external product as exact integer arithmetic (negacyclic NTT modulo
 * 2^64 - 2^32 + 1 on 22-bit limbs of the key); equals the schoolbook product bit for bit. */
typedef struct orc_ntt orc_ntt;
orc_ntt *orc_ntt_new(uint32_t N);
void orc_ntt_free(orc_ntt *t);
size_t orc_ntt_ggsw_len(const orc_params *p);
void orc_ntt_ggsw_prepare(const orc_params *p, const orc_ntt *t, const uint64_t *ggsw_std, uint64_t *ggsw_ntt);
void orc_ntt_bsk_prepare(const orc_params *p, const orc_ntt *t, const uint64_t *bsk_std, uint64_t *bsk_ntt);
void orc_ntt_add_external_product(const orc_params *p, const orc_ntt *t, uint64_t *glwe_out, const uint64_t *ggsw_ntt,
                                  const uint64_t *glwe_in);
void orc_schoolbook_add_external_product(const orc_params *p, uint64_t *glwe_out, const uint64_t *ggsw_std, const uint64_t *glwe_in);
void orc_ntt_pbs(const orc_params *p, const orc_ntt *t, const uint64_t *bsk_ntt, const uint64_t *lwe_small, const uint64_t *lut_glwe,
                 uint64_t *lwe_big_out);
void orc_ntt_pbs_batch(const orc_params *p, const orc_ntt *t, const uint64_t *bsk_ntt, const uint64_t *lwe_in, const uint64_t *luts,
                       const uint32_t *lut_idx, uint64_t *lwe_out, size_t B, int threads);

#ifdef __cplusplus
}
#endif
#endif
