/*
 * tfhe_hip_experimental.h -- EXPERIMENTS next to the drop-in boundary (include/tfhe_hip.h); nothing here is on the
 * production path.
 *
 * the external product of the blind rotation (add_external_product_assign,
 * tfhe/src/core_crypto/fft_impl/fft64/crypto/ggsw.rs:477-598) computed as EXACT integer arithmetic -- a negacyclic
 * number-theoretic transform modulo 2^64 - 2^32 + 1 on 22-bit limbs of the standard-domain GGSW -- instead of the
 * reference's f64 FFT: the result equals the schoolbook product of the reference's integer code
 * (core_crypto/algorithms/polynomial_algorithms.rs:130-199) bit for bit, on every machine.  What it costs is in
 * DESIGN.md section 8 (the experiment's conclusion: ~20 x the f64 path on this hardware, so it stays an experiment).
 */
#ifndef TFHE_HIP_EXPERIMENTAL_H
#define TFHE_HIP_EXPERIMENTAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* glwe_acc[ct] += ggsw (x) glwe_in[ct] for B ciphertexts (k = 1, N = 2048, one decomposition level of base
 * 2^pbs_base_log <= 2^23), host buffers: ggsw_std [row 2][col 2][N] (standard domain, entities/ggsw_ciphertext.rs:185-197),
 * glwe_in / glwe_acc B x 2 x N.  timing_reps > 0: the kernel is run that many more times and *kernel_ms is its average
 * duration (HIP events).  0 = ok, 1 = error (tfhe_hip_last_error). */
int tfhe_hip_exact_external_product_batch(int device_id, const uint64_t *ggsw_std, uint32_t pbs_base_log, const uint64_t *glwe_in,
                                          uint64_t *glwe_acc, size_t B, int timing_reps, double *kernel_ms);

#ifdef __cplusplus
}
#endif
#endif /* TFHE_HIP_EXPERIMENTAL_H */
