// kernels.hpp -- launch wrappers of the HIP kernels (implemented in kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace tfhe_hip {

struct DevParams {
    int n;        // small LWE dimension
    int k;        // glwe dimension (1)
    int N;        // polynomial size (2048)
    int log2N;
    int pbs_base_log, pbs_level;
    int ks_base_log, ks_level;
};

// per-launch control block of the blind-rotation kernels (passed by value as a kernel argument)
struct BrCtl {
    uint32_t *err = nullptr;   // the key's device-visible error word (device_math.hpp: ERR_*); null = do not report
    uint32_t n_luts = 1;       // lut_idx[ct] must be < n_luts
    int skip_signal_step = -1; // DIAGNOSTICS (tfhe_hip_debug_skip_pair_signal): the body wave of ciphertext 0 withholds
                               // its hand-off signals from this CMUX step on, so its partner must time out
};

constexpr int KS_BT = 8;          // ciphertexts per keyswitch workgroup
constexpr int KS_COLS = 256;      // output columns per keyswitch workgroup (= threads)
constexpr int KS_MAX_SPLITS = 16; // split of the 2048 input coefficients across workgroups

// number of j-splits the keyswitch uses for a batch of B (also sizes the partial-sum workspace)
int ks_splits_for(const DevParams &p, size_t B, int cu_count);

// colsum[o] = 2^(ks_base_log-1) * sum_{j,l} ksk[j][l][o]   (one-off, at key upload)
hipError_t launch_ksk_colsum(const DevParams &p, const uint64_t *d_ksk, uint64_t *d_colsum, hipStream_t s);
// byte-plane re-layout of the KSK for the dot4 keyswitch (same size as the KSK, one-off)
hipError_t launch_ksk_to_planes(const DevParams &p, const uint64_t *d_ksk, uint32_t *d_planes, hipStream_t s);
// partial[split][ct][o] = sum over the split's j and all levels of (digit + half) * ksk[j][l][o]
hipError_t launch_keyswitch(const DevParams &p, const uint64_t *d_lwe_in, const uint32_t *d_ksk_planes,
                            uint64_t *d_partial, size_t B, int splits, hipStream_t s);
// out[ct][o] = (o == n ? body_in : 0) + colsum[o] - sum_split partial[split][ct][o]
hipError_t launch_keyswitch_finalize(const DevParams &p, const uint64_t *d_lwe_in, const uint64_t *d_colsum,
                                     const uint64_t *d_partial, uint64_t *d_lwe_out, size_t B, int splits,
                                     hipStream_t s);
// ---- the keyswitch on the matrix cores (keyswitch_mfma.hip): exact int8 x int8 -> int32 products of the signed digits with
// the byte planes of the key; same results as the dot4 kernels above
bool keyswitch_mfma_supported(const DevParams &p);
size_t keyswitch_mfma_key_bytes(const DevParams &p);            // the key in MFMA fragment order
size_t keyswitch_mfma_digit_bytes(const DevParams &p, size_t B); // the digits of a batch in fragment order
int keyswitch_mfma_splits_for(const DevParams &p, size_t B, int cu_count);
hipError_t launch_ksk_to_mfma_tiles(const DevParams &p, const uint64_t *d_ksk, void *d_tiles, hipStream_t s);
// d_sums: ceil(B / 32) * 32 ints; d_partial: splits x B x (n + 1) words
hipError_t launch_keyswitch_mfma(const DevParams &p, const uint64_t *d_lwe_in, const void *d_tiles, void *d_digits, int *d_sums,
                                 uint64_t *d_partial, uint64_t *d_lwe_out, size_t B, int splits, hipStream_t s);
// standard BSK polynomial -> Fourier (forward_as_torus), one workgroup per polynomial;
// output in DIF position order, interleaved (re, im)
hipError_t launch_bsk_to_fourier(const DevParams &p, const uint64_t *d_bsk_std, double *d_bsk_f, size_t n_polys,
                                 const double *d_twist, const double *d_w, hipStream_t s);
// ---- seeded keys (csprng.hip): the reference's AES-128-CTR generator and seeded_*_decompression on the device ----
void aes128_round_keys(const uint8_t key[16], uint32_t rk[44]);
hipError_t upload_aes_table(uint32_t **d_te0);
// generator words [first_word, first_word + count) scattered as units: dst[(w / mask_words) * unit_words + w % mask_words]
hipError_t launch_csprng_fill(const uint32_t *d_te0, const uint8_t seed[16], uint64_t first_word, uint64_t count, uint64_t mask_words,
                              uint64_t unit_words, uint64_t *d_dst, hipStream_t s);
// `units` ciphertexts: mask_words generator words, then body_words words of d_bodies
hipError_t launch_seeded_expand(const uint32_t *d_te0, const uint8_t seed[16], const uint64_t *d_bodies, uint64_t units, uint64_t mask_words,
                                uint64_t body_words, uint64_t *d_dst, hipStream_t s);
// one seed per ciphertext (compressed ciphertexts): d_rks = [units][44] round-key words, body_words = 1
hipError_t launch_seeded_expand_units(const uint32_t *d_te0, const uint32_t *d_rks, const uint64_t *d_bodies, uint64_t units,
                                      uint64_t mask_words, uint64_t *d_dst, hipStream_t s);
// rows of lazy linear combinations, evaluated on the device (CSR over the terms; src: device row addresses)
hipError_t launch_lincomb_rows(uint64_t *d_dst, size_t rows, int row_words, const uint32_t *d_row_ptr, const uint64_t *d_coef,
                               const uint64_t *const *d_src, const uint64_t *d_body, hipStream_t s);
// blind rotation + sample extraction, variant 0: one 256-thread workgroup per ciphertext, LDS FFT; k = 1, any power-of-two
// polynomial size 256 .. 4096, any level count
bool blind_rotate_v0_supported(const DevParams &p);
hipError_t launch_blind_rotate_v0(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts,
                                  const uint32_t *d_lut_idx, const double *d_bsk_f, const double *d_twist,
                                  const double *d_w, uint64_t *d_lwe_out, size_t B, const BrCtl &ctl, hipStream_t s);

} // namespace tfhe_hip

namespace tfhe_hip {
// ---- variant 1: wave-per-ciphertext blind rotation (blind_rotate_wave.hip) --------------------
bool blind_rotate_wave_supported(const DevParams &p);
// re-orders the Fourier BSK (DIF position order) into the lane-ordered stream the wave kernel reads
hipError_t launch_bsk_permute_wave(const DevParams &p, const double *d_bsk_f, double *d_bsk_w, hipStream_t s);
hipError_t launch_blind_rotate_wave(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts,
                                    const uint32_t *d_lut_idx, const double *d_bsk_w, const char *d_gtab,
                                    const double *wconsts /* host: W[64], W[128], W[192], W[384] */,
                                    uint64_t *d_lwe_out, size_t B, const BrCtl &ctl, hipStream_t s);
} // namespace tfhe_hip

namespace tfhe_hip {
// ---- variant 2: wave-pair blind rotation (blind_rotate_pair.hip): 2 waves per ciphertext -------
hipError_t launch_bsk_permute_pair(const DevParams &p, const double *d_bsk_f, double *d_bsk_p, hipStream_t s);
// the oracle-order Fourier key reconstructed (exactly) from the pair-order copy
hipError_t launch_bsk_unpermute_pair(const DevParams &p, const double *d_bsk_p, double *d_bsk_f, hipStream_t s);
hipError_t launch_blind_rotate_pair(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts,
                                    const uint32_t *d_lut_idx, const double *d_bsk_p, const char *d_gtab,
                                    const double *wconsts, uint64_t *d_lwe_out, size_t B, const BrCtl &ctl, hipStream_t s);
} // namespace tfhe_hip

namespace tfhe_hip {
// ---- variant 3: workgroup-per-ciphertext blind rotation (blind_rotate_block.hip): the latency kernel
hipError_t launch_bsk_permute_block(const DevParams &p, const double *d_bsk_f, double *d_bsk_b, hipStream_t s);
hipError_t launch_blind_rotate_block(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts,
                                     const uint32_t *d_lut_idx, const double *d_bsk_b, const double *d_twist,
                                     const double *d_w, uint64_t *d_lwe_out, size_t B, const BrCtl &ctl, hipStream_t s);
// variant 4: 256-thread workgroups (each thread does both polynomials of its positions), 64 KiB of LDS: two per CU
hipError_t launch_blind_rotate_half(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts,
                                    const uint32_t *d_lut_idx, const double *d_bsk_b, const double *d_twist,
                                    const double *d_w, uint64_t *d_lwe_out, size_t B, const BrCtl &ctl, hipStream_t s);
} // namespace tfhe_hip

namespace tfhe_hip {
// ---- variant 5: four wavefronts per ciphertext, 8 points per thread, one LDS transposition per transform
// (blind_rotate_quad.hip); 256-thread workgroups, two per CU
hipError_t launch_bsk_permute_quad(const DevParams &p, const double *d_bsk_f, double *d_bsk_q, hipStream_t s);
hipError_t launch_blind_rotate_quad(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts,
                                    const uint32_t *d_lut_idx, const double *d_bsk_q, const double *d_twist,
                                    const double *d_w, uint64_t *d_lwe_out, size_t B, const BrCtl &ctl, hipStream_t s);
} // namespace tfhe_hip

namespace tfhe_hip {
// ---- multi-bit blind rotation (blind_rotate_block_mb.hip): grouping factor 2 (pbs base 2^22) or 3 (2^21)
bool blind_rotate_multi_bit_supported(const DevParams &p, int grouping_factor);
// d_bsk_f: Fourier GGSWs in the reference order [n/g groups][2^g][row][col][N/2]
hipError_t launch_bsk_permute_block_mb(const DevParams &p, int grouping_factor, const double *d_bsk_f, double *d_bsk_mb,
                                       hipStream_t s);
hipError_t launch_blind_rotate_block_mb(const DevParams &p, int grouping_factor, const uint64_t *d_lwe_small,
                                        const uint64_t *d_luts, const uint32_t *d_lut_idx, const double *d_bsk_mb,
                                        const double *d_twist, const double *d_w, uint64_t *d_lwe_out, size_t B,
                                        const BrCtl &ctl, hipStream_t s);
} // namespace tfhe_hip
