/*
 * tfhe_hip.h -- C-ABI of the MI355X (gfx950) keyswitch + programmable-bootstrap engine.
 *
 * This is the drop-in boundary for ONE path of M-Bln/tfhe-rs-string (tfhe-rs 0.5.0 fork):
 * `shortint::ServerKey::apply_lookup_table_assign` -> `keyswitch_programmable_bootstrap_assign`
 * (tfhe/src/shortint/server_key/mod.rs:465-476, 783-857), i.e. the two core_crypto free functions it
 * bottoms out in, batched over independent ciphertexts:
 *     keyswitch_lwe_ciphertext                      (core_crypto/algorithms/lwe_keyswitch.rs:96-170)
 *     programmable_bootstrap_lwe_ciphertext_mem_optimized
 *                                                   (core_crypto/algorithms/lwe_programmable_bootstrapping.rs:1067-1111)
 *
 * Conventions mirror the reference's own cbindgen C API (tfhe/src/c_api):
 *   - every function returns int: 0 = ok, 1 = error (c_api/utils.rs:3-12 `catch_panic`); the
 *     reference panics (assert!) on dimension mismatch (lwe_keyswitch.rs:106-141), here the call
 *     fails with 1 and `tfhe_hip_last_error()` names the check;
 *   - out-params `T **` are set to NULL first (c_api/shortint/server_key/pbs.rs:24-29);
 *   - opaque handles with explicit *_destroy;
 *   - raw slices with explicit dimensions (c_api/core_crypto/mod.rs:37-50); all containers are
 *     the reference's flat layouts: LWE = mask then body (entities/lwe_ciphertext.rs:598-609),
 *     GLWE = k mask polynomials then the body polynomial (entities/glwe_ciphertext.rs:423-433),
 *     LweKeyswitchKey = [k*N][ks_level, level l first][n+1] (entities/lwe_keyswitch_key.rs:102-108),
 *     LweBootstrapKey (standard domain) = [n][pbs_level, level 1 first][k+1][k+1][N]
 *     (entities/lwe_bootstrap_key.rs:99-106, entities/ggsw_ciphertext.rs:185-197).
 *   - pointers must be non-null and 8-byte aligned (c_api/utils.rs:14-28).
 *
 * No torch / HIP types appear in any signature; `stream` is a `hipStream_t` passed as void*.
 * Thread safety: a server key may be used from many host threads (the reference shares
 * `ServerKey` immutably across rayon workers, shortint/engine/mod.rs:23-25): the host entry points
 * queue up on the key (one staging workspace; a batch fills the GPU anyway); the device entry points keep a workspace
 * per (key, stream) and may run concurrently on different streams, one thread per stream.
 */
#ifndef TFHE_HIP_H
#define TFHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tfhe_hip_context tfhe_hip_context;
typedef struct tfhe_hip_server_key tfhe_hip_server_key;

/* shortint/parameters/mod.rs:703-717 (ClassicPBSParameters) -- the fields the server side uses.  In brackets:
 * PARAM_MESSAGE_2_CARRY_2_KS_PBS, the set the wave / workgroup kernels are built for (with the other three 4-bit sets and
 * the PBS_KS / multi-bit 2_2 sets).  Accepted beyond it, on the generic kernel: glwe_dim 1 ... 5, poly_size a power of two
 * in 256 ... 4096 with (glwe_dim + 1) * poly_size <= 10240, any pbs decomposition with base_log <= 31 and level * base_log <= 63,
 * keyswitch decompositions (level, base_log) in {(5,3), (4,4), (3,5), (2,7), (2,5), (3,4), (4,3), (6,3)} -- i.e. all fifteen
 * classic KS_PBS sets of 1 ... 5 bits (shortint/parameters/mod.rs:599-820) and the string layer's PARAM_MESSAGE_2_CARRY_2_TEST. */
typedef struct tfhe_hip_params {
    uint32_t lwe_dim;         /* n: small LWE dimension (742) */
    uint32_t glwe_dim;        /* k (1) */
    uint32_t poly_size;       /* N (2048) */
    uint32_t pbs_base_log;    /* 23 */
    uint32_t pbs_level;       /* 1 */
    uint32_t ks_base_log;     /* 3 */
    uint32_t ks_level;        /* 5 */
    uint32_t message_modulus; /* 4 */
    uint32_t carry_modulus;   /* 4 */
    uint32_t pbs_order;       /* 0 = KeyswitchBootstrap (EncryptionKeyChoice::Big), 1 = BootstrapKeyswitch (Small) */
} tfhe_hip_params;

/* Thread-local message of the last failing call on this thread ("" if none). */
const char *tfhe_hip_last_error(void);

/* A context over n_devices GPUs of this process (no `backend` argument: there is no CPU backend).
 * device_ids[0] is the PRIMARY device: the device-buffer entry points take pointers and streams of that device.  A server
 * key created on a multi-device context is replicated on every device; every batch entry point then splits its B
 * ciphertexts contiguously over the devices (ceil(B / n) rows each; a device only takes part from 64 ciphertexts on,
 * TFHE_HIP_MULTI_MIN) -- the fan-out the reference gets from rayon's par_iter over independent blocks
 * (tfhe/src/integer/server_key/radix_parallel/comparison.rs:22-28) -- and the results come back into the caller's buffer
 * (host entry points: every device writes its rows of the host array; device entry points: peer copies over xGMI behind
 * events, nothing blocks the host).  The op-graph layer (tfhe_hip_strings.h) shards every level the same way because it
 * runs on these entry points.  n_devices = 1 is the plain single-GPU engine.  The one-process-per-GPU mode
 * (torch.distributed / RCCL, bench.py) uses one single-device context per rank instead.
 * UNMEASURED at n_devices > 1 on distinct GPUs (one-GPU test boxes); exercised with the same device listed twice. */
int tfhe_hip_context_create(tfhe_hip_context **out, const int *device_ids, int n_devices);
int tfhe_hip_context_destroy(tfhe_hip_context *ctx);
int tfhe_hip_context_device_count(const tfhe_hip_context *ctx, int *n_devices);
/* Name of the (primary) device ("gfx950...") and its CU count, for reports. */
int tfhe_hip_context_info(const tfhe_hip_context *ctx, char *arch_out, size_t arch_cap, int *cu_count);

/*
 * Replaces ServerKey construction on the server (shortint/engine/server_side.rs:54-...): uploads the
 * keyswitch key, uploads the standard-domain bootstrap key and converts it to the Fourier domain on
 * the GPU (par_convert_standard_lwe_bootstrap_key_to_fourier, lwe_bootstrap_key_conversion.rs:99-152).
 * Host pointers; the data is copied, the caller keeps ownership.
 */
int tfhe_hip_server_key_create(tfhe_hip_context *ctx, tfhe_hip_server_key **out, const tfhe_hip_params *params,
                               const uint64_t *bsk_std, size_t bsk_len, const uint64_t *ksk, size_t ksk_len);
/*
 * Multi-bit PBS server key (shortint MultiBitPBSParameters, shortint/parameters/multi_bit.rs; key layout
 * entities/lwe_multi_bit_bootstrap_key.rs: [lwe_dim / g groups][2^g GGSWs][level][row][col][N]).
 * grouping_factor g = 2 (pbs base 2^22, ks 5x3: PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS) or
 * 3 (2^21, ks 7x2: ..._GROUP_3_KS_PBS).  The batch entry points below then run keyswitch +
 * multi_bit_blind_rotate_assign + sample extraction
 * (core_crypto/algorithms/lwe_multi_bit_programmable_bootstrapping.rs:295-545, 1035-1127), the groups in
 * key order (the reference's `deterministic` execution order).
 */
int tfhe_hip_server_key_create_multi_bit(tfhe_hip_context *ctx, tfhe_hip_server_key **out,
                                         const tfhe_hip_params *params, uint32_t grouping_factor,
                                         const uint64_t *bsk_std, size_t bsk_len, const uint64_t *ksk,
                                         size_t ksk_len);
/*
 * Seeded (compressed) keys -- shortint::CompressedServerKey (tfhe/src/shortint/server_key/compressed.rs:43-55): only the
 * bodies are stored, the masks are the output of the reference's CSPRNG (concrete-csprng: AES-128-CTR keyed with the
 * 128-bit compression seed) and are generated ON THE DEVICE (csprng.hip), i.e. this replaces
 * decompress_seeded_lwe_bootstrap_key / decompress_seeded_lwe_keyswitch_key
 * (core_crypto/algorithms/seeded_lwe_bootstrap_key_decompression.rs:37-60, seeded_lwe_keyswitch_key_decompression.rs)
 * followed by tfhe_hip_server_key_create.  bsk_bodies: [ggsw][level][row][N] body polynomials
 * (entities/seeded_ggsw_ciphertext_list.rs), ksk_bodies: [k*N][ks_level] bodies (entities/seeded_lwe_keyswitch_key.rs);
 * seeds: the 16 native (little-endian) bytes of Seed(u128).  grouping_factor 0 = classic PBS, 2 / 3 = multi-bit.
 */
int tfhe_hip_server_key_create_seeded(tfhe_hip_context *ctx, tfhe_hip_server_key **out, const tfhe_hip_params *params,
                                      uint32_t grouping_factor, const uint64_t *bsk_bodies, size_t bsk_bodies_len,
                                      const uint8_t *bsk_seed, const uint64_t *ksk_bodies, size_t ksk_bodies_len,
                                      const uint8_t *ksk_seed);
/* introspection: u64 outputs [first_word, first_word + count) of a generator created from `seed`, computed by the
 * device kernel (parity tests against the oracle) */
int tfhe_hip_csprng_fill(tfhe_hip_context *ctx, const uint8_t *seed, uint64_t first_word, uint64_t *out, size_t count);
/*
 * Wire format (wire.cpp): `bincode::serialize` images (bincode 1.3 defaults) of shortint::CompressedServerKey and
 * shortint::Ciphertext as a tfhe-rs 0.5 client writes them.  PARITY UNPINNED (no serialized fixture in the reference):
 * the layout is restated from the serde derives cited in wire.cpp.
 *   _info: parameters found in a serialized CompressedServerKey (no device work);
 *   _create_from_compressed: parse + tfhe_hip_server_key_create_seeded;
 *   _serialize: the writers (out == NULL: *written = size needed).
 */
int tfhe_hip_compressed_server_key_info(const uint8_t *bytes, size_t len, tfhe_hip_params *params, uint32_t *grouping_factor,
                                        uint64_t *max_degree, size_t *consumed);
int tfhe_hip_server_key_create_from_compressed(tfhe_hip_context *ctx, tfhe_hip_server_key **out, const uint8_t *bytes,
                                               size_t len, tfhe_hip_params *params_out, uint32_t *grouping_factor_out);
int tfhe_hip_compressed_server_key_serialize(const tfhe_hip_params *params, uint32_t grouping_factor, const uint64_t *bsk_bodies,
                                             size_t bsk_bodies_len, const uint8_t *bsk_seed, const uint64_t *ksk_bodies,
                                             size_t ksk_bodies_len, const uint8_t *ksk_seed, uint8_t *out, size_t cap,
                                             size_t *written);
int tfhe_hip_shortint_ciphertext_serialize(const uint64_t *lwe, size_t lwe_len, uint64_t degree, uint64_t noise_level,
                                           uint32_t message_modulus, uint32_t carry_modulus, uint32_t pbs_order, uint8_t *out,
                                           size_t cap, size_t *written);
int tfhe_hip_shortint_ciphertext_deserialize(const uint8_t *bytes, size_t len, uint64_t *lwe_out, size_t lwe_cap, size_t *lwe_len,
                                             uint64_t *degree, uint64_t *noise_level, uint32_t *message_modulus,
                                             uint32_t *carry_modulus, uint32_t *pbs_order, size_t *consumed);
/* The STANDARD-domain (uncompressed) keys, `bincode::serialize(&LweKeyswitchKey<Vec<u64>>)` / `(&LweBootstrapKey<Vec<u64>>)`
 * (core_crypto/entities/lwe_keyswitch_key.rs:76-108, lwe_bootstrap_key.rs:98-106 -> ggsw_ciphertext_list.rs:9-20): plain serde
 * derives, a Vec<u64> in the container orders of the top of this file plus the decomposition metadata.  *_deserialize with
 * out == NULL reports the length (two-call pattern); _create_from_standard_keys = both readers + tfhe_hip_server_key_create
 * (message_modulus / carry_modulus / pbs_order are ServerKey fields the two keys do not carry).  PARITY UNPINNED, like the rest
 * of the wire format (no serialized fixture in the reference). */
int tfhe_hip_lwe_keyswitch_key_deserialize(const uint8_t *bytes, size_t len, uint64_t *ksk_out, size_t cap, size_t *ksk_len,
                                           uint32_t *decomp_base_log, uint32_t *decomp_level_count, uint32_t *input_lwe_dim,
                                           uint32_t *output_lwe_dim, size_t *consumed);
int tfhe_hip_lwe_bootstrap_key_deserialize(const uint8_t *bytes, size_t len, uint64_t *bsk_out, size_t cap, size_t *bsk_len,
                                           uint32_t *glwe_dim, uint32_t *poly_size, uint32_t *decomp_base_log,
                                           uint32_t *decomp_level_count, uint32_t *input_lwe_dim, size_t *consumed);
int tfhe_hip_lwe_keyswitch_key_serialize(const uint64_t *ksk, size_t ksk_len, uint32_t decomp_base_log, uint32_t decomp_level_count,
                                         uint32_t output_lwe_dim, uint8_t *out, size_t cap, size_t *written);
int tfhe_hip_lwe_bootstrap_key_serialize(const uint64_t *bsk, size_t bsk_len, uint32_t glwe_dim, uint32_t poly_size,
                                         uint32_t decomp_base_log, uint32_t decomp_level_count, uint8_t *out, size_t cap,
                                         size_t *written);
int tfhe_hip_server_key_create_from_standard_keys(tfhe_hip_context *ctx, tfhe_hip_server_key **out, const uint8_t *ksk_bytes,
                                                  size_t ksk_bytes_len, const uint8_t *bsk_bytes, size_t bsk_bytes_len,
                                                  uint32_t message_modulus, uint32_t carry_modulus, uint32_t pbs_order,
                                                  tfhe_hip_params *params_out);
/* shortint::CompressedCiphertext (shortint/ciphertext/mod.rs:470-478): a seeded LWE ciphertext = its body and its own seed.
 * _decompress_seeded_batch = decompress_seeded_lwe_ciphertext (core_crypto/algorithms/seeded_lwe_ciphertext_decompression.rs:8-40)
 * for B ciphertexts on the device: bodies [B], seeds [B][16] -> lwe_out [B][lwe_dim + 1]. */
int tfhe_hip_lwe_decompress_seeded_batch(tfhe_hip_context *ctx, const uint64_t *bodies, const uint8_t *seeds, uint32_t lwe_dim,
                                         uint64_t *lwe_out, size_t B);
int tfhe_hip_shortint_compressed_ciphertext_serialize(uint64_t body, uint64_t lwe_size, const uint8_t *seed, uint64_t degree,
                                                      uint64_t noise_level, uint32_t message_modulus, uint32_t carry_modulus,
                                                      uint32_t pbs_order, uint8_t *out, size_t cap, size_t *written);
int tfhe_hip_shortint_compressed_ciphertext_deserialize(const uint8_t *bytes, size_t len, uint64_t *body, uint64_t *lwe_size,
                                                        uint8_t *seed, uint64_t *degree, uint64_t *noise_level,
                                                        uint32_t *message_modulus, uint32_t *carry_modulus, uint32_t *pbs_order,
                                                        size_t *consumed);
/* The batch sizes the kernels work in: *latency_round ciphertexts (one per CU) cost one pass of the latency kernel,
 * *throughput_round (four per CU; 0 if the key has no throughput kernel) one pass of the throughput kernel.  Level
 * schedulers (host_ops.hpp) fill their launches to these boundaries. */
/* the parameters the key was created with (wrappers check theirs against them) */
int tfhe_hip_server_key_get_params(const tfhe_hip_server_key *sk, tfhe_hip_params *out);
int tfhe_hip_server_key_batch_rounds(const tfhe_hip_server_key *sk, size_t *latency_round, size_t *throughput_round);
int tfhe_hip_server_key_destroy(tfhe_hip_server_key *sk);

/* ServerKey::generate_lookup_table / fill_accumulator (shortint/server_key/mod.rs:383-399,
 * shortint/engine/mod.rs:72-128).  table[i] = f(i), i < message_modulus*carry_modulus.
 * Host-side integer code; glwe_out has (k+1)*N elements; *max_out = max f (the LUT degree). */
int tfhe_hip_lut_generate(const tfhe_hip_params *params, const uint64_t *table, uint64_t *glwe_out, uint64_t *max_out);

/* Page-locked host memory for the buffers handed to the host-buffer entry points below.  Optional: those accept
 * any host memory (small calls are staged through an internal pinned buffer).  With buffers from here the staging
 * copy disappears and batches of several thousand ciphertexts are pipelined: chunks are uploaded, bootstrapped
 * and downloaded on alternating streams, so the PCIe transfers overlap the kernels. */
int tfhe_hip_host_alloc(void **out, size_t bytes);
int tfhe_hip_host_free(void *p);

/* ---- host-buffer entry points (copy in, compute, copy out, synchronous) -------------------- */

/* keyswitch_lwe_ciphertext for B ciphertexts: lwe_in B x (kN+1), lwe_out B x (n+1).  Computed as an exact int8 matrix
 * product on the matrix cores (digits x byte planes of the key; keyswitch_mfma.hip); the environment variable
 * TFHE_HIP_KS_MFMA_MIN, read at key creation, sets the batch size from which that kernel is used (0: never -- the
 * dot4 vector kernels; same bits either way). */
int tfhe_hip_keyswitch_batch(tfhe_hip_server_key *sk, const uint64_t *lwe_in, uint64_t *lwe_out, size_t B);
/* programmable_bootstrap_lwe_ciphertext_mem_optimized for B ciphertexts: lwe_in B x (n+1),
 * luts n_luts x (k+1)N, lut_idx[B] selects the accumulator of each ciphertext, lwe_out B x (kN+1). */
int tfhe_hip_pbs_batch(tfhe_hip_server_key *sk, const uint64_t *lwe_in, const uint64_t *luts, size_t n_luts,
                       const uint32_t *lut_idx, uint64_t *lwe_out, size_t B);
/* keyswitch_programmable_bootstrap_assign (crypto part) for B ciphertexts: B x (kN+1) in and out. */
int tfhe_hip_ks_pbs_batch(tfhe_hip_server_key *sk, const uint64_t *lwe_in, const uint64_t *luts, size_t n_luts,
                          const uint32_t *lut_idx, uint64_t *lwe_out, size_t B);

/* programmable_bootstrap_keyswitch_assign (crypto part; shortint/server_key/mod.rs:859-932, the order of
 * the *_PBS_KS parameter sets): PBS then keyswitch, B x (n+1) in and out. */
int tfhe_hip_pbs_ks_batch(tfhe_hip_server_key *sk, const uint64_t *lwe_in, const uint64_t *luts, size_t n_luts,
                          const uint32_t *lut_idx, uint64_t *lwe_out, size_t B);
/* ServerKey::apply_lookup_table_assign (shortint/server_key/mod.rs:465-476): dispatches on the
 * key's pbs_order to tfhe_hip_ks_pbs_batch or tfhe_hip_pbs_ks_batch. */
int tfhe_hip_apply_lookup_table_batch(tfhe_hip_server_key *sk, const uint64_t *lwe_in, const uint64_t *luts,
                                      size_t n_luts, const uint32_t *lut_idx, uint64_t *lwe_out, size_t B);

/* ---- device-buffer entry points (pointers are device memory of the context's GPU; the work is
 *      enqueued on `stream` (hipStream_t as void*, NULL = default stream) and NOT synchronised) --- */
int tfhe_hip_keyswitch_batch_device(tfhe_hip_server_key *sk, const uint64_t *d_lwe_in, uint64_t *d_lwe_out, size_t B,
                                    void *stream);
int tfhe_hip_pbs_batch_device(tfhe_hip_server_key *sk, const uint64_t *d_lwe_in, const uint64_t *d_luts,
                              size_t n_luts, const uint32_t *d_lut_idx, uint64_t *d_lwe_out, size_t B, void *stream);
int tfhe_hip_ks_pbs_batch_device(tfhe_hip_server_key *sk, const uint64_t *d_lwe_in, const uint64_t *d_luts,
                                 size_t n_luts, const uint32_t *d_lut_idx, uint64_t *d_lwe_out, size_t B,
                                 void *stream);

int tfhe_hip_pbs_ks_batch_device(tfhe_hip_server_key *sk, const uint64_t *d_lwe_in, const uint64_t *d_luts,
                                 size_t n_luts, const uint32_t *d_lut_idx, uint64_t *d_lwe_out, size_t B,
                                 void *stream);
int tfhe_hip_apply_lookup_table_batch_device(tfhe_hip_server_key *sk, const uint64_t *d_lwe_in,
                                             const uint64_t *d_luts, size_t n_luts, const uint32_t *d_lut_idx,
                                             uint64_t *d_lwe_out, size_t B, void *stream);

/* ---- device-resident buffers for level schedulers -------------------------------------------------------------
 * An op-graph runs many batched launches in a row whose inputs are leveled combinations of earlier outputs.  With
 * these calls the ciphertexts stay in HBM between the levels: operands are uploaded once, every level's input rows are
 * formed on the device (tfhe_hip_device_lincomb_rows: the reference's lwe_ciphertext_add / cleartext_mul /
 * plaintext_add chains, core_crypto/algorithms/lwe_linear_algebra.rs, evaluated lazily) and fed to the *_device entry
 * points with stream NULL, and only the final results are downloaded.  All of them run on the default stream in call
 * order.  lincomb_rows: d_dst[r] = sum_{t = row_ptr[r] .. row_ptr[r+1]-1} coef[t] * src[t][0 .. row_words) + (0, .., 0, body[r]);
 * row_ptr (rows + 1), coef, src (DEVICE row addresses) and body are host arrays. */
int tfhe_hip_device_alloc(tfhe_hip_server_key *sk, uint64_t **out, size_t words);
int tfhe_hip_device_free(uint64_t *p);
int tfhe_hip_device_upload(tfhe_hip_server_key *sk, uint64_t *d_dst, const void *h_src, size_t bytes);
int tfhe_hip_device_download(tfhe_hip_server_key *sk, void *h_dst, const uint64_t *d_src, size_t bytes);
/* upload: a page-locked source (tfhe_hip_host_alloc) is read asynchronously and must stay unchanged until a download or
 * tfhe_hip_device_synchronize; a pageable one is staged before the call returns.  download waits for the data. */
int tfhe_hip_device_synchronize(tfhe_hip_server_key *sk);
/* upload on the key's TRANSFER stream, for pipelines that feed a level in pieces: the copy runs beside the kernels that
 * are already enqueued (they must not touch the destination), after the earlier overlapped uploads, and -- with
 * wait_prior != 0 -- after everything enqueued on the default stream so far; whatever is enqueued on the default stream
 * after the call waits for it.  The source must be page-locked (tfhe_hip_host_alloc) and stay unchanged until a
 * download or tfhe_hip_device_synchronize. */
int tfhe_hip_device_upload_overlapped(tfhe_hip_server_key *sk, uint64_t *d_dst, const void *h_src, size_t bytes, int wait_prior);
int tfhe_hip_device_lincomb_rows(tfhe_hip_server_key *sk, uint64_t *d_dst, size_t rows, size_t row_words,
                                 const uint32_t *row_ptr, const uint64_t *coef, const uint64_t *const *src,
                                 const uint64_t *body);

/* ---- measurement hooks ----------------------------------------------------------------------
 * With profiling on, every *_device / host call brackets its kernels with HIP events on the
 * launch stream; the accumulated durations are read back (after synchronising) here.
 * kernel: 0 = keyswitch, 1 = blind rotation (+ fused sample extract). */
int tfhe_hip_profiling_enable(tfhe_hip_server_key *sk, int on);
int tfhe_hip_profiling_read(tfhe_hip_server_key *sk, int kernel, double *total_ms, uint64_t *launches, int reset);

/* ---- failure reporting of the kernels themselves -------------------------------------------------
 * The reference's path cannot fail half-way (a panic unwinds the call).  A GPU launch can: a kernel that detects that it
 * cannot produce a valid result (bit 0: a wave pair of the throughput kernel timed out waiting for its partner; bit 1: a
 * lut_idx >= n_luts reached a device-buffer entry point, whose arguments the host cannot inspect) sets a bit in a
 * device-visible word of the key and runs to its end.  Every host-buffer call checks the word after its synchronisation
 * and returns 1 instead of handing out the data; the word is STICKY: every later call on the key (device-buffer entry
 * points included, before they enqueue; tfhe_hip_device_synchronize / _download after they wait) returns 1 until it
 * has been read and cleared here.  Callers of the asynchronous *_device entry points synchronise their stream and then
 * call this with clear = 0 to learn whether the results may be used.  *code: 0 = ok. */
int tfhe_hip_server_key_device_status(tfhe_hip_server_key *sk, uint32_t *code, int clear);
/* Per-stream scratch.  A device-buffer call keeps a workspace (small LWEs, keyswitch partial sums: ~20 MB at B = 1024) per
 * (key, caller stream), the counterpart of the reference's thread-local buffers (shortint/engine/mod.rs:23-25).  At most
 * TFHE_HIP_MAX_STREAM_WS (default 8) caller streams keep one at a time -- the least recently used is released when a ninth
 * stream shows up -- and a host that creates a stream per request can hand a stream's workspace back itself before it
 * destroys the stream.  *workspaces_left (optional): workspaces the key still holds, its own streams included. */
int tfhe_hip_server_key_release_stream(tfhe_hip_server_key *sk, void *stream, size_t *workspaces_left);
/* DIAGNOSTICS (tests only): from CMUX step `step` on, one wave of ciphertext 0 of every throughput-kernel launch
 * withholds its hand-off signals, so its partner must run into the time-out above; -1 switches it off. */
int tfhe_hip_debug_skip_pair_signal(tfhe_hip_server_key *sk, int step);

/* ---- introspection used by the parity tests ----------------------------------------------- */
/* Downloads the Fourier-domain bootstrap key in the reference container order
 * [n][level][row][col][N/2] as separate re / im planes (DIF position order). */
int tfhe_hip_server_key_fourier_bsk(tfhe_hip_server_key *sk, double *re_out, double *im_out, size_t len);
/* Selects the blind-rotation kernel variant: -1 = automatic (default when the parameters allow it): up to one
 * ciphertext per CU variant 3, up to two per CU variant 5, up to three per CU variant 5 on two CU-loads and variant 3 on
 * the third, beyond that whole rounds of variant 2 (four per CU) and the remainder by the same rule;
 * 5 = four wavefronts per ciphertext, 8 points per thread, two ciphertexts per CU, 4 = 256-thread workgroup per
 * ciphertext (each thread both polynomials), two per CU, 3 = 512-thread workgroup per ciphertext (latency),
 * 2 = wave pair per ciphertext (throughput), 1 = wave per ciphertext, 0 = generic LDS kernel (any PBS level count,
 * N = 256 ... 4096).  All variants produce identical bits; DESIGN.md section 6. */
int tfhe_hip_set_variant(tfhe_hip_server_key *sk, int variant);

#ifdef __cplusplus
}
#endif
#endif /* TFHE_HIP_H */
