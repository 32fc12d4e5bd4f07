// wire.cpp -- bincode images of the reference's serde types on the keyswitch + bootstrap path, so that keys and
// ciphertexts written by a tfhe-rs 0.5 client (`bincode::serialize`, bincode 1.3 with its default options:
// little-endian, fixed-width integers, usize and sequence lengths as u64, enum variants as u32, u128 as 16 bytes)
// can be loaded:
//   shortint::CompressedServerKey             tfhe/src/shortint/server_key/compressed.rs:10-17, 43-55
//     SeededLweKeyswitchKey<Vec<u64>>          tfhe/src/core_crypto/entities/seeded_lwe_keyswitch_key.rs:10-21
//     SeededLweBootstrapKey / SeededGgswCiphertextList   entities/seeded_lwe_bootstrap_key.rs:15-24, seeded_ggsw_ciphertext_list.rs:11-23
//     SeededLweMultiBitBootstrapKey            entities/seeded_lwe_multi_bit_bootstrap_key.rs:16-25
//     CompressionSeed { seed: Seed(u128) }     core_crypto/commons/math/random/generator.rs:25-39
//     CiphertextModulus -> { modulus: u128 (0 = native), scalar_bits: usize }   core_crypto/commons/ciphertext_modulus.rs:41-64
//   LweKeyswitchKey<Vec<u64>> (standard, uncompressed)   tfhe/src/core_crypto/entities/lwe_keyswitch_key.rs:76-108
//   LweBootstrapKey<Vec<u64>> = { ggsw_list: GgswCiphertextList }   entities/lwe_bootstrap_key.rs:98-106, ggsw_ciphertext_list.rs:9-20
//     (the STANDARD-domain keys: plain serde derives over Vec<u64> + metadata.  The Fourier key inside a serialized
//      shortint::ServerKey is in concrete-fft's private order and is not read -- a server loads these two instead.)
//   shortint::Ciphertext                       tfhe/src/shortint/ciphertext/mod.rs:261-270 (LweCiphertext: entities/lwe_ciphertext.rs:500-507)
// The field order is the declaration order of the derives.  PARITY UNPINNED: the reference holds no serialized
// fixture and cannot be built here; the layout is restated from the sources above and from bincode's specification.
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tfhe_hip.h"

#define TFHE_HIP_API extern "C" __attribute__((visibility("default")))
extern "C" __attribute__((visibility("hidden"))) int tfhe_hip_set_error(const char *msg);

namespace {

struct Reader {
    const uint8_t *p;
    size_t len, pos = 0;
    bool ok = true;
    std::string err;
    bool fail(const std::string &m)
    {
        if (ok) err = m + " at byte " + std::to_string(pos);
        ok = false;
        return false;
    }
    bool take(void *dst, size_t n, const char *what)
    {
        if (!ok) return false;
        if (len - pos < n) return fail(std::string("truncated input while reading ") + what);
        std::memcpy(dst, p + pos, n);
        pos += n;
        return true;
    }
    uint64_t u64(const char *what)
    {
        uint64_t v = 0;
        take(&v, 8, what);
        return v;
    }
    uint32_t u32(const char *what)
    {
        uint32_t v = 0;
        take(&v, 4, what);
        return v;
    }
    // Vec<u64>: length then elements; `limit` guards against absurd lengths in corrupt input
    bool vec_u64(std::vector<uint64_t> &out, const char *what)
    {
        const uint64_t n = u64(what);
        if (!ok) return false;
        if (n > (len - pos) / 8) return fail(std::string("sequence length exceeds the input while reading ") + what);
        out.resize((size_t)n);
        return take(out.data(), (size_t)n * 8, what);
    }
    // CiphertextModulus<u64>: only the native modulus 2^64 is supported by the kernels
    bool native_modulus(const char *what)
    {
        uint8_t m[16];
        take(m, 16, what);
        const uint64_t bits = u64(what);
        if (!ok) return false;
        for (uint8_t b : m)
            if (b) return fail(std::string(what) + ": custom ciphertext moduli are not supported (native 2^64 only)");
        if (bits != 64) return fail(std::string(what) + ": expected 64-bit scalars");
        return true;
    }
};

struct Writer {
    std::vector<uint8_t> b;
    void raw(const void *src, size_t n)
    {
        const uint8_t *s = (const uint8_t *)src;
        b.insert(b.end(), s, s + n);
    }
    void u64(uint64_t v) { raw(&v, 8); }
    void u32(uint32_t v) { raw(&v, 4); }
    void vec_u64(const uint64_t *v, size_t n)
    {
        u64(n);
        raw(v, n * 8);
    }
    void native_modulus()
    {
        const uint8_t zero[16] = {0};
        raw(zero, 16);
        u64(64);
    }
};

struct CompressedKey {
    tfhe_hip_params params{};
    uint32_t grouping_factor = 0;
    uint64_t max_degree = 0;
    std::vector<uint64_t> ksk_bodies, bsk_bodies;
    uint8_t ksk_seed[16], bsk_seed[16];
};

constexpr uint64_t kMaxDim = 0xFFFFFFFFull; // every dimension is handed on as a uint32

// The header fields of every key are untrusted u64s: each one is bounded BEFORE it is multiplied with another (a wrapped
// product would be a division by zero further down) or narrowed to the uint32 the parameters carry.
bool decomposition_ok(uint64_t base_log, uint64_t level) { return base_log >= 1 && base_log <= 64 && level >= 1 && level <= 64 && base_log * level <= 64; }
bool poly_size_ok(uint64_t n) { return n >= 1 && n <= (1u << 20) && (n & (n - 1)) == 0; }

bool read_compressed_key(Reader &r, CompressedKey &k, bool want_data)
{
    tfhe_hip_params &p = k.params;
    // key_switching_key: SeededLweKeyswitchKey { data, decomp_base_log, decomp_level_count, output_lwe_size, compression_seed, ciphertext_modulus }
    const size_t ksk_at = r.pos;
    if (!r.vec_u64(k.ksk_bodies, "keyswitch key bodies")) return false;
    const uint64_t ks_base_log = r.u64("keyswitch decomp_base_log"), ks_level = r.u64("keyswitch decomp_level_count");
    const uint64_t out_size = r.u64("keyswitch output_lwe_size");
    r.take(k.ksk_seed, 16, "keyswitch compression seed");
    if (!r.native_modulus("keyswitch key")) return false;
    if (out_size < 2 || out_size > kMaxDim || !decomposition_ok(ks_base_log, ks_level) || k.ksk_bodies.size() % ks_level ||
        k.ksk_bodies.size() / ks_level > kMaxDim) {
        r.pos = ksk_at;
        return r.fail("inconsistent seeded keyswitch key header");
    }
    p.ks_base_log = (uint32_t)ks_base_log;
    p.ks_level = (uint32_t)ks_level;
    p.lwe_dim = (uint32_t)(out_size - 1);
    const size_t in_dim = k.ksk_bodies.size() / p.ks_level;
    // bootstrapping_key: enum ShortintCompressedBootstrappingKey { Classic(..), MultiBit { seeded_bsk, deterministic_execution } }
    const uint32_t variant = r.u32("bootstrapping key variant");
    if (r.ok && variant > 1) return r.fail("unknown ShortintCompressedBootstrappingKey variant");
    // SeededGgswCiphertextList { data, glwe_size, polynomial_size, decomp_base_log, decomp_level_count, compression_seed, ciphertext_modulus }
    if (!r.vec_u64(k.bsk_bodies, "bootstrap key bodies")) return false;
    const uint64_t glwe_size = r.u64("glwe_size"), poly_size = r.u64("polynomial_size");
    const uint64_t pbs_base_log = r.u64("bootstrap decomp_base_log"), pbs_level = r.u64("bootstrap decomp_level_count");
    r.take(k.bsk_seed, 16, "bootstrap compression seed");
    if (!r.native_modulus("bootstrap key")) return false;
    if (variant == 1) {
        const uint64_t g = r.u64("grouping_factor");
        uint8_t deterministic = 0;
        r.take(&deterministic, 1, "deterministic_execution");
        if (r.ok && (g < 1 || g > 16)) return r.fail("grouping_factor out of range");
        k.grouping_factor = (uint32_t)g;
    }
    if (!r.ok) return false;
    if (glwe_size < 2 || glwe_size > 64 || !poly_size_ok(poly_size) || !decomposition_ok(pbs_base_log, pbs_level))
        return r.fail("inconsistent seeded bootstrap key header");
    p.poly_size = (uint32_t)poly_size;
    p.pbs_base_log = (uint32_t)pbs_base_log;
    p.pbs_level = (uint32_t)pbs_level;
    const uint64_t mm = r.u64("message_modulus"), cm = r.u64("carry_modulus");
    if (r.ok && (mm < 1 || mm > (1u << 16) || cm < 1 || cm > (1u << 16))) return r.fail("message / carry modulus out of range");
    p.message_modulus = (uint32_t)mm;
    p.carry_modulus = (uint32_t)cm;
    k.max_degree = r.u64("max_degree");
    if (!r.native_modulus("server key")) return false;
    const uint32_t order = r.u32("pbs_order");
    if (!r.ok) return false;
    if (order > 1) return r.fail("unknown PBSOrder variant");
    p.pbs_order = order;
    p.glwe_dim = (uint32_t)(glwe_size - 1);
    if (in_dim != (size_t)p.glwe_dim * p.poly_size) return r.fail("keyswitch key input dimension does not match glwe_dim * poly_size");
    const size_t per_ggsw = (size_t)p.pbs_level * glwe_size * p.poly_size;
    if (k.bsk_bodies.size() % per_ggsw) return r.fail("bootstrap key body count is not a whole number of GGSW ciphertexts");
    const size_t ggsw = k.bsk_bodies.size() / per_ggsw;
    const size_t want = k.grouping_factor ? ((size_t)p.lwe_dim / k.grouping_factor) << k.grouping_factor : (size_t)p.lwe_dim;
    if (ggsw != want || (k.grouping_factor && p.lwe_dim % k.grouping_factor))
        return r.fail("bootstrap key GGSW count does not match the keyswitch key's output dimension");
    if (!want_data) {
        k.ksk_bodies.clear();
        k.bsk_bodies.clear();
    }
    return true;
}

int report(const Reader &r) { return tfhe_hip_set_error(("bincode: " + r.err).c_str()); }

} // namespace

TFHE_HIP_API int tfhe_hip_compressed_server_key_info(const uint8_t *bytes, size_t len, tfhe_hip_params *params,
                                                     uint32_t *grouping_factor, uint64_t *max_degree, size_t *consumed)
{
    if (!bytes || !params) return tfhe_hip_set_error("bytes / params pointer is null");
    Reader r{bytes, len};
    CompressedKey k;
    if (!read_compressed_key(r, k, false)) return report(r);
    *params = k.params;
    if (grouping_factor) *grouping_factor = k.grouping_factor;
    if (max_degree) *max_degree = k.max_degree;
    if (consumed) *consumed = r.pos;
    return 0;
}

TFHE_HIP_API int tfhe_hip_server_key_create_from_compressed(tfhe_hip_context *ctx, tfhe_hip_server_key **out, const uint8_t *bytes,
                                                            size_t len, tfhe_hip_params *params_out, uint32_t *grouping_factor_out)
{
    if (out) *out = nullptr;
    if (!bytes) return tfhe_hip_set_error("bytes pointer is null");
    Reader r{bytes, len};
    CompressedKey k;
    if (!read_compressed_key(r, k, true)) return report(r);
    if (params_out) *params_out = k.params;
    if (grouping_factor_out) *grouping_factor_out = k.grouping_factor;
    return tfhe_hip_server_key_create_seeded(ctx, out, &k.params, k.grouping_factor, k.bsk_bodies.data(), k.bsk_bodies.size(), k.bsk_seed,
                                             k.ksk_bodies.data(), k.ksk_bodies.size(), k.ksk_seed);
}

TFHE_HIP_API int tfhe_hip_compressed_server_key_serialize(const tfhe_hip_params *params, uint32_t grouping_factor,
                                                          const uint64_t *bsk_bodies, size_t bsk_bodies_len, const uint8_t *bsk_seed,
                                                          const uint64_t *ksk_bodies, size_t ksk_bodies_len, const uint8_t *ksk_seed,
                                                          uint8_t *out, size_t cap, size_t *written)
{
    if (!params || !bsk_bodies || !ksk_bodies || !bsk_seed || !ksk_seed || !written) return tfhe_hip_set_error("null pointer argument");
    const tfhe_hip_params &p = *params;
    Writer w;
    w.vec_u64(ksk_bodies, ksk_bodies_len);
    w.u64(p.ks_base_log);
    w.u64(p.ks_level);
    w.u64((uint64_t)p.lwe_dim + 1);
    w.raw(ksk_seed, 16);
    w.native_modulus();
    w.u32(grouping_factor ? 1 : 0);
    w.vec_u64(bsk_bodies, bsk_bodies_len);
    w.u64((uint64_t)p.glwe_dim + 1);
    w.u64(p.poly_size);
    w.u64(p.pbs_base_log);
    w.u64(p.pbs_level);
    w.raw(bsk_seed, 16);
    w.native_modulus();
    if (grouping_factor) {
        w.u64(grouping_factor);
        const uint8_t deterministic = 1;
        w.raw(&deterministic, 1);
    }
    w.u64(p.message_modulus);
    w.u64(p.carry_modulus);
    w.u64((uint64_t)p.message_modulus * p.carry_modulus - 1); // MaxDegree (shortint/engine/server_side.rs)
    w.native_modulus();
    w.u32(p.pbs_order);
    *written = w.b.size();
    if (!out || cap < w.b.size()) return out ? tfhe_hip_set_error("output buffer too small") : 0; // out == NULL: size query
    std::memcpy(out, w.b.data(), w.b.size());
    return 0;
}

// shortint::Ciphertext { ct: LweCiphertext { data, ciphertext_modulus }, degree, noise_level, message_modulus, carry_modulus, pbs_order }
TFHE_HIP_API int tfhe_hip_shortint_ciphertext_serialize(const uint64_t *lwe, size_t lwe_len, uint64_t degree, uint64_t noise_level,
                                                        uint32_t message_modulus, uint32_t carry_modulus, uint32_t pbs_order,
                                                        uint8_t *out, size_t cap, size_t *written)
{
    if (!lwe || !written) return tfhe_hip_set_error("null pointer argument");
    if (pbs_order > 1) return tfhe_hip_set_error("pbs_order must be 0 or 1");
    Writer w;
    w.vec_u64(lwe, lwe_len);
    w.native_modulus();
    w.u64(degree);
    w.u64(noise_level);
    w.u64(message_modulus);
    w.u64(carry_modulus);
    w.u32(pbs_order);
    *written = w.b.size();
    if (!out || cap < w.b.size()) return out ? tfhe_hip_set_error("output buffer too small") : 0;
    std::memcpy(out, w.b.data(), w.b.size());
    return 0;
}

TFHE_HIP_API int tfhe_hip_shortint_ciphertext_deserialize(const uint8_t *bytes, size_t len, uint64_t *lwe_out, size_t lwe_cap,
                                                          size_t *lwe_len, uint64_t *degree, uint64_t *noise_level,
                                                          uint32_t *message_modulus, uint32_t *carry_modulus, uint32_t *pbs_order,
                                                          size_t *consumed)
{
    if (!bytes || !lwe_len) return tfhe_hip_set_error("null pointer argument");
    Reader r{bytes, len};
    std::vector<uint64_t> data;
    if (!r.vec_u64(data, "LWE ciphertext data") || !r.native_modulus("ciphertext")) return report(r);
    const uint64_t deg = r.u64("degree"), noise = r.u64("noise_level"), mm = r.u64("message_modulus"), cm = r.u64("carry_modulus");
    const uint32_t order = r.u32("pbs_order");
    if (!r.ok) return report(r);
    if (order > 1) {
        r.fail("unknown PBSOrder variant");
        return report(r);
    }
    *lwe_len = data.size();
    if (degree) *degree = deg;
    if (noise_level) *noise_level = noise;
    if (message_modulus) *message_modulus = (uint32_t)mm;
    if (carry_modulus) *carry_modulus = (uint32_t)cm;
    if (pbs_order) *pbs_order = order;
    if (consumed) *consumed = r.pos;
    if (lwe_out) {
        if (lwe_cap < data.size()) return tfhe_hip_set_error("lwe_out is too small");
        std::memcpy(lwe_out, data.data(), data.size() * 8);
    }
    return 0;
}

// shortint::CompressedCiphertext { ct: SeededLweCiphertext { data: u64 (the body), lwe_size, compression_seed, ciphertext_modulus },
//                                  degree, message_modulus, carry_modulus, pbs_order, noise_level }
// (tfhe/src/shortint/ciphertext/mod.rs:470-478, core_crypto/entities/seeded_lwe_ciphertext.rs:12-18)
TFHE_HIP_API int tfhe_hip_shortint_compressed_ciphertext_serialize(uint64_t body, uint64_t lwe_size, const uint8_t *seed, uint64_t degree,
                                                                   uint64_t noise_level, uint32_t message_modulus,
                                                                   uint32_t carry_modulus, uint32_t pbs_order, uint8_t *out, size_t cap,
                                                                   size_t *written)
{
    if (!seed || !written) return tfhe_hip_set_error("null pointer argument");
    if (pbs_order > 1) return tfhe_hip_set_error("pbs_order must be 0 or 1");
    Writer w;
    w.u64(body);
    w.u64(lwe_size);
    w.raw(seed, 16);
    w.native_modulus();
    w.u64(degree);
    w.u64(message_modulus);
    w.u64(carry_modulus);
    w.u32(pbs_order);
    w.u64(noise_level);
    *written = w.b.size();
    if (!out || cap < w.b.size()) return out ? tfhe_hip_set_error("output buffer too small") : 0;
    std::memcpy(out, w.b.data(), w.b.size());
    return 0;
}

TFHE_HIP_API int tfhe_hip_shortint_compressed_ciphertext_deserialize(const uint8_t *bytes, size_t len, uint64_t *body, uint64_t *lwe_size,
                                                                     uint8_t *seed, uint64_t *degree, uint64_t *noise_level,
                                                                     uint32_t *message_modulus, uint32_t *carry_modulus,
                                                                     uint32_t *pbs_order, size_t *consumed)
{
    if (!bytes || !body || !lwe_size || !seed) return tfhe_hip_set_error("null pointer argument");
    Reader r{bytes, len};
    const uint64_t b = r.u64("body"), size = r.u64("lwe_size");
    uint8_t sd[16];
    r.take(sd, 16, "compression seed");
    if (!r.native_modulus("compressed ciphertext")) return report(r);
    const uint64_t deg = r.u64("degree"), mm = r.u64("message_modulus"), cm = r.u64("carry_modulus");
    const uint32_t order = r.u32("pbs_order");
    const uint64_t noise = r.u64("noise_level");
    if (!r.ok) return report(r);
    if (order > 1 || size < 2) {
        r.fail(order > 1 ? "unknown PBSOrder variant" : "lwe_size must be at least 2");
        return report(r);
    }
    *body = b;
    *lwe_size = size;
    std::memcpy(seed, sd, 16);
    if (degree) *degree = deg;
    if (noise_level) *noise_level = noise;
    if (message_modulus) *message_modulus = (uint32_t)mm;
    if (carry_modulus) *carry_modulus = (uint32_t)cm;
    if (pbs_order) *pbs_order = order;
    if (consumed) *consumed = r.pos;
    return 0;
}

// ---- standard-domain (uncompressed) keys ---------------------------------------------------------------------------------
// LweKeyswitchKey { data, decomp_base_log, decomp_level_count, output_lwe_size, ciphertext_modulus }
// LweBootstrapKey { ggsw_list: GgswCiphertextList { data, glwe_size, polynomial_size, decomp_base_log, decomp_level_count,
//                                                   ciphertext_modulus } }   (newtype structs are transparent in bincode)
namespace {

struct StdKsk {
    std::vector<uint64_t> data;
    uint64_t base_log = 0, level = 0, out_size = 0;
};
struct StdBsk {
    std::vector<uint64_t> data;
    uint64_t glwe_size = 0, poly_size = 0, base_log = 0, level = 0;
};
bool read_std_ksk(Reader &r, StdKsk &k)
{
    if (!r.vec_u64(k.data, "keyswitch key data")) return false;
    k.base_log = r.u64("keyswitch decomp_base_log");
    k.level = r.u64("keyswitch decomp_level_count");
    k.out_size = r.u64("keyswitch output_lwe_size");
    if (!r.native_modulus("keyswitch key")) return false;
    // The header fields are untrusted u64s: bound each one BEFORE multiplying (a wrapped product is a division by zero) and
    // so that the uint32 fields handed on hold exactly what was checked here.
    if (!decomposition_ok(k.base_log, k.level)) return r.fail("keyswitch key: decomposition base_log / level_count out of range");
    if (k.out_size < 2 || k.out_size > kMaxDim) return r.fail("keyswitch key: output_lwe_size out of range");
    // LweKeyswitchKey::from_container asserts (lwe_keyswitch_key.rs:175-200): a whole number of input-key elements
    const uint64_t per_input = k.level * k.out_size; // <= 64 * 2^32
    if (k.data.empty() || k.data.size() % per_input) return r.fail("inconsistent keyswitch key header");
    if (k.data.size() / per_input > kMaxDim) return r.fail("keyswitch key: input dimension out of range");
    return true;
}
bool read_std_bsk(Reader &r, StdBsk &k)
{
    if (!r.vec_u64(k.data, "bootstrap key data")) return false;
    k.glwe_size = r.u64("glwe_size");
    k.poly_size = r.u64("polynomial_size");
    k.base_log = r.u64("bootstrap decomp_base_log");
    k.level = r.u64("bootstrap decomp_level_count");
    if (!r.native_modulus("bootstrap key")) return false;
    if (!decomposition_ok(k.base_log, k.level)) return r.fail("bootstrap key: decomposition base_log / level_count out of range");
    if (k.glwe_size < 2 || k.glwe_size > 64) return r.fail("bootstrap key: glwe_size out of range");
    if (!poly_size_ok(k.poly_size))
        return r.fail("bootstrap key: polynomial_size is not a power of two up to 2^20");
    // GgswCiphertextList::from_container asserts (ggsw_ciphertext_list.rs:94-115): a whole number of GGSW ciphertexts
    const uint64_t per_ggsw = k.level * k.glwe_size * k.glwe_size * k.poly_size; // <= 2^6 * 2^12 * 2^20: no wrap
    if (k.data.empty() || k.data.size() % per_ggsw) return r.fail("inconsistent bootstrap key header");
    if (k.data.size() / per_ggsw > kMaxDim) return r.fail("bootstrap key: input dimension out of range");
    return true;
}
int copy_out(const std::vector<uint64_t> &data, uint64_t *out, size_t cap, size_t *out_len)
{
    *out_len = data.size();
    if (out) {
        if (cap < data.size()) return tfhe_hip_set_error("output array is too small");
        std::memcpy(out, data.data(), data.size() * 8);
    }
    return 0;
}

} // namespace

TFHE_HIP_API int tfhe_hip_lwe_keyswitch_key_deserialize(const uint8_t *bytes, size_t len, uint64_t *ksk_out, size_t cap, size_t *ksk_len,
                                                        uint32_t *decomp_base_log, uint32_t *decomp_level_count, uint32_t *input_lwe_dim,
                                                        uint32_t *output_lwe_dim, size_t *consumed)
{
    if (!bytes || !ksk_len) return tfhe_hip_set_error("null pointer argument");
    Reader r{bytes, len};
    StdKsk k;
    if (!read_std_ksk(r, k)) return report(r);
    if (decomp_base_log) *decomp_base_log = (uint32_t)k.base_log;
    if (decomp_level_count) *decomp_level_count = (uint32_t)k.level;
    if (input_lwe_dim) *input_lwe_dim = (uint32_t)(k.data.size() / (k.level * k.out_size));
    if (output_lwe_dim) *output_lwe_dim = (uint32_t)(k.out_size - 1);
    if (consumed) *consumed = r.pos;
    return copy_out(k.data, ksk_out, cap, ksk_len);
}

TFHE_HIP_API int tfhe_hip_lwe_bootstrap_key_deserialize(const uint8_t *bytes, size_t len, uint64_t *bsk_out, size_t cap, size_t *bsk_len,
                                                        uint32_t *glwe_dim, uint32_t *poly_size, uint32_t *decomp_base_log,
                                                        uint32_t *decomp_level_count, uint32_t *input_lwe_dim, size_t *consumed)
{
    if (!bytes || !bsk_len) return tfhe_hip_set_error("null pointer argument");
    Reader r{bytes, len};
    StdBsk k;
    if (!read_std_bsk(r, k)) return report(r);
    if (glwe_dim) *glwe_dim = (uint32_t)(k.glwe_size - 1);
    if (poly_size) *poly_size = (uint32_t)k.poly_size;
    if (decomp_base_log) *decomp_base_log = (uint32_t)k.base_log;
    if (decomp_level_count) *decomp_level_count = (uint32_t)k.level;
    if (input_lwe_dim) *input_lwe_dim = (uint32_t)(k.data.size() / (k.level * k.glwe_size * k.glwe_size * k.poly_size));
    if (consumed) *consumed = r.pos;
    return copy_out(k.data, bsk_out, cap, bsk_len);
}

TFHE_HIP_API int tfhe_hip_lwe_keyswitch_key_serialize(const uint64_t *ksk, size_t ksk_len, uint32_t decomp_base_log,
                                                      uint32_t decomp_level_count, uint32_t output_lwe_dim, uint8_t *out, size_t cap,
                                                      size_t *written)
{
    if (!ksk || !written) return tfhe_hip_set_error("null pointer argument");
    *written = 8 + ksk_len * 8 + 3 * 8 + 24;
    if (!out) return 0; // size query
    if (cap < *written) return tfhe_hip_set_error("output buffer too small");
    Writer w;
    w.b.reserve(*written);
    w.vec_u64(ksk, ksk_len);
    w.u64(decomp_base_log);
    w.u64(decomp_level_count);
    w.u64((uint64_t)output_lwe_dim + 1);
    w.native_modulus();
    std::memcpy(out, w.b.data(), w.b.size());
    return 0;
}

TFHE_HIP_API int tfhe_hip_lwe_bootstrap_key_serialize(const uint64_t *bsk, size_t bsk_len, uint32_t glwe_dim, uint32_t poly_size,
                                                      uint32_t decomp_base_log, uint32_t decomp_level_count, uint8_t *out, size_t cap,
                                                      size_t *written)
{
    if (!bsk || !written) return tfhe_hip_set_error("null pointer argument");
    *written = 8 + bsk_len * 8 + 4 * 8 + 24;
    if (!out) return 0;
    if (cap < *written) return tfhe_hip_set_error("output buffer too small");
    Writer w;
    w.b.reserve(*written);
    w.vec_u64(bsk, bsk_len);
    w.u64((uint64_t)glwe_dim + 1);
    w.u64(poly_size);
    w.u64(decomp_base_log);
    w.u64(decomp_level_count);
    w.native_modulus();
    std::memcpy(out, w.b.data(), w.b.size());
    return 0;
}

// What a server does with the two keys a tfhe-rs client serialized: ServerKey construction from the STANDARD-domain keys
// (shortint/engine/server_side.rs:54-...: the Fourier conversion happens on the device, tfhe_hip_server_key_create).
TFHE_HIP_API int tfhe_hip_server_key_create_from_standard_keys(tfhe_hip_context *ctx, tfhe_hip_server_key **out, const uint8_t *ksk_bytes,
                                                               size_t ksk_bytes_len, const uint8_t *bsk_bytes, size_t bsk_bytes_len,
                                                               uint32_t message_modulus, uint32_t carry_modulus, uint32_t pbs_order,
                                                               tfhe_hip_params *params_out)
{
    if (out) *out = nullptr;
    if (!ksk_bytes || !bsk_bytes) return tfhe_hip_set_error("bytes pointer is null");
    Reader rk{ksk_bytes, ksk_bytes_len}, rb{bsk_bytes, bsk_bytes_len};
    StdKsk k;
    StdBsk b;
    if (!read_std_ksk(rk, k)) return report(rk);
    if (!read_std_bsk(rb, b)) return report(rb);
    tfhe_hip_params p{};
    p.lwe_dim = (uint32_t)(k.out_size - 1);
    p.glwe_dim = (uint32_t)(b.glwe_size - 1);
    p.poly_size = (uint32_t)b.poly_size;
    p.pbs_base_log = (uint32_t)b.base_log;
    p.pbs_level = (uint32_t)b.level;
    p.ks_base_log = (uint32_t)k.base_log;
    p.ks_level = (uint32_t)k.level;
    p.message_modulus = message_modulus;
    p.carry_modulus = carry_modulus;
    p.pbs_order = pbs_order;
    // the two keys must describe the same pair of secret keys (ServerKey's invariants, shortint/server_key/mod.rs:176-230)
    if (k.data.size() / (k.level * k.out_size) != (size_t)p.glwe_dim * p.poly_size)
        return tfhe_hip_set_error("bincode: the keyswitch key's input dimension is not glwe_dim * poly_size of the bootstrap key");
    if (b.data.size() / (b.level * b.glwe_size * b.glwe_size * b.poly_size) != p.lwe_dim)
        return tfhe_hip_set_error("bincode: the bootstrap key's input dimension is not the keyswitch key's output dimension");
    if (params_out) *params_out = p;
    return tfhe_hip_server_key_create(ctx, out, &p, b.data.data(), b.data.size(), k.data.data(), k.data.size());
}
