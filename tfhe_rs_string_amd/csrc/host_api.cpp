// host_api.cpp -- C-ABI of the host op-graph layer (include/tfhe_hip_strings.h) over host_ops.hpp.
#include "../../include/tfhe_hip_strings.h"

#include <cstring>
#include <string>

#include "host_ops.hpp"

#define TFHE_HIP_API extern "C" __attribute__((visibility("default")))

using namespace tfhe_hip::host;

// defined in api.cpp: sets the thread-local error message
extern "C" __attribute__((visibility("hidden"))) int tfhe_hip_set_error(const char *msg);

struct tfhe_hip_host_key {
    ServerKey sk;
    std::mutex mu; // one operation at a time per host key (the device key underneath is shared freely between host keys)
    tfhe_hip_host_key(tfhe_hip_server_key *s, const tfhe_hip_params &p) : sk(s, p) {}
};

namespace {

constexpr size_t BLOCKS_PER_CHAR = 4; // NUMBER_BLOCKS, fhe_strings/ciphertext.rs:14

template <typename F> int guarded(F &&f)
{
    try {
        return f();
    } catch (const std::exception &e) {
        return tfhe_hip_set_error(e.what());
    } catch (...) {
        return tfhe_hip_set_error("unknown exception");
    }
}
// an API call on a host key: whatever the call staged and did not materialise is dropped when it returns
// (pending bootstraps refer to the caller's arrays, which are only valid during the call)
// Outside a caller's batch the call is its own small batch: the outputs are registered while the graph is built and
// written together at the end, by several host threads (a split returns thousands of blocks).
template <typename F> int guarded(tfhe_hip_host_key *hk, F &&f)
{
    struct Scope {
        tfhe_hip_host_key *hk;
        ~Scope()
        {
            if (hk) hk->sk.discard_pending();
        }
    } scope{hk};
    std::unique_lock<std::mutex> lock;
    if (hk) lock = std::unique_lock<std::mutex>(hk->mu);
    return guarded([&]() -> int {
        const bool own_batch = hk && !hk->sk.batching();
        if (own_batch) hk->sk.begin_batch();
        static const bool trace = std::getenv("TFHE_HIP_HOST_TRACE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        int rc;
        try {
            rc = f();
            if (trace)
                std::fprintf(stderr, "[host] graph built in %.2f ms\n",
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        } catch (...) {
            if (own_batch) hk->sk.abort_batch();
            throw;
        }
        if (own_batch) {
            if (rc == 0) hk->sk.end_batch();
            else hk->sk.abort_batch();
        }
        return rc;
    });
}

RadixCiphertext radix_from(const ServerKey &sk, const uint64_t *p, size_t blocks, uint64_t degree)
{
    return sk.from_raw_rows(p, blocks, degree);
}

FheString string_from(const ServerKey &sk, const uint64_t *p, size_t len)
{
    FheString s(len);
    const uint64_t fresh = sk.message_modulus() - 1;
    for (size_t c = 0; c < len; c++) s[c] = radix_from(sk, p + c * BLOCKS_PER_CHAR * sk.big_size(), BLOCKS_PER_CHAR, fresh);
    return s;
}

void put(ServerKey &sk, const Ciphertext &c, uint64_t *out) { sk.to_raw(c, out); }

int need(const void *p, const char *what)
{
    if (!p) return tfhe_hip_set_error((std::string(what) + " pointer is null").c_str());
    return 0;
}

} // namespace

TFHE_HIP_API int tfhe_hip_host_key_create(tfhe_hip_server_key *sk, const tfhe_hip_params *params, tfhe_hip_host_key **out)
{
    return guarded([&]() -> int {
        if (need(out, "out")) return 1;
        *out = nullptr;
        if (need(sk, "server key") || need(params, "params")) return 1;
        // the op-graphs keep their ciphertexts under the big key (EncryptionKeyChoice::Big, what the string layer uses)
        if (params->pbs_order != 0) return tfhe_hip_set_error("the host op-graph layer works on KeyswitchBootstrap (pbs_order 0) keys");
        tfhe_hip_params have;
        if (tfhe_hip_server_key_get_params(sk, &have)) return 1;
        if (std::memcmp(&have, params, sizeof(have)) != 0) // a mismatch would mis-size every ciphertext row
            return tfhe_hip_set_error("params differ from the parameters the server key was created with");
        *out = new tfhe_hip_host_key(sk, *params);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_host_key_destroy(tfhe_hip_host_key *hk)
{
    delete hk;
    return 0;
}

TFHE_HIP_API int tfhe_hip_host_key_batch_begin(tfhe_hip_host_key *hk)
{
    return guarded([&]() -> int {
        if (need(hk, "host key")) return 1;
        std::lock_guard<std::mutex> lock(hk->mu);
        hk->sk.begin_batch();
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_host_key_batch_end(tfhe_hip_host_key *hk)
{
    return guarded([&]() -> int {
        if (need(hk, "host key")) return 1;
        std::lock_guard<std::mutex> lock(hk->mu);
        hk->sk.end_batch();
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_host_key_set_num_blocks(tfhe_hip_host_key *hk, size_t num_blocks)
{
    return guarded([&]() -> int {
        if (need(hk, "host key")) return 1;
        std::lock_guard<std::mutex> lock(hk->mu);
        if (hk->sk.batching()) return tfhe_hip_set_error("num_blocks cannot change while a batch is open");
        hk->sk.set_index_blocks(num_blocks);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_host_key_num_blocks(tfhe_hip_host_key *hk, size_t *num_blocks)
{
    return guarded([&]() -> int {
        if (need(hk, "host key") || need(num_blocks, "num_blocks")) return 1;
        *num_blocks = hk->sk.index_blocks();
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_host_key_set_strict_outputs(tfhe_hip_host_key *hk, int on)
{
    return guarded([&]() -> int {
        if (need(hk, "host key")) return 1;
        std::lock_guard<std::mutex> lock(hk->mu);
        if (hk->sk.batching()) return tfhe_hip_set_error("the output contract cannot change while a batch is open");
        hk->sk.set_strict_outputs(on != 0);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_host_key_output_metadata(tfhe_hip_host_key *hk, const uint64_t *out_blocks, size_t n_blocks,
                                                   uint64_t *degrees, uint64_t *noise_levels)
{
    return guarded([&]() -> int {
        if (need(hk, "host key") || need(out_blocks, "out_blocks")) return 1;
        std::lock_guard<std::mutex> lock(hk->mu);
        for (size_t i = 0; i < n_blocks; i++) {
            uint64_t d = 0, n = 0;
            if (!hk->sk.output_metadata(out_blocks + i * hk->sk.big_size(), &d, &n))
                return tfhe_hip_set_error(("block " + std::to_string(i) + " was not written by the last call / batch on this host key").c_str());
            if (degrees) degrees[i] = d;
            if (noise_levels) noise_levels[i] = n;
        }
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_host_key_stats(tfhe_hip_host_key *hk, uint64_t *pbs_count, uint64_t *launches, int reset)
{
    if (need(hk, "host key")) return 1;
    if (pbs_count) *pbs_count = hk->sk.pbs_count;
    if (launches) *launches = hk->sk.launches;
    if (reset) hk->sk.pbs_count = hk->sk.launches = 0;
    return 0;
}

TFHE_HIP_API int tfhe_hip_shortint_apply_lookup_table_batch(tfhe_hip_host_key *hk, uint64_t *cts, uint64_t *degrees,
                                                            uint64_t *noise_levels, const uint64_t *luts,
                                                            const uint64_t *lut_degrees, size_t n_luts,
                                                            const uint32_t *lut_idx, size_t B)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(cts, "cts") || need(degrees, "degrees") || need(noise_levels, "noise_levels") ||
            need(luts, "luts") || need(lut_degrees, "lut_degrees") || need(lut_idx, "lut_idx"))
            return 1;
        ServerKey &sk = hk->sk;
        const size_t big = sk.big_size(), lut_len = (size_t)(sk.params().glwe_dim + 1) * sk.params().poly_size;
        std::vector<LookupTable> tabs(n_luts);
        for (size_t u = 0; u < n_luts; u++) {
            tabs[u].acc = std::make_shared<const std::vector<uint64_t>>(luts + u * lut_len, luts + (u + 1) * lut_len);
            tabs[u].degree = lut_degrees[u];
        }
        std::vector<Ciphertext> v(B);
        std::vector<Ciphertext *> ptr(B);
        std::vector<const LookupTable *> lp(B);
        for (size_t i = 0; i < B; i++) {
            if (lut_idx[i] >= n_luts) return tfhe_hip_set_error("lut_idx out of range");
            v[i] = sk.from_raw(cts + i * big, degrees[i], noise_levels[i]);
            ptr[i] = &v[i];
            lp[i] = &tabs[lut_idx[i]];
        }
        sk.apply_lookup_table_batch(ptr, lp);
        for (size_t i = 0; i < B; i++) {
            put(sk, v[i], cts + i * big);
            degrees[i] = v[i].degree;
            noise_levels[i] = v[i].noise_level;
        }
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_integer_eq(tfhe_hip_host_key *hk, const uint64_t *lhs, const uint64_t *rhs, size_t num_blocks,
                                     uint64_t *out_block)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(lhs, "lhs") || need(rhs, "rhs") || need(out_block, "out")) return 1;
        ServerKey &sk = hk->sk;
        const uint64_t fresh = sk.message_modulus() - 1;
        put(sk, eq_parallelized(sk, radix_from(sk, lhs, num_blocks, fresh), radix_from(sk, rhs, num_blocks, fresh)), out_block);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_integer_mul(tfhe_hip_host_key *hk, const uint64_t *lhs, const uint64_t *rhs, size_t num_blocks,
                                      uint64_t *out_blocks)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(lhs, "lhs") || need(rhs, "rhs") || need(out_blocks, "out")) return 1;
        ServerKey &sk = hk->sk;
        const uint64_t fresh = sk.message_modulus() - 1;
        const RadixCiphertext r = mul_parallelized(sk, radix_from(sk, lhs, num_blocks, fresh), radix_from(sk, rhs, num_blocks, fresh));
        for (size_t b = 0; b < num_blocks; b++) put(sk, r[b], out_blocks + b * sk.big_size());
        return 0;
    });
}

namespace {
using RadixBin = RadixCiphertext (*)(ServerKey &, const RadixCiphertext &, const RadixCiphertext &);
void put_radix(ServerKey &sk, const RadixCiphertext &r, uint64_t *out)
{
    for (size_t b = 0; b < r.size(); b++) put(sk, r[b], out + b * sk.big_size());
}
int radix_bin_api(tfhe_hip_host_key *hk, const uint64_t *lhs, const uint64_t *rhs, size_t nb, uint64_t *out, RadixBin op)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(lhs, "lhs") || need(rhs, "rhs") || need(out, "out")) return 1;
        ServerKey &sk = hk->sk;
        const uint64_t fresh = sk.message_modulus() - 1;
        put_radix(sk, op(sk, radix_from(sk, lhs, nb, fresh), radix_from(sk, rhs, nb, fresh)), out);
        return 0;
    });
}
int cmp_op(int op, Cmp *out)
{
    switch (op) {
    case 0: *out = Cmp::GT; return 0;
    case 1: *out = Cmp::GE; return 0;
    case 2: *out = Cmp::LT; return 0;
    case 3: *out = Cmp::LE; return 0;
    default: return tfhe_hip_set_error("comparison op must be 0 (gt), 1 (ge), 2 (lt) or 3 (le)");
    }
}
} // namespace

TFHE_HIP_API int tfhe_hip_integer_compare(tfhe_hip_host_key *hk, const uint64_t *lhs, const uint64_t *rhs, size_t num_blocks,
                                          int op, uint64_t *out_block)
{
    return guarded(hk, [&]() -> int {
        Cmp c;
        if (need(hk, "host key") || need(lhs, "lhs") || need(rhs, "rhs") || need(out_block, "out") || cmp_op(op, &c)) return 1;
        ServerKey &sk = hk->sk;
        const uint64_t fresh = sk.message_modulus() - 1;
        put(sk, compare_parallelized(sk, radix_from(sk, lhs, num_blocks, fresh), radix_from(sk, rhs, num_blocks, fresh), c), out_block);
        return 0;
    });
}
TFHE_HIP_API int tfhe_hip_integer_scalar_compare(tfhe_hip_host_key *hk, const uint64_t *lhs, size_t num_blocks,
                                                 uint64_t scalar, int op, uint64_t *out_block)
{
    return guarded(hk, [&]() -> int {
        Cmp c;
        if (need(hk, "host key") || need(lhs, "lhs") || need(out_block, "out") || cmp_op(op, &c)) return 1;
        ServerKey &sk = hk->sk;
        put(sk, scalar_compare_parallelized(sk, radix_from(sk, lhs, num_blocks, sk.message_modulus() - 1), scalar, c), out_block);
        return 0;
    });
}
TFHE_HIP_API int tfhe_hip_integer_min_max(tfhe_hip_host_key *hk, const uint64_t *lhs, const uint64_t *rhs, size_t num_blocks,
                                          int want_max, uint64_t *out_blocks)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(lhs, "lhs") || need(rhs, "rhs") || need(out_blocks, "out")) return 1;
        ServerKey &sk = hk->sk;
        const uint64_t fresh = sk.message_modulus() - 1;
        put_radix(sk, min_or_max_parallelized(sk, radix_from(sk, lhs, num_blocks, fresh), radix_from(sk, rhs, num_blocks, fresh), want_max != 0),
                  out_blocks);
        return 0;
    });
}
TFHE_HIP_API int tfhe_hip_integer_if_then_else(tfhe_hip_host_key *hk, const uint64_t *condition, const uint64_t *true_ct,
                                               const uint64_t *false_ct, size_t num_blocks, uint64_t *out_blocks)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(condition, "condition") || need(true_ct, "true_ct") || need(false_ct, "false_ct") ||
            need(out_blocks, "out"))
            return 1;
        ServerKey &sk = hk->sk;
        const uint64_t fresh = sk.message_modulus() - 1;
        put_radix(sk, if_then_else_parallelized(sk, sk.from_raw(condition, 1), radix_from(sk, true_ct, num_blocks, fresh),
                                                radix_from(sk, false_ct, num_blocks, fresh)),
                  out_blocks);
        return 0;
    });
}
TFHE_HIP_API int tfhe_hip_integer_add(tfhe_hip_host_key *hk, const uint64_t *lhs, const uint64_t *rhs, size_t num_blocks,
                                      uint64_t *out_blocks)
{
    return radix_bin_api(hk, lhs, rhs, num_blocks, out_blocks, add_parallelized);
}
TFHE_HIP_API int tfhe_hip_integer_sub(tfhe_hip_host_key *hk, const uint64_t *lhs, const uint64_t *rhs, size_t num_blocks,
                                      uint64_t *out_blocks)
{
    return radix_bin_api(hk, lhs, rhs, num_blocks, out_blocks, sub_parallelized);
}
TFHE_HIP_API int tfhe_hip_integer_neg(tfhe_hip_host_key *hk, const uint64_t *ct, size_t num_blocks, uint64_t *out_blocks)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(ct, "ct") || need(out_blocks, "out")) return 1;
        ServerKey &sk = hk->sk;
        put_radix(sk, neg_parallelized(sk, radix_from(sk, ct, num_blocks, sk.message_modulus() - 1)), out_blocks);
        return 0;
    });
}
TFHE_HIP_API int tfhe_hip_integer_scalar_add(tfhe_hip_host_key *hk, const uint64_t *lhs, size_t num_blocks, uint64_t scalar,
                                             uint64_t *out_blocks)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(lhs, "lhs") || need(out_blocks, "out")) return 1;
        ServerKey &sk = hk->sk;
        put_radix(sk, scalar_add_parallelized(sk, radix_from(sk, lhs, num_blocks, sk.message_modulus() - 1), scalar), out_blocks);
        return 0;
    });
}
TFHE_HIP_API int tfhe_hip_integer_scalar_mul(tfhe_hip_host_key *hk, const uint64_t *lhs, size_t num_blocks, uint64_t scalar,
                                             uint64_t *out_blocks)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(lhs, "lhs") || need(out_blocks, "out")) return 1;
        ServerKey &sk = hk->sk;
        put_radix(sk, scalar_mul_parallelized(sk, radix_from(sk, lhs, num_blocks, sk.message_modulus() - 1), scalar), out_blocks);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_eq(tfhe_hip_host_key *hk, const uint64_t *s1, size_t len1, const uint64_t *s2, size_t len2,
                                    uint64_t *out_block)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(out_block, "out") || (len1 && need(s1, "s1")) || (len2 && need(s2, "s2"))) return 1;
        ServerKey &sk = hk->sk;
        put(sk, string_eq(sk, string_from(sk, s1, len1), string_from(sk, s2, len2)), out_block);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_contains(tfhe_hip_host_key *hk, const uint64_t *hay, size_t hay_len, const uint64_t *needle,
                                          size_t needle_len, uint64_t *out_block)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(out_block, "out") || (hay_len && need(hay, "haystack")) ||
            (needle_len && need(needle, "needle")))
            return 1;
        ServerKey &sk = hk->sk;
        put(sk, string_contains(sk, string_from(sk, hay, hay_len), string_from(sk, needle, needle_len)), out_block);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_contains_windows(tfhe_hip_host_key *hk, const uint64_t *hay, size_t hay_len,
                                                  const uint64_t *needle, size_t needle_len, size_t w_begin,
                                                  size_t w_end, uint64_t *out_blocks)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(hay, "haystack") || need(needle, "needle")) return 1;
        ServerKey &sk = hk->sk;
        const size_t nw = string_contains_window_count(hay_len, needle_len);
        if (w_begin > w_end || w_end > nw) return tfhe_hip_set_error("window range out of bounds");
        if (w_begin == w_end) return 0;
        if (need(out_blocks, "out")) return 1;
        const std::vector<Ciphertext> m =
            string_contains_windows(sk, string_from(sk, hay, hay_len), string_from(sk, needle, needle_len), w_begin, w_end);
        for (size_t w = 0; w < m.size(); w++) put(sk, m[w], out_blocks + w * sk.big_size());
        return 0;
    });
}

namespace {
using EncOp = Ciphertext (*)(ServerKey &, const FheString &, const FheString &);
using ClearOp = Ciphertext (*)(ServerKey &, const FheString &, const std::string &);

int enc_pattern_api(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint64_t *pat, size_t pat_len,
                    uint64_t *out_block, EncOp op)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(out_block, "out") || (len && need(s, "string")) || (pat_len && need(pat, "pattern")))
            return 1;
        ServerKey &sk = hk->sk;
        put(sk, op(sk, string_from(sk, s, len), string_from(sk, pat, pat_len)), out_block);
        return 0;
    });
}
int clear_pattern_api(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear, size_t clear_len,
                      uint64_t *out_block, ClearOp op)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(out_block, "out") || (len && need(s, "string")) || (clear_len && need(clear, "pattern")))
            return 1;
        ServerKey &sk = hk->sk;
        const std::string pat(reinterpret_cast<const char *>(clear), clear_len);
        put(sk, op(sk, string_from(sk, s, len), pat), out_block);
        return 0;
    });
}
} // namespace

TFHE_HIP_API int tfhe_hip_string_ne(tfhe_hip_host_key *hk, const uint64_t *s1, size_t len1, const uint64_t *s2, size_t len2,
                                    uint64_t *out_block)
{
    return enc_pattern_api(hk, s1, len1, s2, len2, out_block, string_ne);
}
TFHE_HIP_API int tfhe_hip_string_starts_with(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint64_t *prefix,
                                             size_t prefix_len, uint64_t *out_block)
{
    return enc_pattern_api(hk, s, len, prefix, prefix_len, out_block, string_starts_with);
}
TFHE_HIP_API int tfhe_hip_string_ends_with(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint64_t *suffix,
                                           size_t suffix_len, uint64_t *out_block)
{
    return enc_pattern_api(hk, s, len, suffix, suffix_len, out_block, string_ends_with);
}
TFHE_HIP_API int tfhe_hip_string_eq_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                                          size_t clear_len, uint64_t *out_block)
{
    return clear_pattern_api(hk, s, len, clear, clear_len, out_block, string_eq_clear);
}
TFHE_HIP_API int tfhe_hip_string_contains_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                                                size_t clear_len, uint64_t *out_block)
{
    return clear_pattern_api(hk, s, len, clear, clear_len, out_block, string_contains_clear);
}
TFHE_HIP_API int tfhe_hip_string_starts_with_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len,
                                                   const uint8_t *clear, size_t clear_len, uint64_t *out_block)
{
    return clear_pattern_api(hk, s, len, clear, clear_len, out_block, string_starts_with_clear);
}
TFHE_HIP_API int tfhe_hip_string_ends_with_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                                                 size_t clear_len, uint64_t *out_block)
{
    return clear_pattern_api(hk, s, len, clear, clear_len, out_block, string_ends_with_clear);
}

namespace {
Ciphertext op_le(ServerKey &sk, const FheString &a, const FheString &b) { return string_compare(sk, a, b, true); }
Ciphertext op_ge(ServerKey &sk, const FheString &a, const FheString &b) { return string_compare(sk, a, b, false); }
Ciphertext op_le_clear(ServerKey &sk, const FheString &a, const std::string &b) { return string_compare_clear(sk, a, b, true); }
Ciphertext op_ge_clear(ServerKey &sk, const FheString &a, const std::string &b) { return string_compare_clear(sk, a, b, false); }
Ciphertext op_eq_ic_clear(ServerKey &sk, const FheString &a, const std::string &b) { return string_eq_ignore_case_clear(sk, a, b); }
int case_api(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, uint64_t *out, bool upper)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || (len && (need(s, "string") || need(out, "out")))) return 1;
        ServerKey &sk = hk->sk;
        const FheString r = change_case(sk, string_from(sk, s, len), upper);
        for (size_t c = 0; c < len; c++)
            for (size_t b = 0; b < BLOCKS_PER_CHAR; b++) put(sk, r[c][b], out + (c * BLOCKS_PER_CHAR + b) * sk.big_size());
        return 0;
    });
}
} // namespace

TFHE_HIP_API int tfhe_hip_string_le(tfhe_hip_host_key *hk, const uint64_t *s1, size_t len1, const uint64_t *s2, size_t len2,
                                    uint64_t *out_block)
{
    return enc_pattern_api(hk, s1, len1, s2, len2, out_block, op_le);
}
TFHE_HIP_API int tfhe_hip_string_ge(tfhe_hip_host_key *hk, const uint64_t *s1, size_t len1, const uint64_t *s2, size_t len2,
                                    uint64_t *out_block)
{
    return enc_pattern_api(hk, s1, len1, s2, len2, out_block, op_ge);
}
TFHE_HIP_API int tfhe_hip_string_le_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                                          size_t clear_len, uint64_t *out_block)
{
    return clear_pattern_api(hk, s, len, clear, clear_len, out_block, op_le_clear);
}
TFHE_HIP_API int tfhe_hip_string_ge_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                                          size_t clear_len, uint64_t *out_block)
{
    return clear_pattern_api(hk, s, len, clear, clear_len, out_block, op_ge_clear);
}
TFHE_HIP_API int tfhe_hip_string_to_uppercase(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, uint64_t *out)
{
    return case_api(hk, s, len, out, true);
}
TFHE_HIP_API int tfhe_hip_string_to_lowercase(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, uint64_t *out)
{
    return case_api(hk, s, len, out, false);
}
TFHE_HIP_API int tfhe_hip_string_eq_ignore_case(tfhe_hip_host_key *hk, const uint64_t *s1, size_t len1, const uint64_t *s2,
                                                size_t len2, uint64_t *out_block)
{
    return enc_pattern_api(hk, s1, len1, s2, len2, out_block, string_eq_ignore_case);
}
TFHE_HIP_API int tfhe_hip_string_eq_ignore_case_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len,
                                                      const uint8_t *clear, size_t clear_len, uint64_t *out_block)
{
    return clear_pattern_api(hk, s, len, clear, clear_len, out_block, op_eq_ic_clear);
}

TFHE_HIP_API int tfhe_hip_string_find(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint64_t *pattern,
                                      size_t pattern_len, int reverse, uint64_t *out_found, uint64_t *out_index)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(out_found, "out_found") || need(out_index, "out_index") || (len && need(s, "string")) ||
            (pattern_len && need(pattern, "pattern")))
            return 1;
        ServerKey &sk = hk->sk;
        const FheOptionInt r = string_find(sk, string_from(sk, s, len), string_from(sk, pattern, pattern_len), reverse != 0);
        put(sk, r.found, out_found);
        put_radix(sk, r.index, out_index);
        return 0;
    });
}
TFHE_HIP_API int tfhe_hip_string_find_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                                            size_t clear_len, int reverse, uint64_t *out_found, uint64_t *out_index)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(out_found, "out_found") || need(out_index, "out_index") || (len && need(s, "string")) ||
            (clear_len && need(clear, "pattern")))
            return 1;
        ServerKey &sk = hk->sk;
        const std::string pat(reinterpret_cast<const char *>(clear), clear_len);
        const FheOptionInt r = string_find_clear(sk, string_from(sk, s, len), pat, reverse != 0);
        put(sk, r.found, out_found);
        put_radix(sk, r.index, out_index);
        return 0;
    });
}

namespace {
int padding_of(int v, Padding *out)
{
    if (v < 0 || v > 4)
        return tfhe_hip_set_error("padding must be 0 (None), 1 (Final), 2 (Initial), 3 (InitialAndFinal) or 4 (Anywhere)");
    *out = (Padding)v;
    return 0;
}
using PaddedOp = Ciphertext (*)(ServerKey &, const FheString &, Padding, const FheString &, Padding);
int padded_api(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int pad, const uint64_t *pat, size_t pat_len, int pat_pad,
               uint64_t *out_block, PaddedOp op)
{
    return guarded(hk, [&]() -> int {
        Padding p1, p2;
        if (need(hk, "host key") || need(out_block, "out") || (len && need(s, "string")) || (pat_len && need(pat, "pattern")) ||
            padding_of(pad, &p1) || padding_of(pat_pad, &p2))
            return 1;
        ServerKey &sk = hk->sk;
        const FheString a = normalize_padding(sk, string_from(sk, s, len), p1);
        const FheString b = normalize_padding(sk, string_from(sk, pat, pat_len), p2);
        put(sk, op(sk, a, p1, b, p2), out_block);
        return 0;
    });
}
} // namespace

TFHE_HIP_API int tfhe_hip_string_eq_padded(tfhe_hip_host_key *hk, const uint64_t *s1, size_t len1, int padding1,
                                           const uint64_t *s2, size_t len2, int padding2, uint64_t *out_block)
{
    return padded_api(hk, s1, len1, padding1, s2, len2, padding2, out_block, string_eq_padded);
}
TFHE_HIP_API int tfhe_hip_string_starts_with_padded(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding,
                                                    const uint64_t *prefix, size_t prefix_len, int prefix_padding,
                                                    uint64_t *out_block)
{
    return padded_api(hk, s, len, padding, prefix, prefix_len, prefix_padding, out_block, string_starts_with_padded);
}
TFHE_HIP_API int tfhe_hip_string_contains_padded(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding,
                                                 const uint64_t *pattern, size_t pattern_len, int pattern_padding,
                                                 uint64_t *out_block)
{
    return padded_api(hk, s, len, padding, pattern, pattern_len, pattern_padding, out_block, string_contains_padded);
}
TFHE_HIP_API int tfhe_hip_string_find_padded(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding,
                                             const uint64_t *pattern, size_t pattern_len, int pattern_padding, int reverse,
                                             uint64_t *out_found, uint64_t *out_index)
{
    return guarded(hk, [&]() -> int {
        Padding p1, p2;
        if (need(hk, "host key") || need(out_found, "out_found") || need(out_index, "out_index") || (len && need(s, "string")) ||
            (pattern_len && need(pattern, "pattern")) || padding_of(padding, &p1) || padding_of(pattern_padding, &p2))
            return 1;
        ServerKey &sk = hk->sk;
        const FheString a = normalize_padding(sk, string_from(sk, s, len), p1);
        const FheString b = normalize_padding(sk, string_from(sk, pattern, pattern_len), p2);
        const FheOptionInt r = string_find_padded(sk, a, p1, b, p2, reverse != 0);
        put(sk, r.found, out_found);
        put_radix(sk, r.index, out_index);
        return 0;
    });
}

namespace {
void put_string_out(ServerKey &sk, const FheStringOut &v, uint64_t *out_content, uint64_t *out_length)
{
    for (size_t c = 0; c < v.content.size(); c++)
        for (size_t b = 0; b < BLOCKS_PER_CHAR; b++) put(sk, v.content[c][b], out_content + (c * BLOCKS_PER_CHAR + b) * sk.big_size());
    put_radix(sk, v.length, out_length);
}
} // namespace

TFHE_HIP_API int tfhe_hip_string_trim(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int mode, uint64_t *out_content,
                                      uint64_t *out_length)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(out_length, "out_length") || (len && (need(s, "string") || need(out_content, "out_content"))))
            return 1;
        if (mode < 0 || mode > 2) return tfhe_hip_set_error("trim mode must be 0 (trim_start), 1 (trim_end) or 2 (trim)");
        ServerKey &sk = hk->sk;
        const FheString str = string_from(sk, s, len);
        put_string_out(sk, mode == 0 ? string_trim_start(sk, str) : (mode == 1 ? string_trim_end(sk, str) : string_trim(sk, str)),
                       out_content, out_length);
        return 0;
    });
}
TFHE_HIP_API int tfhe_hip_string_trim_char(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, uint8_t character, int left,
                                           int right, uint64_t *out_content, uint64_t *out_length)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(out_length, "out_length") || (len && (need(s, "string") || need(out_content, "out_content"))))
            return 1;
        ServerKey &sk = hk->sk;
        put_string_out(sk, string_trim_char(sk, string_from(sk, s, len), character, left != 0, right != 0), out_content, out_length);
        return 0;
    });
}
TFHE_HIP_API int tfhe_hip_string_strip(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint64_t *pattern,
                                       size_t pattern_len, int suffix, uint64_t *out_found, uint64_t *out_content,
                                       uint64_t *out_length)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(out_found, "out_found") || need(out_length, "out_length") ||
            (len && (need(s, "string") || need(out_content, "out_content"))) || (pattern_len && need(pattern, "pattern")))
            return 1;
        ServerKey &sk = hk->sk;
        const FheOptionString r = string_strip(sk, string_from(sk, s, len), string_from(sk, pattern, pattern_len), suffix != 0);
        put(sk, r.found, out_found);
        put_string_out(sk, r.value, out_content, out_length);
        return 0;
    });
}
TFHE_HIP_API int tfhe_hip_string_strip_clear(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint8_t *clear,
                                             size_t clear_len, int suffix, uint64_t *out_found, uint64_t *out_content,
                                             uint64_t *out_length)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(out_found, "out_found") || need(out_length, "out_length") ||
            (len && (need(s, "string") || need(out_content, "out_content"))) || (clear_len && need(clear, "pattern")))
            return 1;
        ServerKey &sk = hk->sk;
        const std::string pat(reinterpret_cast<const char *>(clear), clear_len);
        const FheOptionString r = string_strip_clear(sk, string_from(sk, s, len), pat, suffix != 0);
        put(sk, r.found, out_found);
        put_string_out(sk, r.value, out_content, out_length);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_push_padding_to_end(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, uint64_t *out)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || (len && (need(s, "string") || need(out, "out")))) return 1;
        ServerKey &sk = hk->sk;
        const FheString r = push_padding_to_end(sk, string_from(sk, s, len));
        for (size_t c = 0; c < len; c++)
            for (size_t b = 0; b < BLOCKS_PER_CHAR; b++) put(sk, r[c][b], out + (c * BLOCKS_PER_CHAR + b) * sk.big_size());
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_len(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, uint64_t *out_length)
{
    return guarded(hk, [&]() -> int {
        Padding p;
        if (need(hk, "host key") || need(out_length, "out_length") || (len && need(s, "string")) || padding_of(padding, &p)) return 1;
        ServerKey &sk = hk->sk;
        put_radix(sk, string_len(sk, string_from(sk, s, len), p), out_length);
        return 0;
    });
}
TFHE_HIP_API int tfhe_hip_string_is_empty(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, uint64_t *out_block)
{
    return guarded(hk, [&]() -> int {
        Padding p;
        if (need(hk, "host key") || need(out_block, "out") || (len && need(s, "string")) || padding_of(padding, &p)) return 1;
        ServerKey &sk = hk->sk;
        put(sk, string_is_empty(sk, string_from(sk, s, len), p), out_block);
        return 0;
    });
}

static int reduce_api(tfhe_hip_host_key *hk, const uint64_t *blocks, size_t n, uint64_t *out_block, bool all)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(out_block, "out") || (n && need(blocks, "blocks"))) return 1;
        ServerKey &sk = hk->sk;
        std::vector<Ciphertext> v(n);
        for (size_t i = 0; i < n; i++) v[i] = sk.from_raw(blocks + i * sk.big_size(), 1);
        put(sk, all ? are_all_comparisons_block_true(sk, std::move(v)) : is_at_least_one_comparisons_block_true(sk, std::move(v)),
            out_block);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_boolean_any(tfhe_hip_host_key *hk, const uint64_t *blocks, size_t n, uint64_t *out_block)
{
    return reduce_api(hk, blocks, n, out_block, false);
}
TFHE_HIP_API int tfhe_hip_boolean_all(tfhe_hip_host_key *hk, const uint64_t *blocks, size_t n, uint64_t *out_block)
{
    return reduce_api(hk, blocks, n, out_block, true);
}

// ---- concatenation, repeat, substring, nth, split family, replace (host_split.hpp) --------------------------------
namespace {
void put_chars(ServerKey &sk, const FheString &s, uint64_t *out)
{
    for (size_t c = 0; c < s.size(); c++)
        for (size_t b = 0; b < BLOCKS_PER_CHAR; b++) put(sk, s[c][b], out + (c * BLOCKS_PER_CHAR + b) * sk.big_size());
}
int radix_arg(const ServerKey &sk, const uint64_t *p, RadixCiphertext *out)
{
    *out = radix_from(sk, p, sk.index_blocks(), sk.message_modulus() - 1);
    return 0;
}
// a pattern argument: exactly one of the encrypted / clear pointers (both null: the empty pattern)
struct PatternArg {
    FheString enc;
    std::string clear;
    bool is_clear = true;
    Padding padding = Padding::None;
    SplitPattern view() const
    {
        SplitPattern p;
        if (is_clear) p.clear = &clear;
        else { p.enc = &enc; p.enc_padding = padding; }
        return p;
    }
};
int pattern_arg(ServerKey &sk, const uint64_t *enc, const uint8_t *clear, size_t len, int padding, PatternArg *out)
{
    if (enc && clear) return tfhe_hip_set_error("pass the pattern either encrypted or clear, not both");
    if (len && !enc && !clear) return tfhe_hip_set_error("pattern pointer is null");
    out->is_clear = enc == nullptr;
    if (enc) {
        if (padding_of(padding, &out->padding)) return 1;
        out->enc = normalize_padding(sk, string_from(sk, enc, len), out->padding); // Initial / Anywhere -> Final
    } else {
        out->clear.assign((const char *)clear, len);
        for (unsigned char ch : out->clear)
            if (ch == 0 || ch > 127) return tfhe_hip_set_error("clear patterns are ASCII strings without NUL");
    }
    return 0;
}
int split_options(int kind, uint64_t n_clear, const RadixCiphertext *n_enc, SplitOptions *o)
{
    *o = SplitOptions();
    switch (kind) {
    case TFHE_HIP_SPLIT: break;
    case TFHE_HIP_RSPLIT: o->reverse = true; break;
    case TFHE_HIP_SPLIT_TERMINATOR: o->terminator = true; break;
    case TFHE_HIP_RSPLIT_TERMINATOR: o->terminator = o->reverse = true; break;
    case TFHE_HIP_SPLIT_INCLUSIVE: o->inclusive = true; break;
    case TFHE_HIP_SPLITN: o->limited = true; break;
    case TFHE_HIP_RSPLITN: o->limited = o->reverse = true; break;
    case TFHE_HIP_SPLIT_ASCII_WHITESPACE: break;
    default: return tfhe_hip_set_error("unknown split kind");
    }
    o->n_clear = n_clear;
    o->n_enc = o->limited ? n_enc : nullptr;
    return 0;
}
size_t max_parts_of(int kind, size_t len, size_t pattern_len, bool pattern_padded, const SplitOptions &o)
{
    if (kind == TFHE_HIP_SPLIT_ASCII_WHITESPACE) return (len + 1) / 2;
    // an encrypted pattern with padding may be as short as one char -- or empty (one more part: the "" in front)
    const bool may_be_empty = pattern_len && pattern_padded;
    if (pattern_len == 0) return len + 2;
    return split_max_parts(len, pattern_len && pattern_padded ? 1 : pattern_len, o) + (may_be_empty ? 1 : 0);
}
} // namespace

TFHE_HIP_API int tfhe_hip_string_split_max_parts(int kind, size_t len, size_t pattern_len, int pattern_padding, int n_is_encrypted,
                                                 uint64_t n_clear, size_t *out)
{
    return guarded([&]() -> int {
        if (need(out, "out")) return 1;
        SplitOptions o;
        const RadixCiphertext dummy;
        if (split_options(kind, n_clear, n_is_encrypted ? &dummy : nullptr, &o)) return 1;
        *out = max_parts_of(kind, len, pattern_len, pattern_padding != 0, o);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_split(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, const uint64_t *pattern,
                                       const uint8_t *clear_pattern, size_t pattern_len, int pattern_padding, int kind, uint64_t n_clear,
                                       const uint64_t *n_enc, uint64_t *out_parts, size_t parts_cap, uint64_t *out_number_parts,
                                       size_t *out_n_parts)
{
    return guarded(hk, [&]() -> int {
        Padding p;
        if (need(hk, "host key") || need(out_number_parts, "out_number_parts") || need(out_n_parts, "out_n_parts") ||
            (len && need(s, "string")) || padding_of(padding, &p))
            return 1;
        ServerKey &sk = hk->sk;
        PatternArg pat;
        RadixCiphertext n;
        SplitOptions o;
        if (kind != TFHE_HIP_SPLIT_ASCII_WHITESPACE && pattern_arg(sk, pattern, clear_pattern, pattern_len, pattern_padding, &pat)) return 1;
        if (n_enc) radix_arg(sk, n_enc, &n);
        if (split_options(kind, n_clear, n_enc ? &n : nullptr, &o)) return 1;
        const size_t need_parts = max_parts_of(kind, len, pattern_len, pattern && pattern_padding != 0, o);
        if (parts_cap < need_parts) return tfhe_hip_set_error("out_parts is too small (tfhe_hip_string_split_max_parts)");
        if (need_parts && len && need(out_parts, "out_parts")) return 1;
        const FheString str = normalize_padding(sk, string_from(sk, s, len), p);
        FheSplit r;
        if (kind == TFHE_HIP_SPLIT_ASCII_WHITESPACE) r = string_split_ascii_whitespace(sk, str, p);
        else if (pattern_len == 0 && (kind == TFHE_HIP_SPLIT || kind == TFHE_HIP_RSPLIT)) r = string_split_empty_pattern(sk, str, p, o.reverse);
        else r = string_split(sk, str, p, pat.view(), o); // incl. the empty pattern in the other kinds
        *out_n_parts = r.parts.size();
        for (size_t q = 0; q < r.parts.size(); q++) put_chars(sk, r.parts[q], out_parts + q * len * BLOCKS_PER_CHAR * sk.big_size());
        put_radix(sk, r.number_parts, out_number_parts);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_split_once(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, const uint64_t *pattern,
                                            const uint8_t *clear_pattern, size_t pattern_len, int pattern_padding, int reverse,
                                            uint64_t *out_found,
                                            uint64_t *out_before, uint64_t *out_after)
{
    return guarded(hk, [&]() -> int {
        Padding p;
        if (need(hk, "host key") || need(out_found, "out_found") || (len && (need(s, "string") || need(out_before, "out_before") ||
            need(out_after, "out_after"))) || padding_of(padding, &p))
            return 1;
        ServerKey &sk = hk->sk;
        PatternArg pat;
        if (pattern_arg(sk, pattern, clear_pattern, pattern_len, pattern_padding, &pat)) return 1;
        const FheString str = normalize_padding(sk, string_from(sk, s, len), p);
        const FheSplitOnce r = string_split_once(sk, str, p, pat.view(), reverse != 0);
        put(sk, r.found, out_found);
        put_chars(sk, r.before, out_before);
        put_chars(sk, r.after, out_after);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_replace_max_len(size_t len, size_t from_len, int from_padding, size_t to_len, size_t *out)
{
    if (need(out, "out")) return 1;
    // an empty pattern (known, or possibly hidden under padding) also matches in front of the string: to_len more slots
    *out = (from_padding || from_len == 0) ? to_len + len * (1 + to_len) : len * (1 + (to_len > from_len ? to_len - from_len : 0));
    return 0;
}

TFHE_HIP_API int tfhe_hip_string_replace(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, const uint64_t *from,
                                         const uint8_t *clear_from, size_t from_len, int from_padding, const uint64_t *to,
                                         const uint8_t *clear_to, size_t to_len, int limited, uint64_t n_clear, const uint64_t *n_enc, uint64_t *out_content,
                                         size_t out_cap, size_t *out_len, int *out_padding)
{
    return guarded(hk, [&]() -> int {
        Padding p;
        if (need(hk, "host key") || need(out_len, "out_len") || need(out_padding, "out_padding") || (len && need(s, "string")) ||
            padding_of(padding, &p))
            return 1;
        ServerKey &sk = hk->sk;
        PatternArg pf, pt;
        RadixCiphertext n;
        // the chars of `to` are inserted as they are: its padding zeros simply stay padding
        if (pattern_arg(sk, from, clear_from, from_len, from_padding, &pf) || pattern_arg(sk, to, clear_to, to_len, 0, &pt)) return 1;
        if (n_enc) radix_arg(sk, n_enc, &n);
        ReplaceTarget target;
        if (pt.is_clear) target.clear = &pt.clear;
        else target.enc = &pt.enc;
        const FheString str = normalize_padding(sk, string_from(sk, s, len), p);
        const PaddedString r = string_replace(sk, str, p, pf.view(), target, limited != 0, n_clear, limited && n_enc ? &n : nullptr);
        if (out_cap < r.content.size()) return tfhe_hip_set_error("out_content is too small (tfhe_hip_string_replace_max_len)");
        if (r.content.size() && need(out_content, "out_content")) return 1;
        put_chars(sk, r.content, out_content);
        *out_len = r.content.size();
        *out_padding = (int)r.padding;
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_add(tfhe_hip_host_key *hk, const uint64_t *s1, size_t len1, int padding1, const uint64_t *s2,
                                     size_t len2, int padding2, uint64_t *out_content, int *out_padding)
{
    return guarded(hk, [&]() -> int {
        Padding p1, p2;
        if (need(hk, "host key") || need(out_padding, "out_padding") || (len1 && need(s1, "s1")) || (len2 && need(s2, "s2")) ||
            ((len1 + len2) && need(out_content, "out_content")) || padding_of(padding1, &p1) || padding_of(padding2, &p2))
            return 1;
        ServerKey &sk = hk->sk;
        const PaddedString r = string_concat(string_from(sk, s1, len1), p1, string_from(sk, s2, len2), p2);
        put_chars(sk, r.content, out_content);
        *out_padding = (int)r.padding;
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_repeat(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, uint64_t n_clear,
                                        const uint64_t *n_enc, size_t n_max, uint64_t *out_content, int *out_padding)
{
    return guarded(hk, [&]() -> int {
        Padding p;
        if (need(hk, "host key") || need(out_padding, "out_padding") || (len && need(s, "string")) || padding_of(padding, &p)) return 1;
        ServerKey &sk = hk->sk;
        const FheString str = string_from(sk, s, len);
        PaddedString r;
        if (n_enc) {
            RadixCiphertext n;
            radix_arg(sk, n_enc, &n);
            r = string_repeat_encrypted(sk, str, p, n, n_max);
        } else r = string_repeat_clear(str, p, (size_t)n_clear);
        if (r.content.size() && need(out_content, "out_content")) return 1;
        put_chars(sk, r.content, out_content);
        *out_padding = (int)r.padding;
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_substring(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, uint64_t start,
                                           uint64_t end, const uint64_t *start_enc, const uint64_t *end_enc, uint64_t *out_in_range,
                                           uint64_t *out_content, int *out_padding)
{
    return guarded(hk, [&]() -> int {
        Padding p;
        if (need(hk, "host key") || need(out_in_range, "out_in_range") || need(out_padding, "out_padding") ||
            (len && need(s, "string")) || padding_of(padding, &p))
            return 1;
        if ((start_enc == nullptr) != (end_enc == nullptr)) return tfhe_hip_set_error("substring: start and end are both clear or both encrypted");
        ServerKey &sk = hk->sk;
        const FheString str = string_from(sk, s, len);
        ResultFheString r;
        if (start_enc) {
            RadixCiphertext a, b;
            radix_arg(sk, start_enc, &a);
            radix_arg(sk, end_enc, &b);
            r = string_substring_encrypted(sk, str, p, a, b);
        } else {
            if (end < start || end > len) return tfhe_hip_set_error("substring: the range is not inside the content (OutOfRange)");
            r = string_substring_clear(sk, str, p, (size_t)start, (size_t)end);
        }
        if (r.value.content.size() && need(out_content, "out_content")) return 1;
        put(sk, r.in_range, out_in_range);
        put_chars(sk, r.value.content, out_content);
        *out_padding = (int)r.value.padding;
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_nth(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int padding, uint64_t n_clear,
                                     const uint64_t *n_enc, uint64_t *out_char)
{
    return guarded(hk, [&]() -> int {
        Padding p;
        if (need(hk, "host key") || need(out_char, "out_char") || (len && need(s, "string")) || padding_of(padding, &p)) return 1;
        ServerKey &sk = hk->sk;
        const FheString str = string_from(sk, s, len);
        RadixCiphertext r;
        if (n_enc) {
            RadixCiphertext n;
            radix_arg(sk, n_enc, &n);
            r = string_nth_encrypted(sk, str, p, n);
        } else r = string_nth_clear(sk, str, p, (size_t)n_clear);
        put_radix(sk, r, out_char);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_cmux_empty(tfhe_hip_host_key *hk, const uint64_t *condition, const uint64_t *s, size_t len,
                                            int padding, uint64_t *out_content, int *out_padding)
{
    return guarded(hk, [&]() -> int {
        Padding p;
        if (need(hk, "host key") || need(condition, "condition") || need(out_padding, "out_padding") ||
            (len && (need(s, "string") || need(out_content, "out_content"))) || padding_of(padding, &p))
            return 1;
        ServerKey &sk = hk->sk;
        const PaddedString r = string_cmux_empty(sk, sk.from_raw(condition, 1), string_from(sk, s, len), p);
        put_chars(sk, r.content, out_content);
        *out_padding = (int)r.padding;
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_trim_encrypted_char(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, const uint64_t *character,
                                                     int left, int right, uint64_t *out_content, uint64_t *out_length)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(character, "character") || need(out_length, "out_length") ||
            (len && (need(s, "string") || need(out_content, "out_content"))))
            return 1;
        ServerKey &sk = hk->sk;
        const RadixCiphertext ch = radix_from(sk, character, BLOCKS_PER_CHAR, sk.message_modulus() - 1);
        put_string_out(sk, string_trim_encrypted_char(sk, string_from(sk, s, len), ch, left != 0, right != 0), out_content, out_length);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_push_padding_to_start(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, uint64_t *out)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || (len && (need(s, "string") || need(out, "out")))) return 1;
        ServerKey &sk = hk->sk;
        put_chars(sk, push_padding_to_start(sk, string_from(sk, s, len)), out);
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_string_pop_non_zero_char(tfhe_hip_host_key *hk, const uint64_t *s, size_t len, int last, uint64_t *out_char,
                                                   uint64_t *out_content)
{
    return guarded(hk, [&]() -> int {
        if (need(hk, "host key") || need(out_char, "out_char") || (len && (need(s, "string") || need(out_content, "out_content")))) return 1;
        ServerKey &sk = hk->sk;
        const PoppedChar r = pop_non_zero_char(sk, string_from(sk, s, len), last != 0);
        put_radix(sk, r.ch, out_char);
        put_chars(sk, r.content, out_content);
        return 0;
    });
}
