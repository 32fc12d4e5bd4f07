// cabi_smoke.cpp -- the C-ABI used from native code, no Python in the loop (what a Rust / C++ host links against).
// Keys and expected results come from the CPU oracle (test infrastructure); everything under test goes through
// include/tfhe_hip.h and include/tfhe_hip_strings.h.  Build + run: tests/test_native_cabi.py (-m gpu).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "tfhe_hip.h"
#include "tfhe_hip_strings.h"
#include "tfhe_oracle.h"

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        if ((call) != 0) {                                                                           \
            std::fprintf(stderr, "%s failed: %s\n", #call, tfhe_hip_last_error());                   \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

int main()
{
    orc_params op = {24, 1, 2048, 23, 1, 3, 5, 4, 4, 7.069849454709433e-6, 2.9403601535432533e-16};
    tfhe_hip_params p = {op.lwe_dim, op.glwe_dim, op.poly_size, op.pbs_base_log, op.pbs_level, op.ks_base_log, op.ks_level, 4, 4, 0};
    const size_t big = (size_t)op.glwe_dim * op.poly_size + 1, lut_len = (size_t)(op.glwe_dim + 1) * op.poly_size;
    orc_rng rng;
    orc_rng_seed(&rng, 0xC0FFEE);
    std::vector<uint64_t> glwe_sk(op.glwe_dim * op.poly_size), lwe_sk(op.lwe_dim), bsk(orc_bsk_len(&op)), ksk(orc_ksk_len(&op));
    orc_gen_binary_key(&rng, glwe_sk.data(), glwe_sk.size());
    orc_gen_binary_key(&rng, lwe_sk.data(), lwe_sk.size());
    orc_gen_bsk(&rng, &op, lwe_sk.data(), glwe_sk.data(), bsk.data());
    orc_gen_ksk(&rng, &op, glwe_sk.data(), lwe_sk.data(), ksk.data());

    tfhe_hip_context *ctx = nullptr;
    tfhe_hip_server_key *sk = nullptr;
    const int device_ids[1] = {0};
    CHECK(tfhe_hip_context_create(&ctx, device_ids, 1));
    CHECK(tfhe_hip_server_key_create(ctx, &sk, &p, bsk.data(), bsk.size(), ksk.data(), ksk.size()));

    // 1. the hot path: 20 KS-PBS with f(x) = x^2 mod 16, bit-compared with the oracle
    const size_t B = 20;
    std::vector<uint64_t> cts(B * big), out(B * big), want(B * big), lut(lut_len), table(16);
    for (uint64_t i = 0; i < 16; i++) table[i] = (i * i) % 16;
    uint64_t degree = 0;
    CHECK(tfhe_hip_lut_generate(&p, table.data(), lut.data(), &degree));
    for (size_t i = 0; i < B; i++)
        orc_lwe_encrypt(&rng, glwe_sk.data(), (uint32_t)big - 1, op.glwe_std, orc_shortint_encode(&op, i % 16), cts.data() + i * big);
    std::vector<uint32_t> idx(B, 0);
    CHECK(tfhe_hip_ks_pbs_batch(sk, cts.data(), lut.data(), 1, idx.data(), out.data(), B));
    orc_fft *fft = orc_fft_new(op.poly_size);
    std::vector<double> bre(bsk.size() / 2), bim(bsk.size() / 2);
    orc_bsk_to_fourier(&op, bsk.data(), bre.data(), bim.data());
    orc_ks_pbs_batch(&op, fft, ksk.data(), bre.data(), bim.data(), cts.data(), lut.data(), idx.data(), want.data(), B, 4);
    if (std::memcmp(out.data(), want.data(), out.size() * 8) != 0) { std::fprintf(stderr, "KS-PBS differs from the oracle\n"); return 1; }
    for (size_t i = 0; i < B; i++) {
        const uint64_t m = orc_shortint_decode(&op, orc_lwe_decrypt(glwe_sk.data(), (uint32_t)big - 1, out.data() + i * big));
        if (m != ((i % 16) * (i % 16)) % 16) { std::fprintf(stderr, "wrong decryption at %zu\n", i); return 1; }
    }

    // 2. a caller: FheString::contains through the op-graph layer
    tfhe_hip_host_key *hk = nullptr;
    CHECK(tfhe_hip_host_key_create(sk, &p, &hk));
    auto encrypt_str = [&](const std::string &s) {
        std::vector<uint64_t> v(s.size() * 4 * big);
        for (size_t c = 0; c < s.size(); c++)
            for (int b = 0; b < 4; b++)
                orc_lwe_encrypt(&rng, glwe_sk.data(), (uint32_t)big - 1, op.glwe_std,
                                orc_shortint_encode(&op, ((unsigned char)s[c] >> (2 * b)) & 3), v.data() + (c * 4 + b) * big);
        return v;
    };
    const std::string hay = "native callers work too", needle = "work";
    std::vector<uint64_t> eh = encrypt_str(hay), en = encrypt_str(needle), res(big), res2(big);
    CHECK(tfhe_hip_string_contains(hk, eh.data(), hay.size(), en.data(), needle.size(), res.data()));
    CHECK(tfhe_hip_string_contains_clear(hk, eh.data(), hay.size(), (const uint8_t *)"walk", 4, res2.data()));
    const uint64_t r1 = orc_shortint_decode(&op, orc_lwe_decrypt(glwe_sk.data(), (uint32_t)big - 1, res.data()));
    const uint64_t r2 = orc_shortint_decode(&op, orc_lwe_decrypt(glwe_sk.data(), (uint32_t)big - 1, res2.data()));
    if (r1 != 1 || r2 != 0) { std::fprintf(stderr, "contains: got %llu / %llu\n", (unsigned long long)r1, (unsigned long long)r2); return 1; }

    // 3. error convention: int return + thread-local message
    if (tfhe_hip_ks_pbs_batch(sk, nullptr, lut.data(), 1, idx.data(), out.data(), B) != 1 || !std::strlen(tfhe_hip_last_error())) {
        std::fprintf(stderr, "null input not rejected\n");
        return 1;
    }
    tfhe_hip_host_key_destroy(hk);
    tfhe_hip_server_key_destroy(sk);
    tfhe_hip_context_destroy(ctx);
    orc_fft_free(fft);
    std::printf("cabi_smoke ok\n");
    return 0;
}
