// oracle_backend.cpp -- TEST-ONLY stand-in for the GPU entry points that host_api.cpp / host_ops.hpp
// call, implemented over the CPU oracle, so that the host op-graph logic (level scheduling, degree
// bookkeeping, trivial shortcut, sharding) is covered by the CPU test-suite.  Linked only into
// tests/emu/libhost_emu.so; the product library never contains it.
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/tfhe_hip_strings.h"
#include "../../oracle/tfhe_oracle.h"

struct tfhe_hip_server_key {
    orc_params p;
    const uint64_t *ksk;
    const double *bsk_re, *bsk_im;
    orc_fft *fft;
    int threads;
};

static thread_local std::string g_err;

extern "C" {

int tfhe_hip_set_error(const char *msg)
{
    g_err = msg ? msg : "";
    return 1;
}
const char *tfhe_hip_last_error(void) { return g_err.c_str(); }

int tfhe_hip_host_alloc(void **out, size_t bytes)
{
    *out = std::malloc(bytes);
    return *out ? 0 : tfhe_hip_set_error("out of memory");
}
int tfhe_hip_host_free(void *p)
{
    std::free(p);
    return 0;
}

// keys stay owned by the caller (numpy arrays kept alive by the test)
tfhe_hip_server_key *emu_server_key_create(const orc_params *p, const uint64_t *ksk, const double *bsk_re,
                                           const double *bsk_im, int threads)
{
    auto *sk = new tfhe_hip_server_key{*p, ksk, bsk_re, bsk_im, orc_fft_new(p->poly_size), threads};
    return sk;
}
void emu_server_key_destroy(tfhe_hip_server_key *sk)
{
    if (sk) orc_fft_free(sk->fft);
    delete sk;
}

// small rounds, so that the level scheduler's filling is exercised by the toy-sized graphs of the CPU tests
int tfhe_hip_server_key_batch_rounds(const tfhe_hip_server_key *, size_t *latency_round, size_t *throughput_round)
{
    *latency_round = 16;
    *throughput_round = 64;
    return 0;
}

int tfhe_hip_lut_generate(const tfhe_hip_params *params, const uint64_t *table, uint64_t *glwe_out, uint64_t *max_out)
{
    orc_params p{};
    p.glwe_dim = params->glwe_dim; p.poly_size = params->poly_size;
    p.message_modulus = params->message_modulus; p.carry_modulus = params->carry_modulus;
    const uint64_t m = orc_lut_generate(&p, table, glwe_out);
    if (max_out) *max_out = m;
    return 0;
}

int tfhe_hip_ks_pbs_batch(tfhe_hip_server_key *sk, const uint64_t *lwe_in, const uint64_t *luts, size_t n_luts,
                          const uint32_t *lut_idx, uint64_t *lwe_out, size_t B)
{
    for (size_t i = 0; i < B; i++)
        if (lut_idx[i] >= n_luts) return tfhe_hip_set_error("lut_idx out of range");
    orc_ks_pbs_batch(&sk->p, sk->fft, sk->ksk, sk->bsk_re, sk->bsk_im, lwe_in, luts, lut_idx, lwe_out, B, sk->threads);
    return 0;
}
}
