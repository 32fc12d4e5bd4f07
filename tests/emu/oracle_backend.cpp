// oracle_backend.cpp -- TEST-ONLY stand-in for the GPU entry points that host_api.cpp / host_ops.hpp
// call, implemented over the CPU oracle, so that the host op-graph logic (level scheduling, degree
// bookkeeping, trivial shortcut, sharding) is covered by the CPU test-suite.  Linked only into
// tests/emu/libhost_emu.so; the product library never contains it.
#include <cstdlib>
#include <cstring>
#include <string>

#include <thread>
#include <vector>

#include "../../include/tfhe_hip_strings.h"
#include "../../oracle/tfhe_oracle.h"
#include "../../tfhe_rs_string_amd/csrc/shard.hpp"

struct tfhe_hip_server_key {
    orc_params p;
    const uint64_t *ksk;
    const double *bsk_re, *bsk_im;
    orc_fft *fft;
    int threads;
    // fake devices of a multi-device context: a batch is split over them exactly as api.cpp splits it (shard.hpp), each
    // shard bootstrapped on its own thread; `shard_log` records the shard sizes of the last call for the tests
    int n_devices = 1;
    size_t multi_min = 4;
    std::vector<size_t> shard_log;
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
void emu_server_key_set_devices(tfhe_hip_server_key *sk, int n_devices, size_t multi_min)
{
    sk->n_devices = n_devices < 1 ? 1 : n_devices;
    sk->multi_min = multi_min;
}
size_t emu_server_key_last_shards(tfhe_hip_server_key *sk, size_t *out, size_t cap)
{
    for (size_t i = 0; i < sk->shard_log.size() && i < cap; i++) out[i] = sk->shard_log[i];
    return sk->shard_log.size();
}
void emu_server_key_destroy(tfhe_hip_server_key *sk)
{
    if (sk) orc_fft_free(sk->fft);
    delete sk;
}

int tfhe_hip_server_key_get_params(const tfhe_hip_server_key *sk, tfhe_hip_params *out)
{
    const orc_params &p = sk->p;
    *out = tfhe_hip_params{p.lwe_dim, p.glwe_dim, p.poly_size, p.pbs_base_log, p.pbs_level, p.ks_base_log, p.ks_level,
                           p.message_modulus, p.carry_modulus, 0};
    return 0;
}

// small rounds, so that the level scheduler's filling is exercised by the toy-sized graphs of the CPU tests
int tfhe_hip_server_key_batch_rounds(const tfhe_hip_server_key *sk, size_t *latency_round, size_t *throughput_round)
{
    *latency_round = 16 * (size_t)sk->n_devices;
    *throughput_round = 64 * (size_t)sk->n_devices;
    if (const char *v = std::getenv("EMU_LATENCY_ROUND")) { // the real chip: 256 / 1024
        *latency_round = std::strtoull(v, nullptr, 10) * (size_t)sk->n_devices;
        *throughput_round = 4 * *latency_round;
    }
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

// "device" memory of the test backend is host memory
int tfhe_hip_device_alloc(tfhe_hip_server_key *, uint64_t **out, size_t words)
{
    *out = (uint64_t *)std::malloc(words * 8);
    return *out ? 0 : tfhe_hip_set_error("out of memory");
}
int tfhe_hip_device_free(uint64_t *p)
{
    std::free(p);
    return 0;
}
int tfhe_hip_device_upload_overlapped(tfhe_hip_server_key *, uint64_t *d_dst, const void *h_src, size_t bytes, int)
{
    std::memcpy(d_dst, h_src, bytes);
    return 0;
}
int tfhe_hip_device_upload(tfhe_hip_server_key *, uint64_t *d_dst, const void *h_src, size_t bytes)
{
    std::memcpy(d_dst, h_src, bytes);
    return 0;
}
int tfhe_hip_device_synchronize(tfhe_hip_server_key *) { return 0; }
int tfhe_hip_device_download(tfhe_hip_server_key *, void *h_dst, const uint64_t *d_src, size_t bytes)
{
    std::memcpy(h_dst, d_src, bytes);
    return 0;
}
int tfhe_hip_device_lincomb_rows(tfhe_hip_server_key *, uint64_t *d_dst, size_t rows, size_t row_words, const uint32_t *row_ptr,
                                 const uint64_t *coef, const uint64_t *const *src, const uint64_t *body)
{
    for (size_t r = 0; r < rows; r++) {
        uint64_t *out = d_dst + r * row_words;
        for (size_t w = 0; w < row_words; w++) out[w] = 0;
        for (uint32_t t = row_ptr[r]; t < row_ptr[r + 1]; t++)
            for (size_t w = 0; w < row_words; w++) out[w] += coef[t] * src[t][w];
        out[row_words - 1] += body[r];
    }
    return 0;
}
int tfhe_hip_ks_pbs_batch(tfhe_hip_server_key *sk, const uint64_t *lwe_in, const uint64_t *luts, size_t n_luts,
                          const uint32_t *lut_idx, uint64_t *lwe_out, size_t B);
int tfhe_hip_ks_pbs_batch_device(tfhe_hip_server_key *sk, const uint64_t *d_lwe_in, const uint64_t *d_luts, size_t n_luts,
                                 const uint32_t *d_lut_idx, uint64_t *d_lwe_out, size_t B, void *)
{
    return tfhe_hip_ks_pbs_batch(sk, d_lwe_in, d_luts, n_luts, d_lut_idx, d_lwe_out, B);
}

int tfhe_hip_ks_pbs_batch(tfhe_hip_server_key *sk, const uint64_t *lwe_in, const uint64_t *luts, size_t n_luts,
                          const uint32_t *lut_idx, uint64_t *lwe_out, size_t B)
{
    for (size_t i = 0; i < B; i++)
        if (lut_idx[i] >= n_luts) return tfhe_hip_set_error("lut_idx out of range");
    const size_t active = tfhe_hip::active_devices(B, (size_t)sk->n_devices, sk->multi_min);
    const size_t big = (size_t)sk->p.glwe_dim * sk->p.poly_size + 1;
    sk->shard_log.assign(active, 0);
    auto work = [&](size_t d) {
        size_t b, e;
        tfhe_hip::shard_range(B, d, active, &b, &e);
        sk->shard_log[d] = e - b;
        orc_ks_pbs_batch(&sk->p, sk->fft, sk->ksk, sk->bsk_re, sk->bsk_im, lwe_in + b * big, luts, lut_idx + b, lwe_out + b * big, e - b,
                         active > 1 ? 1 + sk->threads / (int)active : sk->threads);
    };
    std::vector<std::thread> th;
    for (size_t d = 1; d < active; d++) th.emplace_back(work, d);
    work(0);
    for (auto &t : th) t.join();
    return 0;
}
}
