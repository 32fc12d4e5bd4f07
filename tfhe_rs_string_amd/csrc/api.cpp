// api.cpp -- implementation of the C-ABI declared in include/tfhe_hip.h.
//
// Host-side logic only: parameter checks (the reference's assert!s become error returns),
// device memory ownership, per-(key, stream) workspaces, kernel sequencing, HIP-event profiling.
// There is NO CPU fallback: without a HIP device every compute entry point fails loudly.
#include "../../include/tfhe_hip.h"

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <thread>

#include "kernels.hpp"
#include "shard.hpp"
#include "wave_tables.hpp"

#define TFHE_HIP_API extern "C" __attribute__((visibility("default")))

extern "C" void tfhe_hip_make_tables(uint32_t n, double *tw, double *w);

namespace {

thread_local std::string g_last_error;

int fail(const std::string &msg)
{
    g_last_error = msg;
    return 1;
}

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return fail(std::string(#expr) + " failed: " + hipGetErrorString(e_) + " (" + __FILE__ + ":" +     \
                        std::to_string(__LINE__) + ")");                                                       \
    } while (0)

} // namespace

// shared with host_api.cpp
extern "C" __attribute__((visibility("hidden"))) int tfhe_hip_set_error(const char *msg) { return fail(msg ? msg : ""); }

namespace {

bool aligned8(const void *p) { return p != nullptr && (reinterpret_cast<uintptr_t>(p) % 8) == 0; }

} // namespace

struct tfhe_hip_context {
    int device = 0;   // the PRIMARY device: device-buffer entry points take pointers / streams of this one
    int cu_count = 0;
    std::string arch;
    std::vector<tfhe_hip_context *> subs; // one single-device context per further device of a multi-device context (owned)
};

namespace {

struct Workspace {
    size_t cap_B = 0;       // ciphertexts d_small is sized for
    size_t cap_partial = 0; // (split, ciphertext) pairs d_partial is sized for
    uint64_t *d_small = nullptr;   // B x (n+1)
    uint64_t *d_big = nullptr;     // B x (kN+1)  (BootstrapKeyswitch order only)
    size_t cap_big = 0;
    uint64_t *d_partial = nullptr; // splits x B x (n+1)
    void *d_digits = nullptr;      // matrix-core keyswitch: the digits of a batch in fragment order, then the digit sums
    size_t cap_digits = 0;         // ... in bytes
    // staging for the host-buffer entry points
    uint64_t *d_in = nullptr, *d_out = nullptr, *d_luts = nullptr;
    uint32_t *d_idx = nullptr;
    size_t cap_in = 0, cap_out = 0, cap_luts = 0, cap_idx = 0;
    // pinned staging for small host-buffer calls (pageable hipMemcpyAsync stalls for tens of ms when a
    // small transfer follows a multi-MB one; measured on ROCm 7.2)
    char *h_pin = nullptr;
    size_t cap_pin = 0;
    uint64_t last_use = 0; // the key's use counter at the last device-buffer call on this stream (eviction order)
};

struct EventPair {
    hipEvent_t a, b;
    int kernel;
};

} // namespace

struct tfhe_hip_server_key {
    tfhe_hip_context *ctx = nullptr;
    tfhe_hip_params params{};
    tfhe_hip::DevParams dp{};
    uint64_t *d_ksk = nullptr, *d_colsum = nullptr; // d_ksk is released once the planes are built
    uint32_t *d_ksk_planes = nullptr;               // byte-plane layout read by the dot4 keyswitch kernel
    void *d_ksk_tiles = nullptr;                    // MFMA fragment order read by the matrix-core keyswitch (the default)
    size_t ks_mfma_min = 1;                         // batches from this size on take the matrix-core keyswitch
    double *d_bsk_f = nullptr, *d_twist = nullptr, *d_w = nullptr;
    double *d_bsk_w = nullptr; // lane-ordered copy of the Fourier BSK for the wave kernel (variant 1)
    char *d_gtab = nullptr;    // lane-ordered twiddle tables of the wave kernel
    double wconsts[8] = {0};   // W[64], W[128], W[192], W[384] (kernel-argument constants)
    double *d_bsk_p = nullptr; // [i][column][slot][row][lane] copy for the wave-pair kernel (variant 2)
    char *d_gtab_p = nullptr;  // its twiddle tables
    double *d_bsk_b = nullptr; // [i][column][wavefront][row, t][lane] copy for the workgroup kernel (variant 3)
    double *d_bsk_q = nullptr; // [i][column][wavefront][reg][row][lane] copy for the quad kernel (variant 5)
    size_t block_max_batch = 0; // automatic choice: batches up to this size take variant 3, larger ones variant 2
    int grouping_factor = 0;    // 0: classic PBS; 2 / 3: multi-bit PBS (LweBskGroupingFactor)
    double *d_bsk_mb = nullptr; // multi-bit: [group][2^g][column][wavefront][row, t][lane]
    size_t bsk_polys = 0;
    int variant = 0;           // 0: workgroup per ciphertext (LDS FFT), 1: wave per ciphertext, 2: wave pair per ciphertext,
                               // 3: 512-thread workgroup per ciphertext (latency), -1: 3 for small batches, else 2
    std::mutex mu;
    std::map<hipStream_t, Workspace> ws;
    uint64_t ws_clock = 0;
    size_t ws_limit = 8; // workspaces of CALLER streams kept at once (TFHE_HIP_MAX_STREAM_WS); the least recently used goes first
    hipStream_t pipe_stream[2] = {nullptr, nullptr}; // large host batches from pinned buffers (run_host_pipelined)
    std::mutex host_mu; // the host entry points share one staging workspace: one call at a time per key
    // scratch of tfhe_hip_device_lincomb_rows (CSR of the terms), guarded by host_mu
    char *d_csr = nullptr;
    size_t cap_csr = 0;
    // Multi-device context: full replicas of the key on the context's further devices.  A batch is split contiguously over
    // the devices (shard.hpp); every replica works on its own stream with its own transfer buffers.
    struct PeerIo {
        hipStream_t stream = nullptr;
        hipEvent_t ready = nullptr, done = nullptr;
        uint64_t *d_in = nullptr, *d_out = nullptr, *d_luts = nullptr;
        uint32_t *d_idx = nullptr;
        size_t cap_in = 0, cap_out = 0, cap_luts = 0, cap_idx = 0;
    };
    std::vector<tfhe_hip_server_key *> peers;
    std::vector<PeerIo> peer_io;
    size_t multi_min = 64; // a device takes part in a batch from this many ciphertexts on (TFHE_HIP_MULTI_MIN)
    // device-visible error word (page-locked, mapped): kernels OR in tfhe_hip::ERR_* bits, check_device_status() reads it
    uint32_t *h_err = nullptr, *d_err = nullptr;
    int debug_skip_signal = -1; // tfhe_hip_debug_skip_pair_signal
    // transfer stream of tfhe_hip_device_upload_overlapped (created on first use)
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_after = nullptr, copy_done = nullptr;
    bool profiling = false;
    std::vector<EventPair> pending;
    double total_ms[2] = {0, 0};
    uint64_t launches[2] = {0, 0};
};

namespace {

void free_ws(Workspace &w)
{
    if (w.h_pin) hipHostFree(w.h_pin);
    hipFree(w.d_small); hipFree(w.d_big); hipFree(w.d_partial); hipFree(w.d_digits); hipFree(w.d_in); hipFree(w.d_out); hipFree(w.d_luts); hipFree(w.d_idx);
    w = Workspace{};
}

// The workspace of a device-buffer call on the caller's stream `s` (sk->mu held, and held by the caller until its enqueue is
// complete).  A host that creates a stream per request would otherwise leave ~20 MB behind per stream: beyond ws_limit caller
// streams the least recently used workspace is released (hipFree waits for the work that still uses it; the stream handle
// itself is never touched -- it may be gone).  The key's own streams (null, pipeline, replicas) are not counted or evicted.
Workspace &caller_workspace(tfhe_hip_server_key *sk, hipStream_t s)
{
    Workspace &w = sk->ws[s];
    w.last_use = ++sk->ws_clock;
    if (s == nullptr || s == sk->pipe_stream[0] || s == sk->pipe_stream[1]) return w;
    for (;;) {
        size_t callers = 0;
        auto oldest = sk->ws.end();
        for (auto it = sk->ws.begin(); it != sk->ws.end(); ++it) {
            if (it->first == nullptr || it->first == sk->pipe_stream[0] || it->first == sk->pipe_stream[1]) continue;
            callers++;
            if (it->first != s && (oldest == sk->ws.end() || it->second.last_use < oldest->second.last_use)) oldest = it;
        }
        if (callers <= sk->ws_limit || oldest == sk->ws.end()) return w;
        free_ws(oldest->second);
        sk->ws.erase(oldest);
    }
}

int check_params(const tfhe_hip_params *p)
{
    if (!p) return fail("params pointer is null");
    if (p->pbs_order > 1) return fail("pbs_order must be 0 (KeyswitchBootstrap) or 1 (BootstrapKeyswitch)");
    if (p->glwe_dim < 1 || p->glwe_dim > 5 || p->poly_size < 256 || p->poly_size > 4096 || (p->poly_size & (p->poly_size - 1)) ||
        (uint64_t)(p->glwe_dim + 1) * p->poly_size > 10240)
        return fail("kernels are built for glwe_dim 1 ... 5 and a power-of-two poly_size from 256 to 4096 with (glwe_dim + 1) * poly_size "
                    "<= 10240 (glwe_dim = 1, N = 2048 with one PBS level of base 2^23 -- PARAM_MESSAGE_2_CARRY_2_KS_PBS -- takes the wave / "
                    "workgroup kernels, anything else the generic one)");
    if (p->lwe_dim == 0) return fail("lwe_dim must be > 0");
    if (p->pbs_level == 0 || p->pbs_base_log == 0 || p->pbs_level * p->pbs_base_log > 63 || p->pbs_base_log > 31)
        return fail("unsupported pbs decomposition");
    if (!((p->ks_level == 5 && p->ks_base_log == 3) || (p->ks_level == 4 && p->ks_base_log == 4) ||
          (p->ks_level == 3 && p->ks_base_log == 5) || (p->ks_level == 2 && p->ks_base_log == 7) ||
          (p->ks_level == 2 && p->ks_base_log == 5) || (p->ks_level == 3 && p->ks_base_log == 4) ||
          (p->ks_level == 4 && p->ks_base_log == 3) || (p->ks_level == 6 && p->ks_base_log == 3)))
        return fail("keyswitch kernel is built for (ks_level, ks_base_log) = (5, 3), (4, 4), (3, 5), (2, 7), (2, 5), (3, 4), (4, 3) or (6, 3)");
    if (p->message_modulus == 0 || p->carry_modulus == 0 || p->poly_size % (p->message_modulus * p->carry_modulus))
        return fail("message_modulus * carry_modulus must divide poly_size");
    return 0;
}

// A kernel of an earlier launch on this key could not produce a valid result (device_math.hpp: ERR_*).  Sticky: every
// later call fails too, until the caller has read and cleared the word with tfhe_hip_server_key_device_status.
int check_device_status(const tfhe_hip_server_key *sk)
{
    if (!sk->h_err) return 0;
    uint32_t v = __atomic_load_n(sk->h_err, __ATOMIC_ACQUIRE);
    for (const tfhe_hip_server_key *peer : sk->peers)
        if (peer->h_err) v |= __atomic_load_n(peer->h_err, __ATOMIC_ACQUIRE);
    if (v == 0) return 0;
    std::string msg = "device error word " + std::to_string(v) + ":";
    if (v & 1u) msg += " a wave pair of the throughput kernel timed out waiting for its partner (results of that launch are invalid);";
    if (v & 2u) msg += " lut_idx out of range in a device-buffer call (table 0 was used);";
    return fail(msg + " clear with tfhe_hip_server_key_device_status");
}

bool trace_on()
{
    static const bool on = std::getenv("TFHE_HIP_TRACE") != nullptr;
    return on;
}

template <typename T> int grow(T **ptr, size_t *cap, size_t need_elems)
{
    if (*cap >= need_elems) return 0;
    if (trace_on()) std::fprintf(stderr, "[tfhe_hip] grow %zu -> %zu elems of %zu B\n", *cap, need_elems, sizeof(T));
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc((void **)ptr, need_elems * sizeof(T)));
    *cap = need_elems;
    return 0;
}

// make sure the (key, stream) workspace can hold B ciphertexts with `splits` keyswitch splits
int ensure_ws(tfhe_hip_server_key *sk, Workspace &w, size_t B, int splits)
{
    const size_t small = sk->dp.n + 1;
    // the two buffers grow independently: a small batch with many splits needs no more partial sums than a
    // large one with few (re-allocating synchronises the device, so it must not happen between the levels of an op)
    const size_t pairs = B * (size_t)splits;
    if (w.cap_B < B) {
        if (trace_on()) std::fprintf(stderr, "[tfhe_hip] workspace small LWEs %zu -> %zu\n", w.cap_B, B);
        if (w.d_small) HIP_TRY(hipFree(w.d_small));
        w.d_small = nullptr;
        w.cap_B = 0;
        HIP_TRY(hipMalloc((void **)&w.d_small, B * small * sizeof(uint64_t)));
        w.cap_B = B;
    }
    if (w.cap_partial < pairs) {
        const size_t want = pairs < 4096 ? 4096 : pairs; // covers every batch up to 256 x 16 splits at once
        if (trace_on()) std::fprintf(stderr, "[tfhe_hip] workspace partial sums %zu -> %zu (B %zu x %d splits)\n", w.cap_partial, want, B, splits);
        if (w.d_partial) HIP_TRY(hipFree(w.d_partial));
        w.d_partial = nullptr;
        w.cap_partial = 0;
        HIP_TRY(hipMalloc((void **)&w.d_partial, want * small * sizeof(uint64_t)));
        w.cap_partial = want;
    }
    if (sk->d_ksk_tiles) {
        const size_t need = tfhe_hip::keyswitch_mfma_digit_bytes(sk->dp, B) + (B + 31) / 32 * 32 * sizeof(int);
        if (w.cap_digits < need) {
            const size_t floor_b = tfhe_hip::keyswitch_mfma_digit_bytes(sk->dp, 256) + 256 * sizeof(int);
            const size_t want = std::max(need, floor_b);
            if (w.d_digits) HIP_TRY(hipFree(w.d_digits));
            w.d_digits = nullptr;
            w.cap_digits = 0;
            HIP_TRY(hipMalloc(&w.d_digits, want));
            w.cap_digits = want;
        }
    }
    return 0;
}

int record(tfhe_hip_server_key *sk, hipStream_t s, hipEvent_t *ev)
{
    HIP_TRY(hipEventCreate(ev));
    HIP_TRY(hipEventRecord(*ev, s));
    return 0;
}

// enqueue keyswitch (d_in big -> d_small_out)
int enqueue_keyswitch(tfhe_hip_server_key *sk, Workspace &w, const uint64_t *d_in, uint64_t *d_small_out, size_t B,
                      int splits, hipStream_t s)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (sk->profiling && record(sk, s, &e0)) return 1;
    if (sk->d_ksk_tiles && B >= sk->ks_mfma_min) {
        int *d_sums = reinterpret_cast<int *>(static_cast<char *>(w.d_digits) + tfhe_hip::keyswitch_mfma_digit_bytes(sk->dp, B));
        HIP_TRY(tfhe_hip::launch_keyswitch_mfma(sk->dp, d_in, sk->d_ksk_tiles, w.d_digits, d_sums, w.d_partial, d_small_out, B,
                                                tfhe_hip::keyswitch_mfma_splits_for(sk->dp, B, sk->ctx->cu_count), s));
    } else {
        HIP_TRY(tfhe_hip::launch_keyswitch(sk->dp, d_in, sk->d_ksk_planes, w.d_partial, B, splits, s));
        HIP_TRY(tfhe_hip::launch_keyswitch_finalize(sk->dp, d_in, sk->d_colsum, w.d_partial, d_small_out, B, splits, s));
    }
    if (sk->profiling) {
        if (record(sk, s, &e1)) return 1;
        sk->pending.push_back({e0, e1, 0});
    }
    return 0;
}

// the Fourier key in oracle order / lane order, for the kernels and entry points outside the default path (sk->mu held
// or single-threaded use: key introspection)
int ensure_bsk_f(tfhe_hip_server_key *sk)
{
    if (sk->d_bsk_f) return 0;
    if (!sk->d_bsk_p) return fail("this key has no Fourier bootstrap key to rebuild the oracle-order copy from");
    const size_t bytes = sk->bsk_polys * (size_t)sk->dp.N * 8;
    double *fresh = nullptr; // published only once the rebuild has succeeded: a failed one must not leave a garbage key behind
    HIP_TRY(hipMalloc((void **)&fresh, bytes));
    if (tfhe_hip::launch_bsk_unpermute_pair(sk->dp, sk->d_bsk_p, fresh, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
        (void)hipGetLastError();
        hipFree(fresh);
        return fail("rebuilding the oracle-order Fourier bootstrap key failed");
    }
    sk->d_bsk_f = fresh;
    return 0;
}
int ensure_bsk_w(tfhe_hip_server_key *sk)
{
    if (sk->d_bsk_w) return 0;
    if (ensure_bsk_f(sk)) return 1;
    double *fresh = nullptr;
    HIP_TRY(hipMalloc((void **)&fresh, sk->bsk_polys * (size_t)sk->dp.N * 8));
    if (tfhe_hip::launch_bsk_permute_wave(sk->dp, sk->d_bsk_f, fresh, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
        (void)hipGetLastError();
        hipFree(fresh);
        return fail("building the lane-order Fourier bootstrap key failed");
    }
    sk->d_bsk_w = fresh;
    return 0;
}

int ensure_bsk_q(tfhe_hip_server_key *sk) // (built with the key for the parameter sets of the default path)
{
    if (sk->d_bsk_q) return 0;
    if (ensure_bsk_f(sk)) return 1;
    double *fresh = nullptr;
    HIP_TRY(hipMalloc((void **)&fresh, sk->bsk_polys * (size_t)sk->dp.N * 8));
    if (tfhe_hip::launch_bsk_permute_quad(sk->dp, sk->d_bsk_f, fresh, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
        (void)hipGetLastError();
        hipFree(fresh);
        return fail("building the quad-order Fourier bootstrap key failed");
    }
    sk->d_bsk_q = fresh;
    return 0;
}

int enqueue_blind_rotate(tfhe_hip_server_key *sk, const uint64_t *d_small, const uint64_t *d_luts, size_t n_luts,
                         const uint32_t *d_idx, uint64_t *d_out, size_t B, hipStream_t s)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    tfhe_hip::BrCtl ctl;
    ctl.err = sk->d_err;
    ctl.n_luts = (uint32_t)std::min<size_t>(n_luts, 0xFFFFFFFFu);
    ctl.skip_signal_step = sk->debug_skip_signal;
    if (sk->profiling && record(sk, s, &e0)) return 1;
    int variant = sk->variant;
    if (variant < 0 && !sk->grouping_factor) {
        // up to one ciphertext per CU: the latency kernel (2.66 ms); up to two per CU: the quad kernel, 256-thread
        // workgroups, two per CU (4.5 ms against 5.3 for two rounds); up to three: that plus one round of the latency
        // kernel (7.2 ms); beyond: whole rounds of the wave-pair kernel (4 per CU, 8.1 ms) plus a remainder handled the same way
        const size_t cus = (size_t)sk->ctx->cu_count, round = 4 * cus, small = (size_t)sk->dp.n + 1, big = (size_t)sk->dp.k * sk->dp.N + 1;
        auto tail = [&](size_t first, size_t count) -> int { // count <= 3 cus
            if (count == 0) return 0;
            size_t two = 0; // ciphertexts that go two per CU
            if (count > cus) two = std::min(count, 2 * cus);
            if (two)
                HIP_TRY(tfhe_hip::launch_blind_rotate_quad(sk->dp, d_small + first * small, d_luts, d_idx + first, sk->d_bsk_q, sk->d_twist,
                                                           sk->d_w, d_out + first * big, two, ctl, s));
            if (count > two)
                HIP_TRY(tfhe_hip::launch_blind_rotate_block(sk->dp, d_small + (first + two) * small, d_luts, d_idx + first + two, sk->d_bsk_b,
                                                            sk->d_twist, sk->d_w, d_out + (first + two) * big, count - two, ctl, s));
            return 0;
        };
        const size_t rem = B % round;
        if (B <= sk->block_max_batch) {
            if (tail(0, B)) return 1;
        } else if (rem != 0 && rem <= sk->block_max_batch) {
            const size_t head = B - rem;
            HIP_TRY(tfhe_hip::launch_blind_rotate_pair(sk->dp, d_small, d_luts, d_idx, sk->d_bsk_p, sk->d_gtab_p, sk->wconsts, d_out, head, ctl, s));
            if (tail(head, rem)) return 1;
        } else {
            HIP_TRY(tfhe_hip::launch_blind_rotate_pair(sk->dp, d_small, d_luts, d_idx, sk->d_bsk_p, sk->d_gtab_p, sk->wconsts, d_out, B, ctl, s));
        }
        variant = -2; // done
    }
    if (sk->grouping_factor) variant = 100;
    switch (variant) {
    case -2:
        break;
    case 100:
        HIP_TRY(tfhe_hip::launch_blind_rotate_block_mb(sk->dp, sk->grouping_factor, d_small, d_luts, d_idx, sk->d_bsk_mb, sk->d_twist,
                                                       sk->d_w, d_out, B, ctl, s));
        break;
    case 3:
        HIP_TRY(tfhe_hip::launch_blind_rotate_block(sk->dp, d_small, d_luts, d_idx, sk->d_bsk_b, sk->d_twist, sk->d_w, d_out, B, ctl, s));
        break;
    case 4:
        HIP_TRY(tfhe_hip::launch_blind_rotate_half(sk->dp, d_small, d_luts, d_idx, sk->d_bsk_b, sk->d_twist, sk->d_w, d_out, B, ctl, s));
        break;
    case 5:
        if (ensure_bsk_q(sk)) return 1;
        HIP_TRY(tfhe_hip::launch_blind_rotate_quad(sk->dp, d_small, d_luts, d_idx, sk->d_bsk_q, sk->d_twist, sk->d_w, d_out, B, ctl, s));
        break;
    case 2:
        HIP_TRY(tfhe_hip::launch_blind_rotate_pair(sk->dp, d_small, d_luts, d_idx, sk->d_bsk_p, sk->d_gtab_p, sk->wconsts, d_out, B, ctl, s));
        break;
    case 1:
        if (ensure_bsk_w(sk)) return 1;
        HIP_TRY(tfhe_hip::launch_blind_rotate_wave(sk->dp, d_small, d_luts, d_idx, sk->d_bsk_w, sk->d_gtab, sk->wconsts, d_out, B, ctl, s));
        break;
    default:
        if (ensure_bsk_f(sk)) return 1;
        HIP_TRY(tfhe_hip::launch_blind_rotate_v0(sk->dp, d_small, d_luts, d_idx, sk->d_bsk_f, sk->d_twist, sk->d_w,
                                                 d_out, B, ctl, s));
    }
    if (sk->profiling) {
        if (record(sk, s, &e1)) return 1;
        sk->pending.push_back({e0, e1, 1});
    }
    return 0;
}

enum class Op { KS, PBS, KS_PBS, PBS_KS };

int run_device_one(tfhe_hip_server_key *sk, Op op, const uint64_t *d_in, const uint64_t *d_luts, size_t n_luts,
                   const uint32_t *d_idx, uint64_t *d_out, size_t B, hipStream_t s)
{
    if (B == 0) return 0;
    if (check_device_status(sk)) return 1;
    HIP_TRY(hipSetDevice(sk->ctx->device));
    std::lock_guard<std::mutex> lock(sk->mu);
    Workspace &w = caller_workspace(sk, s);
    const int splits = tfhe_hip::ks_splits_for(sk->dp, B, sk->ctx->cu_count);
    if (ensure_ws(sk, w, B, std::max(splits, tfhe_hip::keyswitch_mfma_splits_for(sk->dp, B, sk->ctx->cu_count)))) return 1;
    if (op == Op::KS) return enqueue_keyswitch(sk, w, d_in, d_out, B, splits, s);
    if (op == Op::PBS) return enqueue_blind_rotate(sk, d_in, d_luts, n_luts, d_idx, d_out, B, s);
    if (op == Op::PBS_KS) { // programmable_bootstrap_keyswitch_assign, shortint/server_key/mod.rs:859-932
        if (grow(&w.d_big, &w.cap_big, B * ((size_t)sk->dp.k * sk->dp.N + 1))) return 1;
        if (enqueue_blind_rotate(sk, d_in, d_luts, n_luts, d_idx, w.d_big, B, s)) return 1;
        return enqueue_keyswitch(sk, w, w.d_big, d_out, B, splits, s);
    }
    if (enqueue_keyswitch(sk, w, d_in, w.d_small, B, splits, s)) return 1;
    return enqueue_blind_rotate(sk, w.d_small, d_luts, n_luts, d_idx, d_out, B, s);
}

int copy_between(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t s)
{
    if (dst_dev == src_dev) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    else HIP_TRY(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, s));
    return 0;
}

// Device-buffer entry points on a multi-device key: the pointers and the stream belong to the PRIMARY device.  The batch is
// split contiguously (shard.hpp); shard 0 runs in place, every further shard is copied to its device behind an event on
// the caller's stream, bootstrapped there on the replica's own stream and copied back into the caller's output rows; the
// caller's stream then waits for those copies.  Nothing here blocks the host.
int run_device(tfhe_hip_server_key *sk, Op op, const uint64_t *d_in, const uint64_t *d_luts, size_t n_luts,
               const uint32_t *d_idx, uint64_t *d_out, size_t B, hipStream_t s)
{
    const size_t active = tfhe_hip::active_devices(B, 1 + sk->peers.size(), sk->multi_min);
    if (active <= 1) return run_device_one(sk, op, d_in, d_luts, n_luts, d_idx, d_out, B, s);
    if (check_device_status(sk)) return 1;
    const size_t big = (size_t)sk->dp.k * sk->dp.N + 1, small = sk->dp.n + 1, lut_len = (size_t)(sk->dp.k + 1) * sk->dp.N;
    const size_t in_w = (op == Op::PBS || op == Op::PBS_KS) ? small : big, out_w = (op == Op::KS || op == Op::PBS_KS) ? small : big;
    const int dev0 = sk->ctx->device;
    // host_mu: the replicas' transfer buffers and streams belong to the key, so the ENQUEUE sequences of two sharded calls must
    // not interleave (copy in / kernels / copy out of one call are ordered on io.stream only as a block).  Everything under the
    // lock is an asynchronous enqueue (microseconds; a buffer grows once): the calls of concurrent host threads execute
    // concurrently on the devices, they only queue one after the other.
    std::lock_guard<std::mutex> multi_lock(sk->host_mu);
    size_t recorded = 0; // peers whose `done` event is on its stream
    auto enqueue_peers = [&]() -> int {
        for (size_t d = 1; d < active; d++) {
            tfhe_hip_server_key *peer = sk->peers[d - 1];
            tfhe_hip_server_key::PeerIo &io = sk->peer_io[d - 1];
            const int dev = peer->ctx->device;
            size_t b, e;
            tfhe_hip::shard_range(B, d, active, &b, &e);
            const size_t nb = e - b;
            HIP_TRY(hipSetDevice(dev0));
            HIP_TRY(hipEventRecord(io.ready, s)); // the caller's inputs are complete
            HIP_TRY(hipSetDevice(dev));
            HIP_TRY(hipStreamWaitEvent(io.stream, io.ready, 0));
            if (grow(&io.d_in, &io.cap_in, nb * in_w) || grow(&io.d_out, &io.cap_out, nb * out_w)) return 1;
            if (copy_between(io.d_in, dev, d_in + b * in_w, dev0, nb * in_w * 8, io.stream)) return 1;
            if (op != Op::KS) {
                if (grow(&io.d_luts, &io.cap_luts, n_luts * lut_len) || grow(&io.d_idx, &io.cap_idx, nb)) return 1;
                if (copy_between(io.d_luts, dev, d_luts, dev0, n_luts * lut_len * 8, io.stream)) return 1;
                if (copy_between(io.d_idx, dev, d_idx + b, dev0, nb * 4, io.stream)) return 1;
            }
            if (run_device_one(peer, op, io.d_in, io.d_luts, n_luts, io.d_idx, io.d_out, nb, io.stream)) return 1;
            if (copy_between(d_out + b * out_w, dev0, io.d_out, dev, nb * out_w * 8, io.stream)) return 1;
            HIP_TRY(hipEventRecord(io.done, io.stream));
            recorded = d;
        }
        HIP_TRY(hipSetDevice(dev0));
        size_t b0, e0;
        tfhe_hip::shard_range(B, 0, active, &b0, &e0);
        return run_device_one(sk, op, d_in, d_luts, n_luts, d_idx, d_out, e0, s);
    };
    const int rc = enqueue_peers();
    // Success or failure: the caller's stream waits for every copy into its output that is already in flight, a peer that
    // failed half way is drained, and the current device is the primary one again.
    int rc_tail = 0;
    if (hipSetDevice(dev0) != hipSuccess) rc_tail = fail("hipSetDevice(primary) failed");
    for (size_t d = 1; d <= recorded; d++)
        if (hipStreamWaitEvent(s, sk->peer_io[d - 1].done, 0) != hipSuccess) rc_tail = fail("hipStreamWaitEvent on a replica's results failed");
    if (rc && recorded + 1 < active) (void)hipStreamSynchronize(sk->peer_io[recorded].stream);
    return rc ? rc : rc_tail;
}

int check_batch_ptrs(const tfhe_hip_server_key *sk, const void *in, const void *out)
{
    if (!sk) return fail("server key handle is null");
    if (!aligned8(in)) return fail("input ciphertext pointer is null or not 8-byte aligned");
    if (!aligned8(out)) return fail("output ciphertext pointer is null or not 8-byte aligned");
    return 0;
}

bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError(); // plain malloc memory: not an error
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

// Large batch from page-locked buffers: chunks of PIPE_CHUNK ciphertexts (one round of the throughput kernel) on two
// alternating streams, so that the upload of chunk i+1 and the download of chunk i-1 overlap the kernels of chunk i.
// The last chunk takes the remainder (< 2 PIPE_CHUNK ciphertexts): a small tail on its own stream would run its
// latency kernel against the throughput kernel of the chunk before it, and both lose (measured: 1536 = 1024 + 512
// took 23.5 ms split over the streams, 16.9 ms in one piece).
static size_t env_size(const char *name, size_t dflt)
{
    const char *v = std::getenv(name);
    return v && *v ? (size_t)std::strtoull(v, nullptr, 10) : dflt;
}
static const size_t PIPE_CHUNK = std::max<size_t>(64, env_size("TFHE_HIP_PIPE_CHUNK", 1024));
// below ~2.5 chunks the overlap does not pay for the interference between the streams (2148: 25.4 ms piped, 24.9 ms not)
static const size_t PIPE_MIN = std::max(2 * PIPE_CHUNK, env_size("TFHE_HIP_PIPE_MIN", 5 * PIPE_CHUNK / 2));
int run_host_pipelined(tfhe_hip_server_key *sk, Op op, const uint64_t *in, const uint64_t *luts, size_t n_luts,
                       const uint32_t *idx, uint64_t *out, size_t B)
{
    const size_t big = (size_t)sk->dp.k * sk->dp.N + 1, small = sk->dp.n + 1, lut_len = (size_t)(sk->dp.k + 1) * sk->dp.N;
    const size_t in_w = (op == Op::PBS || op == Op::PBS_KS) ? small : big, out_w = (op == Op::KS || op == Op::PBS_KS) ? small : big;
    hipStream_t st[2];
    {
        std::lock_guard<std::mutex> lock(sk->mu);
        for (int i = 0; i < 2; i++) {
            if (!sk->pipe_stream[i]) HIP_TRY(hipStreamCreateWithFlags(&sk->pipe_stream[i], hipStreamNonBlocking));
            st[i] = sk->pipe_stream[i];
        }
    }
    uint64_t *d_in[2], *d_out[2], *d_luts[2] = {nullptr, nullptr};
    uint32_t *d_idx[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; i++) {
        std::lock_guard<std::mutex> lock(sk->mu);
        Workspace &w = sk->ws[st[i]];
        if (grow(&w.d_in, &w.cap_in, 2 * PIPE_CHUNK * in_w) || grow(&w.d_out, &w.cap_out, 2 * PIPE_CHUNK * out_w)) return 1;
        if (op != Op::KS && (grow(&w.d_luts, &w.cap_luts, n_luts * lut_len) || grow(&w.d_idx, &w.cap_idx, 2 * PIPE_CHUNK))) return 1;
        d_in[i] = w.d_in; d_out[i] = w.d_out; d_luts[i] = w.d_luts; d_idx[i] = w.d_idx;
    }
    if (op != Op::KS)
        for (int i = 0; i < 2; i++) HIP_TRY(hipMemcpyAsync(d_luts[i], luts, n_luts * lut_len * 8, hipMemcpyHostToDevice, st[i]));
    size_t c = 0;
    for (size_t b0 = 0; b0 < B; c++) {
        const size_t nb = B - b0 < 2 * PIPE_CHUNK ? B - b0 : PIPE_CHUNK;
        const int k = (int)(c & 1);
        HIP_TRY(hipMemcpyAsync(d_in[k], in + b0 * in_w, nb * in_w * 8, hipMemcpyHostToDevice, st[k]));
        if (op != Op::KS) HIP_TRY(hipMemcpyAsync(d_idx[k], idx + b0, nb * 4, hipMemcpyHostToDevice, st[k]));
        if (run_device_one(sk, op, d_in[k], d_luts[k], n_luts, d_idx[k], d_out[k], nb, st[k])) return 1;
        HIP_TRY(hipMemcpyAsync(out + b0 * out_w, d_out[k], nb * out_w * 8, hipMemcpyDeviceToHost, st[k]));
        b0 += nb;
    }
    HIP_TRY(hipStreamSynchronize(st[0]));
    HIP_TRY(hipStreamSynchronize(st[1]));
    return check_device_status(sk);
}

int run_host_one(tfhe_hip_server_key *sk, Op op, const uint64_t *in, const uint64_t *luts, size_t n_luts,
                 const uint32_t *idx, uint64_t *out, size_t B)
{
    if (check_batch_ptrs(sk, in, out)) return 1;
    if (B == 0) return 0;
    const size_t big = (size_t)sk->dp.k * sk->dp.N + 1, small = sk->dp.n + 1, lut_len = (size_t)(sk->dp.k + 1) * sk->dp.N;
    const size_t in_len = B * ((op == Op::PBS || op == Op::PBS_KS) ? small : big);
    const size_t out_len = B * ((op == Op::KS || op == Op::PBS_KS) ? small : big);
    if (op != Op::KS) {
        if (!aligned8(luts) || !idx || n_luts == 0) return fail("lookup tables / lut_idx pointer is null or misaligned");
        for (size_t i = 0; i < B; i++)
            if (idx[i] >= n_luts) return fail("lut_idx[" + std::to_string(i) + "] out of range");
    }
    HIP_TRY(hipSetDevice(sk->ctx->device));
    // re-entrant from many host threads (the reference shares its ServerKey between rayon workers): calls on one key
    // queue up here -- the GPU is one resource, a batch fills it
    std::lock_guard<std::mutex> host_lock(sk->host_mu);
    const bool pinned = is_pinned_host(in) && is_pinned_host(out) && (op == Op::KS || is_pinned_host(luts));
    if (pinned && B >= PIPE_MIN) return run_host_pipelined(sk, op, in, luts, n_luts, idx, out, B);
    hipStream_t s = nullptr;
    uint64_t *d_in, *d_out, *d_luts = nullptr;
    uint32_t *d_idx = nullptr;
    {
        std::lock_guard<std::mutex> lock(sk->mu);
        Workspace &w = sk->ws[s];
        if (grow(&w.d_in, &w.cap_in, in_len) || grow(&w.d_out, &w.cap_out, out_len)) return 1;
        if (op != Op::KS && (grow(&w.d_luts, &w.cap_luts, n_luts * lut_len) || grow(&w.d_idx, &w.cap_idx, B))) return 1;
        d_in = w.d_in; d_out = w.d_out; d_luts = w.d_luts; d_idx = w.d_idx;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const size_t lut_bytes = (op != Op::KS) ? n_luts * lut_len * 8 : 0, idx_bytes = (op != Op::KS) ? B * 4 : 0;
    const size_t stage_bytes = in_len * 8 + out_len * 8 + lut_bytes + idx_bytes;
    constexpr size_t PIN_LIMIT = 64u << 20; // covers a 1024-ciphertext batch in + out
    char *pin = nullptr;
    if (stage_bytes <= PIN_LIMIT && !pinned) {
        std::lock_guard<std::mutex> lock(sk->mu);
        Workspace &w = sk->ws[s];
        if (w.cap_pin < stage_bytes) {
            if (w.h_pin) HIP_TRY(hipHostFree(w.h_pin));
            w.h_pin = nullptr;
            w.cap_pin = 0;
            HIP_TRY(hipHostMalloc((void **)&w.h_pin, PIN_LIMIT, hipHostMallocDefault));
            w.cap_pin = PIN_LIMIT;
        }
        pin = w.h_pin;
    }
    const void *src_in = in, *src_luts = luts, *src_idx = idx;
    char *pin_out = nullptr;
    if (pin) { // small call: go through pinned memory (host entry points serialise per stream anyway)
        char *q = pin;
        std::memcpy(q, in, in_len * 8); src_in = q; q += in_len * 8;
        if (op != Op::KS) {
            std::memcpy(q, luts, lut_bytes); src_luts = q; q += lut_bytes;
            std::memcpy(q, idx, idx_bytes); src_idx = q; q += (idx_bytes + 7) / 8 * 8;
        }
        pin_out = q;
    }
    HIP_TRY(hipMemcpyAsync(d_in, src_in, in_len * 8, hipMemcpyHostToDevice, s));
    if (op != Op::KS) {
        HIP_TRY(hipMemcpyAsync(d_luts, src_luts, lut_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_idx, src_idx, idx_bytes, hipMemcpyHostToDevice, s));
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (run_device_one(sk, op, d_in, d_luts, n_luts, d_idx, d_out, B, s)) return 1;
    const auto t2 = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpyAsync(pin_out ? (void *)pin_out : (void *)out, d_out, out_len * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (check_device_status(sk)) return 1; // the launch flagged its own results as invalid: do not hand them out
    if (pin_out) std::memcpy(out, pin_out, out_len * 8);
    if (trace_on()) {
        const auto t3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[tfhe_hip] host call B=%zu: h2d-enqueue %.2f ms, launch %.2f ms, d2h+sync %.2f ms\n", B, ms(t0, t1),
                     ms(t1, t2), ms(t2, t3));
    }
    return 0;
}

// Host-buffer entry points on a multi-device key: every device takes a contiguous shard of the caller's arrays (shard.hpp)
// through its own replica -- one host thread per device, each with its own staging buffers and stream --, and writes its
// results straight into the caller's output rows: no exchange between the devices.
int run_host(tfhe_hip_server_key *sk, Op op, const uint64_t *in, const uint64_t *luts, size_t n_luts,
             const uint32_t *idx, uint64_t *out, size_t B)
{
    if (!sk) return fail("server key handle is null");
    const size_t active = tfhe_hip::active_devices(B, 1 + sk->peers.size(), sk->multi_min);
    if (active <= 1) return run_host_one(sk, op, in, luts, n_luts, idx, out, B);
    if (check_batch_ptrs(sk, in, out)) return 1;
    const size_t big = (size_t)sk->dp.k * sk->dp.N + 1, small = sk->dp.n + 1;
    const size_t in_w = (op == Op::PBS || op == Op::PBS_KS) ? small : big, out_w = (op == Op::KS || op == Op::PBS_KS) ? small : big;
    std::vector<int> rc(active, 0);
    std::vector<std::string> msg(active);
    auto work = [&](size_t d) {
        size_t b, e;
        tfhe_hip::shard_range(B, d, active, &b, &e);
        tfhe_hip_server_key *k = d == 0 ? sk : sk->peers[d - 1];
        rc[d] = run_host_one(k, op, in + b * in_w, luts, n_luts, idx ? idx + b : nullptr, out + b * out_w, e - b);
        if (rc[d]) msg[d] = g_last_error; // thread-local: carry it over to the caller's thread
    };
    std::vector<std::thread> th;
    for (size_t d = 1; d < active; d++) th.emplace_back(work, d);
    work(0);
    for (auto &t : th) t.join();
    for (size_t d = 0; d < active; d++)
        if (rc[d]) return fail("device shard " + std::to_string(d) + ": " + msg[d]);
    return 0;
}

template <typename F> int guarded(F &&f)
{
    try {
        return f();
    } catch (const std::exception &e) {
        return fail(std::string("exception: ") + e.what());
    } catch (...) {
        return fail("unknown exception");
    }
}

} // namespace

TFHE_HIP_API const char *tfhe_hip_last_error(void) { return g_last_error.c_str(); }

TFHE_HIP_API int tfhe_hip_host_alloc(void **out, size_t bytes)
{
    if (!out) return fail("out pointer is null");
    *out = nullptr;
    if (bytes == 0) return fail("cannot allocate zero bytes");
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return 0;
}

TFHE_HIP_API int tfhe_hip_host_free(void *p)
{
    if (p) HIP_TRY(hipHostFree(p));
    return 0;
}

static int make_context(tfhe_hip_context **out, int device_id, int count)
{
    if (device_id < 0 || device_id >= count) return fail("device_id out of range");
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    auto *ctx = new tfhe_hip_context;
    ctx->device = device_id;
    ctx->cu_count = prop.multiProcessorCount;
    ctx->arch = prop.gcnArchName;
    *out = ctx;
    return 0;
}

TFHE_HIP_API int tfhe_hip_context_create(tfhe_hip_context **out, const int *device_ids, int n_devices)
{
    return guarded([&]() -> int {
        if (!out) return fail("out pointer is null");
        *out = nullptr;
        if (!device_ids || n_devices <= 0) return fail("device_ids is null or n_devices is not positive");
        int count = 0;
        HIP_TRY(hipGetDeviceCount(&count));
        if (count <= 0) return fail("no HIP device visible: the engine has no CPU fallback");
        tfhe_hip_context *ctx = nullptr;
        if (make_context(&ctx, device_ids[0], count)) return 1;
        for (int i = 1; i < n_devices; i++) {
            tfhe_hip_context *sub = nullptr;
            if (make_context(&sub, device_ids[i], count)) {
                tfhe_hip_context_destroy(ctx);
                return 1;
            }
            ctx->subs.push_back(sub);
            if (device_ids[i] != ctx->device) { // direct copies between the devices where the fabric allows them (xGMI)
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, ctx->device, device_ids[i]) == hipSuccess && can) {
                    hipSetDevice(ctx->device);
                    (void)hipDeviceEnablePeerAccess(device_ids[i], 0);
                    hipSetDevice(device_ids[i]);
                    (void)hipDeviceEnablePeerAccess(ctx->device, 0);
                }
                (void)hipGetLastError(); // "already enabled" is not an error
            }
        }
        HIP_TRY(hipSetDevice(ctx->device));
        *out = ctx;
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_context_destroy(tfhe_hip_context *ctx)
{
    if (!ctx) return 0;
    for (tfhe_hip_context *sub : ctx->subs) delete sub;
    delete ctx;
    return 0;
}

TFHE_HIP_API int tfhe_hip_context_device_count(const tfhe_hip_context *ctx, int *n_devices)
{
    if (!ctx || !n_devices) return fail("null pointer argument");
    *n_devices = 1 + (int)ctx->subs.size();
    return 0;
}

TFHE_HIP_API int tfhe_hip_context_info(const tfhe_hip_context *ctx, char *arch_out, size_t arch_cap, int *cu_count)
{
    if (!ctx) return fail("context handle is null");
    if (arch_out && arch_cap) {
        std::strncpy(arch_out, ctx->arch.c_str(), arch_cap - 1);
        arch_out[arch_cap - 1] = 0;
    }
    if (cu_count) *cu_count = ctx->cu_count;
    return 0;
}

// bsk_seed / ksk_seed null: bsk_std / ksk are the full keys; else they are the BODIES of the seeded keys and the masks
// are generated on the device (csprng.hip)
static int create_server_key(tfhe_hip_context *ctx, tfhe_hip_server_key **out, const tfhe_hip_params *params,
                             int grouping_factor, const uint64_t *bsk_std, size_t bsk_len, const uint64_t *ksk, size_t ksk_len,
                             const uint8_t *bsk_seed = nullptr, const uint8_t *ksk_seed = nullptr);

TFHE_HIP_API int tfhe_hip_server_key_create(tfhe_hip_context *ctx, tfhe_hip_server_key **out,
                                            const tfhe_hip_params *params, const uint64_t *bsk_std, size_t bsk_len,
                                            const uint64_t *ksk, size_t ksk_len)
{
    return create_server_key(ctx, out, params, 0, bsk_std, bsk_len, ksk, ksk_len);
}

TFHE_HIP_API int tfhe_hip_server_key_create_multi_bit(tfhe_hip_context *ctx, tfhe_hip_server_key **out,
                                                      const tfhe_hip_params *params, uint32_t grouping_factor,
                                                      const uint64_t *bsk_std, size_t bsk_len, const uint64_t *ksk,
                                                      size_t ksk_len)
{
    if (grouping_factor != 2 && grouping_factor != 3) {
        if (out) *out = nullptr;
        return fail("grouping_factor must be 2 or 3");
    }
    return create_server_key(ctx, out, params, (int)grouping_factor, bsk_std, bsk_len, ksk, ksk_len);
}

TFHE_HIP_API int tfhe_hip_server_key_create_seeded(tfhe_hip_context *ctx, tfhe_hip_server_key **out,
                                                   const tfhe_hip_params *params, uint32_t grouping_factor,
                                                   const uint64_t *bsk_bodies, size_t bsk_bodies_len, const uint8_t *bsk_seed,
                                                   const uint64_t *ksk_bodies, size_t ksk_bodies_len, const uint8_t *ksk_seed)
{
    if (!bsk_seed || !ksk_seed) {
        if (out) *out = nullptr;
        return fail("seed pointer is null");
    }
    if (grouping_factor != 0 && grouping_factor != 2 && grouping_factor != 3) {
        if (out) *out = nullptr;
        return fail("grouping_factor must be 0 (classic), 2 or 3");
    }
    return create_server_key(ctx, out, params, (int)grouping_factor, bsk_bodies, bsk_bodies_len, ksk_bodies, ksk_bodies_len, bsk_seed,
                             ksk_seed);
}

static int create_server_key(tfhe_hip_context *ctx, tfhe_hip_server_key **out, const tfhe_hip_params *params,
                             int grouping_factor, const uint64_t *bsk_std, size_t bsk_len, const uint64_t *ksk, size_t ksk_len,
                             const uint8_t *bsk_seed, const uint8_t *ksk_seed)
{
    return guarded([&]() -> int {
        if (!out) return fail("out pointer is null");
        *out = nullptr;
        if (!ctx) return fail("context handle is null");
        if (check_params(params)) return 1;
        if (!aligned8(bsk_std) || !aligned8(ksk)) return fail("key pointer is null or not 8-byte aligned");
        const tfhe_hip_params &p = *params;
        const size_t gs = p.glwe_dim + 1;
        if (grouping_factor && (p.lwe_dim % (uint32_t)grouping_factor || p.pbs_order != 0))
            return fail("multi-bit PBS: lwe_dim must be a multiple of the grouping factor, pbs_order KeyswitchBootstrap");
        // LweMultiBitBootstrapKey: 2^g GGSWs per group of g key bits (entities/lwe_multi_bit_bootstrap_key.rs)
        const size_t ggsw_count = grouping_factor ? (size_t)(p.lwe_dim / grouping_factor) << grouping_factor : (size_t)p.lwe_dim;
        const size_t want_bsk = ggsw_count * p.pbs_level * gs * gs * p.poly_size;
        const size_t want_ksk = (size_t)p.glwe_dim * p.poly_size * p.ks_level * (p.lwe_dim + 1);
        const bool seeded = bsk_seed != nullptr;
        // seeded keys hold one body polynomial per GLWE row / one body per keyswitch ciphertext
        // (entities/seeded_ggsw_ciphertext_list.rs, seeded_lwe_keyswitch_key.rs)
        const size_t bsk_units = ggsw_count * p.pbs_level * gs, ksk_units = (size_t)p.glwe_dim * p.poly_size * p.ks_level;
        const size_t given_bsk = seeded ? bsk_units * p.poly_size : want_bsk, given_ksk = seeded ? ksk_units : want_ksk;
        // the reference asserts these in LweBootstrapKey::from_container / LweKeyswitchKey::from_container
        if (bsk_len != given_bsk) return fail("bsk_len does not match the parameters (expected " + std::to_string(given_bsk) + ")");
        if (ksk_len != given_ksk) return fail("ksk_len does not match the parameters (expected " + std::to_string(given_ksk) + ")");
        bsk_len = want_bsk;
        ksk_len = want_ksk;
        HIP_TRY(hipSetDevice(ctx->device));

        auto *sk = new (std::nothrow) tfhe_hip_server_key;
        if (!sk) return fail("out of host memory");
        sk->ctx = ctx;
        sk->params = p;
        sk->dp.n = (int)p.lwe_dim; sk->dp.k = (int)p.glwe_dim; sk->dp.N = (int)p.poly_size;
        sk->dp.log2N = 0;
        while ((1u << sk->dp.log2N) < p.poly_size) sk->dp.log2N++;
        sk->dp.pbs_base_log = (int)p.pbs_base_log; sk->dp.pbs_level = (int)p.pbs_level;
        sk->dp.ks_base_log = (int)p.ks_base_log; sk->dp.ks_level = (int)p.ks_level;
        sk->bsk_polys = want_bsk / p.poly_size;
        sk->grouping_factor = grouping_factor;
        if (grouping_factor && !tfhe_hip::blind_rotate_multi_bit_supported(sk->dp, grouping_factor)) {
            delete sk;
            return fail("multi-bit kernel is built for k = 1, N = 2048, one PBS level, base 2^22 (g = 2) or 2^21 (g = 3)");
        }

        auto cleanup = [&]() { tfhe_hip_server_key_destroy(sk); };
        const uint32_t fn = p.poly_size / 2;
        std::vector<double> tw(2 * fn), w(2 * fn);
        tfhe_hip_make_tables(fn, tw.data(), w.data());
        uint64_t *d_bsk_std = nullptr, *d_bodies = nullptr;
        uint32_t *d_te0 = nullptr;
#define HIP_TRY_C(expr)                                                                                        \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            hipFree(d_bsk_std);                                                                                \
            hipFree(d_bodies);                                                                                 \
            hipFree(d_te0);                                                                                    \
            cleanup();                                                                                         \
            return fail(std::string(#expr) + " failed: " + hipGetErrorString(e_));                             \
        }                                                                                                      \
    } while (0)
        HIP_TRY_C(hipHostMalloc((void **)&sk->h_err, 64, hipHostMallocMapped));
        *sk->h_err = 0;
        HIP_TRY_C(hipHostGetDevicePointer((void **)&sk->d_err, sk->h_err, 0));
        HIP_TRY_C(hipMalloc((void **)&sk->d_twist, tw.size() * 8));
        HIP_TRY_C(hipMalloc((void **)&sk->d_w, w.size() * 8));
        HIP_TRY_C(hipMemcpy(sk->d_twist, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY_C(hipMemcpy(sk->d_w, w.data(), w.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY_C(hipMalloc((void **)&sk->d_ksk, ksk_len * 8));
        if (seeded) { // seeded_lwe_keyswitch_key_decompression.rs on the device
            HIP_TRY_C(tfhe_hip::upload_aes_table(&d_te0));
            HIP_TRY_C(hipMalloc((void **)&d_bodies, std::max(given_ksk, given_bsk) * 8));
            HIP_TRY_C(hipMemcpy(d_bodies, ksk, given_ksk * 8, hipMemcpyHostToDevice));
            HIP_TRY_C(tfhe_hip::launch_seeded_expand(d_te0, ksk_seed, d_bodies, ksk_units, p.lwe_dim, 1, sk->d_ksk, nullptr));
        } else {
            HIP_TRY_C(hipMemcpy(sk->d_ksk, ksk, ksk_len * 8, hipMemcpyHostToDevice));
        }
        HIP_TRY_C(hipMalloc((void **)&sk->d_colsum, (size_t)(p.lwe_dim + 1) * 8));
        HIP_TRY_C(tfhe_hip::launch_ksk_colsum(sk->dp, sk->d_ksk, sk->d_colsum, nullptr));
        HIP_TRY_C(hipMalloc((void **)&sk->d_ksk_planes, ksk_len * 8));
        HIP_TRY_C(tfhe_hip::launch_ksk_to_planes(sk->dp, sk->d_ksk, sk->d_ksk_planes, nullptr));
        if (tfhe_hip::keyswitch_mfma_supported(sk->dp)) {
            HIP_TRY_C(hipMalloc(&sk->d_ksk_tiles, tfhe_hip::keyswitch_mfma_key_bytes(sk->dp)));
            HIP_TRY_C(tfhe_hip::launch_ksk_to_mfma_tiles(sk->dp, sk->d_ksk, sk->d_ksk_tiles, nullptr));
            if (const char *v = std::getenv("TFHE_HIP_KS_MFMA_MIN")) sk->ks_mfma_min = std::strtoull(v, nullptr, 10); // 0: never
            if (sk->ks_mfma_min == 0) sk->ks_mfma_min = (size_t)-1;
        }
        HIP_TRY_C(hipDeviceSynchronize());
        HIP_TRY_C(hipFree(sk->d_ksk));
        sk->d_ksk = nullptr;
        HIP_TRY_C(hipMalloc((void **)&d_bsk_std, bsk_len * 8));
        if (seeded) { // seeded_lwe_bootstrap_key_decompression.rs on the device
            HIP_TRY_C(hipMemcpy(d_bodies, bsk_std, given_bsk * 8, hipMemcpyHostToDevice));
            HIP_TRY_C(tfhe_hip::launch_seeded_expand(d_te0, bsk_seed, d_bodies, bsk_units, (size_t)p.glwe_dim * p.poly_size, p.poly_size,
                                                     d_bsk_std, nullptr));
            HIP_TRY_C(hipDeviceSynchronize());
            hipFree(d_bodies);
            hipFree(d_te0);
            d_bodies = nullptr;
            d_te0 = nullptr;
        } else {
            HIP_TRY_C(hipMemcpy(d_bsk_std, bsk_std, bsk_len * 8, hipMemcpyHostToDevice));
        }
        HIP_TRY_C(hipMalloc((void **)&sk->d_bsk_f, bsk_len * 8)); // N/2 complex = N doubles per polynomial
        HIP_TRY_C(tfhe_hip::launch_bsk_to_fourier(sk->dp, d_bsk_std, sk->d_bsk_f, sk->bsk_polys, sk->d_twist, sk->d_w, nullptr));
        if (grouping_factor) {
            HIP_TRY_C(hipMalloc((void **)&sk->d_bsk_mb, bsk_len * 8));
            HIP_TRY_C(tfhe_hip::launch_bsk_permute_block_mb(sk->dp, grouping_factor, sk->d_bsk_f, sk->d_bsk_mb, nullptr));
        } else if (tfhe_hip::blind_rotate_wave_supported(sk->dp)) {
            const std::vector<char> gtab = tfhe_hip::wave::build_tables(tw.data(), w.data());
            HIP_TRY_C(hipMalloc((void **)&sk->d_gtab, gtab.size()));
            HIP_TRY_C(hipMemcpy(sk->d_gtab, gtab.data(), gtab.size(), hipMemcpyHostToDevice));
            const int wc_idx[4] = {64, 128, 192, 384};
            for (int i = 0; i < 4; i++) {
                sk->wconsts[2 * i] = w[2 * wc_idx[i]];
                sk->wconsts[2 * i + 1] = w[2 * wc_idx[i] + 1];
            }
            const std::vector<char> gtab_p = tfhe_hip::pair::build_tables(tw.data(), w.data());
            HIP_TRY_C(hipMalloc((void **)&sk->d_gtab_p, gtab_p.size()));
            HIP_TRY_C(hipMemcpy(sk->d_gtab_p, gtab_p.data(), gtab_p.size(), hipMemcpyHostToDevice));
            HIP_TRY_C(hipMalloc((void **)&sk->d_bsk_p, bsk_len * 8));
            HIP_TRY_C(tfhe_hip::launch_bsk_permute_pair(sk->dp, sk->d_bsk_f, sk->d_bsk_p, nullptr));
            HIP_TRY_C(hipMalloc((void **)&sk->d_bsk_b, bsk_len * 8));
            HIP_TRY_C(tfhe_hip::launch_bsk_permute_block(sk->dp, sk->d_bsk_f, sk->d_bsk_b, nullptr));
            HIP_TRY_C(hipMalloc((void **)&sk->d_bsk_q, bsk_len * 8));
            HIP_TRY_C(tfhe_hip::launch_bsk_permute_quad(sk->dp, sk->d_bsk_f, sk->d_bsk_q, nullptr));
            // default: the latency kernel while its rounds (one workgroup per CU, 2.8 ms each) beat the flat
            // 8.9 ms of the wave-pair kernel, i.e. up to three rounds; the wave-pair kernel beyond
            sk->block_max_batch = 3 * (size_t)ctx->cu_count;
            sk->variant = -1;
            // the two default kernels read their own copies: the oracle-order image (variant 0, introspection) and the
            // lane-order one (variant 1) are rebuilt from the pair-order copy when somebody asks for them (ensure_bsk_*)
            HIP_TRY_C(hipDeviceSynchronize());
            HIP_TRY_C(hipFree(sk->d_bsk_f));
            sk->d_bsk_f = nullptr;
        }
        HIP_TRY_C(hipDeviceSynchronize());
        HIP_TRY_C(hipFree(d_bsk_std));
#undef HIP_TRY_C
        // multi-device context: a full replica of the key on every further device, each with its stream and events
        if (const char *v = std::getenv("TFHE_HIP_MULTI_MIN")) sk->multi_min = std::max<size_t>(1, std::strtoull(v, nullptr, 10));
        if (const char *v = std::getenv("TFHE_HIP_MAX_STREAM_WS")) sk->ws_limit = std::max<size_t>(1, std::strtoull(v, nullptr, 10));
        for (tfhe_hip_context *sub : ctx->subs) {
            tfhe_hip_server_key *peer = nullptr;
            if (create_server_key(sub, &peer, params, grouping_factor, bsk_std, seeded ? given_bsk : want_bsk, ksk,
                                  seeded ? given_ksk : want_ksk, bsk_seed, ksk_seed)) {
                cleanup();
                return 1;
            }
            sk->peers.push_back(peer);
            sk->peer_io.emplace_back();
            tfhe_hip_server_key::PeerIo &io = sk->peer_io.back();
            hipError_t e = hipSetDevice(sub->device);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&io.stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&io.done, hipEventDisableTiming);
            if (e == hipSuccess) e = hipSetDevice(ctx->device);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&io.ready, hipEventDisableTiming);
            if (e != hipSuccess) {
                cleanup();
                return fail(std::string("multi-device key set-up failed: ") + hipGetErrorString(e));
            }
        }
        HIP_TRY(hipSetDevice(ctx->device));
        *out = sk;
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_csprng_fill(tfhe_hip_context *ctx, const uint8_t *seed, uint64_t first_word, uint64_t *out, size_t count)
{
    return guarded([&]() -> int {
        if (!ctx) return fail("context handle is null");
        if (!seed) return fail("seed pointer is null");
        if (count == 0) return 0;
        if (!aligned8(out)) return fail("out pointer is null or not 8-byte aligned");
        HIP_TRY(hipSetDevice(ctx->device));
        uint32_t *d_te0 = nullptr;
        uint64_t *d_out = nullptr;
        hipError_t e = tfhe_hip::upload_aes_table(&d_te0);
        if (e == hipSuccess) e = hipMalloc((void **)&d_out, count * 8);
        if (e == hipSuccess) e = tfhe_hip::launch_csprng_fill(d_te0, seed, first_word, count, count, count, d_out, nullptr);
        if (e == hipSuccess) e = hipMemcpy(out, d_out, count * 8, hipMemcpyDeviceToHost);
        hipFree(d_out);
        hipFree(d_te0);
        if (e != hipSuccess) return fail(std::string("csprng fill failed: ") + hipGetErrorString(e));
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_lwe_decompress_seeded_batch(tfhe_hip_context *ctx, const uint64_t *bodies, const uint8_t *seeds,
                                                      uint32_t lwe_dim, uint64_t *lwe_out, size_t B)
{
    return guarded([&]() -> int {
        if (!ctx) return fail("context handle is null");
        if (B == 0) return 0;
        if (!aligned8(bodies) || !aligned8(lwe_out) || !seeds) return fail("bodies / seeds / lwe_out pointer is null or misaligned");
        if (lwe_dim == 0) return fail("lwe_dim must be positive");
        HIP_TRY(hipSetDevice(ctx->device));
        std::vector<uint32_t> rks(B * 44);
        for (size_t i = 0; i < B; i++) tfhe_hip::aes128_round_keys(seeds + 16 * i, rks.data() + 44 * i);
        uint32_t *d_te0 = nullptr, *d_rks = nullptr;
        uint64_t *d_bodies = nullptr, *d_out = nullptr;
        const size_t CHUNK = 32768; // grid.y limit
        hipError_t e = tfhe_hip::upload_aes_table(&d_te0);
        const size_t cap = std::min(B, CHUNK);
        if (e == hipSuccess) e = hipMalloc((void **)&d_rks, cap * 44 * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&d_bodies, cap * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&d_out, cap * (size_t)(lwe_dim + 1) * 8);
        for (size_t b0 = 0; b0 < B && e == hipSuccess; b0 += CHUNK) {
            const size_t nb = std::min(CHUNK, B - b0);
            e = hipMemcpy(d_rks, rks.data() + 44 * b0, nb * 44 * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(d_bodies, bodies + b0, nb * 8, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = tfhe_hip::launch_seeded_expand_units(d_te0, d_rks, d_bodies, nb, lwe_dim, d_out, nullptr);
            if (e == hipSuccess) e = hipMemcpy(lwe_out + b0 * (size_t)(lwe_dim + 1), d_out, nb * (size_t)(lwe_dim + 1) * 8, hipMemcpyDeviceToHost);
        }
        hipFree(d_out);
        hipFree(d_bodies);
        hipFree(d_rks);
        hipFree(d_te0);
        if (e != hipSuccess) return fail(std::string("seeded ciphertext decompression failed: ") + hipGetErrorString(e));
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_server_key_get_params(const tfhe_hip_server_key *sk, tfhe_hip_params *out)
{
    if (!sk || !out) return fail("null pointer argument");
    *out = sk->params;
    return 0;
}

TFHE_HIP_API int tfhe_hip_server_key_batch_rounds(const tfhe_hip_server_key *sk, size_t *latency_round, size_t *throughput_round)
{
    if (!sk || !latency_round || !throughput_round) return fail("null pointer argument");
    const size_t cus = (sk->ctx ? (size_t)sk->ctx->cu_count : 0) * (1 + sk->peers.size()); // the devices of the context together
    *latency_round = cus;
    *throughput_round = (sk->grouping_factor == 0 && sk->d_bsk_p) ? 4 * cus : 0;
    return 0;
}

// ---- device-resident buffers for level schedulers (host_ops.hpp): ciphertexts stay in HBM between the levels of an
// op-graph; only the inputs go up and the results come down.  Everything runs on the default stream, in call order.
TFHE_HIP_API int tfhe_hip_device_alloc(tfhe_hip_server_key *sk, uint64_t **out, size_t words)
{
    return guarded([&]() -> int {
        if (!out) return fail("out pointer is null");
        *out = nullptr;
        if (!sk) return fail("server key handle is null");
        if (words == 0) return fail("cannot allocate zero words");
        HIP_TRY(hipSetDevice(sk->ctx->device));
        HIP_TRY(hipMalloc((void **)out, words * 8));
        return 0;
    });
}
TFHE_HIP_API int tfhe_hip_device_free(uint64_t *p)
{
    if (!p) return 0; // (no key argument: pools may outlive the key they were filled for)
    return hipFree(p) == hipSuccess ? 0 : fail("hipFree failed");
}
TFHE_HIP_API int tfhe_hip_device_upload(tfhe_hip_server_key *sk, uint64_t *d_dst, const void *h_src, size_t bytes)
{
    return guarded([&]() -> int {
        if (!sk || !d_dst || !h_src) return fail("null pointer argument");
        HIP_TRY(hipSetDevice(sk->ctx->device));
        HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, nullptr));
        // pageable sources are staged by the runtime before the call returns; page-locked ones are read by the DMA engine
        // later: the caller keeps them unchanged until tfhe_hip_device_synchronize / _download
        return 0;
    });
}
// The transfer stream does not synchronise with the default stream by itself (hipStreamNonBlocking): the two events carry
// the ordering the header promises.  An event is re-recorded per call; a wait refers to the record that precedes it.
TFHE_HIP_API int tfhe_hip_device_upload_overlapped(tfhe_hip_server_key *sk, uint64_t *d_dst, const void *h_src, size_t bytes,
                                                   int wait_prior)
{
    return guarded([&]() -> int {
        if (!sk || !d_dst || !h_src) return fail("null pointer argument");
        if (!is_pinned_host(h_src)) return fail("an overlapped upload needs a page-locked source (tfhe_hip_host_alloc)");
        HIP_TRY(hipSetDevice(sk->ctx->device));
        std::lock_guard<std::mutex> lock(sk->mu);
        if (!sk->copy_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&sk->copy_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&sk->copy_after, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&sk->copy_done, hipEventDisableTiming));
        }
        if (wait_prior) {
            HIP_TRY(hipEventRecord(sk->copy_after, nullptr));
            HIP_TRY(hipStreamWaitEvent(sk->copy_stream, sk->copy_after, 0));
        }
        HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, sk->copy_stream));
        HIP_TRY(hipEventRecord(sk->copy_done, sk->copy_stream));
        HIP_TRY(hipStreamWaitEvent(nullptr, sk->copy_done, 0));
        return 0;
    });
}
TFHE_HIP_API int tfhe_hip_device_synchronize(tfhe_hip_server_key *sk)
{
    return guarded([&]() -> int {
        if (!sk) return fail("server key handle is null");
        HIP_TRY(hipSetDevice(sk->ctx->device));
        HIP_TRY(hipStreamSynchronize(nullptr));
        return check_device_status(sk);
    });
}
TFHE_HIP_API int tfhe_hip_device_download(tfhe_hip_server_key *sk, void *h_dst, const uint64_t *d_src, size_t bytes)
{
    return guarded([&]() -> int {
        if (!sk || !h_dst || !d_src) return fail("null pointer argument");
        HIP_TRY(hipSetDevice(sk->ctx->device));
        HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        return check_device_status(sk);
    });
}
// d_dst[r][..] = sum_{t = row_ptr[r] .. row_ptr[r+1]} coef[t] * src[t][..] + (0, .., 0, body[r]); row_ptr, coef, src, body are HOST arrays
TFHE_HIP_API int tfhe_hip_device_lincomb_rows(tfhe_hip_server_key *sk, uint64_t *d_dst, size_t rows, size_t row_words,
                                              const uint32_t *row_ptr, const uint64_t *coef, const uint64_t *const *src,
                                              const uint64_t *body)
{
    return guarded([&]() -> int {
        if (!sk) return fail("server key handle is null");
        if (rows == 0) return 0;
        if (!d_dst || !row_ptr || !body) return fail("null pointer argument");
        if (row_words == 0 || row_words > (1u << 24)) return fail("row_words out of range");
        if (row_ptr[0] != 0) return fail("row_ptr must start at 0");
        for (size_t r = 0; r < rows; r++)
            if (row_ptr[r + 1] < row_ptr[r]) return fail("row_ptr must be non-decreasing");
        const size_t terms = row_ptr[rows];
        if (terms && (!coef || !src)) return fail("null pointer argument");
        for (size_t t = 0; t < terms; t++)
            if (!src[t] || ((uintptr_t)src[t] & 7)) return fail("term source is null or misaligned");
        HIP_TRY(hipSetDevice(sk->ctx->device));
        std::lock_guard<std::mutex> lock(sk->host_mu);
        // one upload: [row_ptr (rows + 1) u32, padded][coef][src][body]
        const size_t o_coef = ((rows + 1) * 4 + 7) / 8 * 8, o_src = o_coef + terms * 8, o_body = o_src + terms * 8, total = o_body + rows * 8;
        std::vector<char> h(total);
        std::memcpy(h.data(), row_ptr, (rows + 1) * 4);
        if (terms) {
            std::memcpy(h.data() + o_coef, coef, terms * 8);
            std::memcpy(h.data() + o_src, src, terms * 8);
        }
        std::memcpy(h.data() + o_body, body, rows * 8);
        if (sk->cap_csr < total) {
            HIP_TRY(hipFree(sk->d_csr));
            sk->d_csr = nullptr;
            sk->cap_csr = 0;
            HIP_TRY(hipMalloc((void **)&sk->d_csr, total * 2));
            sk->cap_csr = total * 2;
        }
        HIP_TRY(hipMemcpyAsync(sk->d_csr, h.data(), total, hipMemcpyHostToDevice, nullptr)); // pageable: staged before it returns
        HIP_TRY(tfhe_hip::launch_lincomb_rows(d_dst, rows, (int)row_words, (const uint32_t *)sk->d_csr, (const uint64_t *)(sk->d_csr + o_coef),
                                              (const uint64_t *const *)(sk->d_csr + o_src), (const uint64_t *)(sk->d_csr + o_body), nullptr));
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_server_key_destroy(tfhe_hip_server_key *sk)
{
    if (!sk) return 0;
    for (size_t i = 0; i < sk->peers.size(); i++) {
        if (i < sk->peer_io.size()) {
            tfhe_hip_server_key::PeerIo &io = sk->peer_io[i];
            hipSetDevice(sk->peers[i]->ctx->device);
            hipDeviceSynchronize();
            if (io.stream) hipStreamDestroy(io.stream);
            if (io.done) hipEventDestroy(io.done);
            hipFree(io.d_in); hipFree(io.d_out); hipFree(io.d_luts); hipFree(io.d_idx);
            if (io.ready) {
                hipSetDevice(sk->ctx->device);
                hipEventDestroy(io.ready);
            }
        }
        tfhe_hip_server_key_destroy(sk->peers[i]);
    }
    sk->peers.clear();
    if (sk->ctx) hipSetDevice(sk->ctx->device);
    hipDeviceSynchronize();
    for (auto &kv : sk->ws) free_ws(kv.second);
    for (auto &st : sk->pipe_stream)
        if (st) hipStreamDestroy(st);
    for (auto &ep : sk->pending) { hipEventDestroy(ep.a); hipEventDestroy(ep.b); }
    hipFree(sk->d_ksk); hipFree(sk->d_ksk_planes); hipFree(sk->d_ksk_tiles); hipFree(sk->d_colsum); hipFree(sk->d_bsk_f); hipFree(sk->d_twist); hipFree(sk->d_w);
    hipFree(sk->d_csr);
    if (sk->copy_stream) hipStreamDestroy(sk->copy_stream);
    if (sk->copy_after) hipEventDestroy(sk->copy_after);
    if (sk->copy_done) hipEventDestroy(sk->copy_done);
    if (sk->h_err) hipHostFree(sk->h_err);
    hipFree(sk->d_bsk_w); hipFree(sk->d_gtab); hipFree(sk->d_bsk_p); hipFree(sk->d_gtab_p); hipFree(sk->d_bsk_b); hipFree(sk->d_bsk_q); hipFree(sk->d_bsk_mb);
    delete sk;
    return 0;
}

TFHE_HIP_API int tfhe_hip_lut_generate(const tfhe_hip_params *params, const uint64_t *table, uint64_t *glwe_out,
                                       uint64_t *max_out)
{
    // shortint/engine/mod.rs:72-128 (fill_accumulator)
    if (!params) return fail("params pointer is null");
    if (!aligned8(table) || !aligned8(glwe_out)) return fail("table / glwe_out pointer is null or misaligned");
    const tfhe_hip_params &p = *params;
    const uint32_t modulus_sup = p.message_modulus * p.carry_modulus;
    if (modulus_sup == 0 || p.poly_size % modulus_sup) return fail("message_modulus * carry_modulus must divide poly_size");
    const size_t N = p.poly_size, k = p.glwe_dim;
    const size_t box = N / modulus_sup, half = box / 2;
    const uint64_t delta = ((uint64_t)1 << 63) / modulus_sup;
    std::memset(glwe_out, 0, sizeof(uint64_t) * k * N);
    uint64_t *body = glwe_out + k * N;
    uint64_t max_value = 0;
    // body before rotation: box i holds f(i)*delta, first half-box negated; then rotate_left(half)
    for (size_t j = 0; j < N; j++) {
        const size_t src = (j + half) % N;
        const uint64_t f = table[src / box];
        uint64_t v = f * delta;
        if (src < half) v = (uint64_t)0 - v;
        body[j] = v;
    }
    for (uint32_t i = 0; i < modulus_sup; i++)
        if (table[i] > max_value) max_value = table[i];
    if (max_out) *max_out = max_value;
    return 0;
}

TFHE_HIP_API int tfhe_hip_keyswitch_batch(tfhe_hip_server_key *sk, const uint64_t *lwe_in, uint64_t *lwe_out, size_t B)
{
    return guarded([&]() -> int { return run_host(sk, Op::KS, lwe_in, nullptr, 0, nullptr, lwe_out, B); });
}

TFHE_HIP_API int tfhe_hip_pbs_batch(tfhe_hip_server_key *sk, const uint64_t *lwe_in, const uint64_t *luts,
                                    size_t n_luts, const uint32_t *lut_idx, uint64_t *lwe_out, size_t B)
{
    return guarded([&]() -> int { return run_host(sk, Op::PBS, lwe_in, luts, n_luts, lut_idx, lwe_out, B); });
}

TFHE_HIP_API int tfhe_hip_ks_pbs_batch(tfhe_hip_server_key *sk, const uint64_t *lwe_in, const uint64_t *luts,
                                       size_t n_luts, const uint32_t *lut_idx, uint64_t *lwe_out, size_t B)
{
    return guarded([&]() -> int { return run_host(sk, Op::KS_PBS, lwe_in, luts, n_luts, lut_idx, lwe_out, B); });
}

TFHE_HIP_API int tfhe_hip_keyswitch_batch_device(tfhe_hip_server_key *sk, const uint64_t *d_lwe_in,
                                                 uint64_t *d_lwe_out, size_t B, void *stream)
{
    return guarded([&]() -> int {
        if (check_batch_ptrs(sk, d_lwe_in, d_lwe_out)) return 1;
        return run_device(sk, Op::KS, d_lwe_in, nullptr, 0, nullptr, d_lwe_out, B, (hipStream_t)stream);
    });
}

TFHE_HIP_API int tfhe_hip_pbs_batch_device(tfhe_hip_server_key *sk, const uint64_t *d_lwe_in, const uint64_t *d_luts,
                                           size_t n_luts, const uint32_t *d_lut_idx, uint64_t *d_lwe_out, size_t B,
                                           void *stream)
{
    return guarded([&]() -> int {
        if (check_batch_ptrs(sk, d_lwe_in, d_lwe_out)) return 1;
        if (!aligned8(d_luts) || !d_lut_idx || n_luts == 0) return fail("lookup tables / lut_idx pointer is null or misaligned");
        return run_device(sk, Op::PBS, d_lwe_in, d_luts, n_luts, d_lut_idx, d_lwe_out, B, (hipStream_t)stream);
    });
}

TFHE_HIP_API int tfhe_hip_ks_pbs_batch_device(tfhe_hip_server_key *sk, const uint64_t *d_lwe_in,
                                              const uint64_t *d_luts, size_t n_luts, const uint32_t *d_lut_idx,
                                              uint64_t *d_lwe_out, size_t B, void *stream)
{
    return guarded([&]() -> int {
        if (check_batch_ptrs(sk, d_lwe_in, d_lwe_out)) return 1;
        if (!aligned8(d_luts) || !d_lut_idx || n_luts == 0) return fail("lookup tables / lut_idx pointer is null or misaligned");
        return run_device(sk, Op::KS_PBS, d_lwe_in, d_luts, n_luts, d_lut_idx, d_lwe_out, B, (hipStream_t)stream);
    });
}

TFHE_HIP_API int tfhe_hip_pbs_ks_batch(tfhe_hip_server_key *sk, const uint64_t *lwe_in, const uint64_t *luts,
                                       size_t n_luts, const uint32_t *lut_idx, uint64_t *lwe_out, size_t B)
{
    return guarded([&]() -> int { return run_host(sk, Op::PBS_KS, lwe_in, luts, n_luts, lut_idx, lwe_out, B); });
}

TFHE_HIP_API int tfhe_hip_pbs_ks_batch_device(tfhe_hip_server_key *sk, const uint64_t *d_lwe_in, const uint64_t *d_luts,
                                              size_t n_luts, const uint32_t *d_lut_idx, uint64_t *d_lwe_out, size_t B,
                                              void *stream)
{
    return guarded([&]() -> int {
        if (check_batch_ptrs(sk, d_lwe_in, d_lwe_out)) return 1;
        if (!aligned8(d_luts) || !d_lut_idx || n_luts == 0) return fail("lookup tables / lut_idx pointer is null or misaligned");
        return run_device(sk, Op::PBS_KS, d_lwe_in, d_luts, n_luts, d_lut_idx, d_lwe_out, B, (hipStream_t)stream);
    });
}

// ServerKey::apply_lookup_table_assign: dispatch on the key's PBSOrder (shortint/server_key/mod.rs:465-476)
TFHE_HIP_API int tfhe_hip_apply_lookup_table_batch(tfhe_hip_server_key *sk, const uint64_t *lwe_in, const uint64_t *luts,
                                                   size_t n_luts, const uint32_t *lut_idx, uint64_t *lwe_out, size_t B)
{
    return guarded([&]() -> int {
        if (!sk) return fail("server key handle is null");
        return run_host(sk, sk->params.pbs_order == 0 ? Op::KS_PBS : Op::PBS_KS, lwe_in, luts, n_luts, lut_idx, lwe_out, B);
    });
}

TFHE_HIP_API int tfhe_hip_apply_lookup_table_batch_device(tfhe_hip_server_key *sk, const uint64_t *d_lwe_in,
                                                          const uint64_t *d_luts, size_t n_luts,
                                                          const uint32_t *d_lut_idx, uint64_t *d_lwe_out, size_t B,
                                                          void *stream)
{
    return guarded([&]() -> int {
        if (check_batch_ptrs(sk, d_lwe_in, d_lwe_out)) return 1;
        if (!aligned8(d_luts) || !d_lut_idx || n_luts == 0) return fail("lookup tables / lut_idx pointer is null or misaligned");
        return run_device(sk, sk->params.pbs_order == 0 ? Op::KS_PBS : Op::PBS_KS, d_lwe_in, d_luts, n_luts, d_lut_idx,
                          d_lwe_out, B, (hipStream_t)stream);
    });
}

TFHE_HIP_API int tfhe_hip_profiling_enable(tfhe_hip_server_key *sk, int on)
{
    if (!sk) return fail("server key handle is null");
    std::lock_guard<std::mutex> lock(sk->mu);
    sk->profiling = on != 0;
    return 0;
}

TFHE_HIP_API int tfhe_hip_profiling_read(tfhe_hip_server_key *sk, int kernel, double *total_ms, uint64_t *launches,
                                         int reset)
{
    return guarded([&]() -> int {
        if (!sk) return fail("server key handle is null");
        if (kernel < 0 || kernel > 1) return fail("kernel must be 0 (keyswitch) or 1 (blind rotation)");
        HIP_TRY(hipSetDevice(sk->ctx->device));
        std::lock_guard<std::mutex> lock(sk->mu);
        for (auto &ep : sk->pending) {
            HIP_TRY(hipEventSynchronize(ep.b));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, ep.a, ep.b));
            sk->total_ms[ep.kernel] += ms;
            sk->launches[ep.kernel] += 1;
            hipEventDestroy(ep.a);
            hipEventDestroy(ep.b);
        }
        sk->pending.clear();
        if (total_ms) *total_ms = sk->total_ms[kernel];
        if (launches) *launches = sk->launches[kernel];
        if (reset) {
            sk->total_ms[0] = sk->total_ms[1] = 0;
            sk->launches[0] = sk->launches[1] = 0;
        }
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_server_key_fourier_bsk(tfhe_hip_server_key *sk, double *re_out, double *im_out, size_t len)
{
    return guarded([&]() -> int {
        if (!sk) return fail("server key handle is null");
        if (!aligned8(re_out) || !aligned8(im_out)) return fail("output pointer is null or misaligned");
        const size_t want = sk->bsk_polys * (sk->dp.N / 2);
        if (len != want) return fail("len must be " + std::to_string(want));
        HIP_TRY(hipSetDevice(sk->ctx->device));
        std::vector<double> tmp(2 * want);
        {
            std::lock_guard<std::mutex> lock(sk->mu);
            if (ensure_bsk_f(sk)) return 1;
        }
        HIP_TRY(hipMemcpy(tmp.data(), sk->d_bsk_f, tmp.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < want; i++) {
            re_out[i] = tmp[2 * i];
            im_out[i] = tmp[2 * i + 1];
        }
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_server_key_device_status(tfhe_hip_server_key *sk, uint32_t *code, int clear)
{
    return guarded([&]() -> int {
        if (!sk) return fail("server key handle is null");
        uint32_t v = sk->h_err ? __atomic_load_n(sk->h_err, __ATOMIC_ACQUIRE) : 0;
        for (tfhe_hip_server_key *peer : sk->peers) {
            uint32_t pv = 0;
            if (tfhe_hip_server_key_device_status(peer, &pv, clear)) return 1;
            v |= pv;
        }
        if (code) *code = v;
        if (clear && sk->h_err) {
            HIP_TRY(hipSetDevice(sk->ctx->device));
            HIP_TRY(hipDeviceSynchronize()); // nothing in flight may set it again behind our back
            __atomic_store_n(sk->h_err, 0u, __ATOMIC_RELEASE);
        }
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_server_key_release_stream(tfhe_hip_server_key *sk, void *stream_, size_t *workspaces_left)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    return guarded([&]() -> int {
        if (!sk) return fail("server key handle is null");
        HIP_TRY(hipSetDevice(sk->ctx->device));
        std::lock_guard<std::mutex> lock(sk->mu);
        auto it = sk->ws.find(stream);
        if (it != sk->ws.end() && stream != nullptr && stream != sk->pipe_stream[0] && stream != sk->pipe_stream[1]) {
            free_ws(it->second); // hipFree waits for the kernels that still use the buffers
            sk->ws.erase(it);
        }
        if (workspaces_left) *workspaces_left = sk->ws.size();
        return 0;
    });
}

TFHE_HIP_API int tfhe_hip_debug_skip_pair_signal(tfhe_hip_server_key *sk, int step)
{
    if (!sk) return fail("server key handle is null");
    for (tfhe_hip_server_key *peer : sk->peers) tfhe_hip_debug_skip_pair_signal(peer, step);
    std::lock_guard<std::mutex> lock(sk->mu);
    sk->debug_skip_signal = step;
    return 0;
}

TFHE_HIP_API int tfhe_hip_set_variant(tfhe_hip_server_key *sk, int variant)
{
    if (!sk) return fail("server key handle is null");
    if (variant < -1 || variant > 5) return fail("unknown blind-rotation variant");
    if (sk->grouping_factor) return fail("a multi-bit server key has a single blind-rotation kernel");
    if (variant != 0 && !sk->d_bsk_p) return fail("variants 1 / 2 / 3 (wave kernels) do not support these parameters");
    for (tfhe_hip_server_key *peer : sk->peers)
        if (tfhe_hip_set_variant(peer, variant)) return 1;
    std::lock_guard<std::mutex> lock(sk->mu);
    sk->variant = variant;
    return 0;
}
