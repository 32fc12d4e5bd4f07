// exact_ntt.hip -- EXPERIMENT (include/tfhe_hip_experimental.h): the external product of the blind
// rotation as EXACT integer arithmetic on the GPU -- a negacyclic number-theoretic transform modulo the "Goldilocks" prime
// p = 2^64 - 2^32 + 1 on 22-bit limbs of the standard-domain GGSW -- next to the f64 transform the reference (and the
// production kernels) use (fft64/crypto/ggsw.rs:477-598).  Same decomposition, rows and columns; the result equals the
// schoolbook product of the reference's integer code (polynomial_algorithms.rs:130-199) bit for bit (T0, deterministic on
// every machine), where the f64 path carries ~2^-26 of transform noise per product.  NOT a production path: it measures
// what determinism would cost: 2 forward + 6 inverse transforms of N = 2048 points with 64-bit modular
// multiplications (no 64-bit multiplier on CDNA4: four quarter-rate 32-bit multiplies each) against 2 + 2 complex f64
// transforms of 1024 points.
//
// One 256-thread workgroup per ciphertext; three 16 KiB polynomials in LDS (the two transformed digit polynomials and the
// working buffer); radix-2 stages with a workgroup barrier each (forward: decimation in frequency, natural in / bit-reversed
// out; inverse: decimation in time), so pointwise products need no permutation -- the layout of oracle/tfhe_oracle_ntt.c.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/tfhe_hip_experimental.h"

#define TFHE_HIP_API extern "C" __attribute__((visibility("default")))
extern "C" __attribute__((visibility("hidden"))) int tfhe_hip_set_error(const char *msg);

namespace {

constexpr uint64_t GP = 0xFFFFFFFF00000001ull;
constexpr uint64_t EPS = 0xFFFFFFFFull; // 2^64 mod p
constexpr int NN = 2048, THREADS = 256, LIMBS = 3;
__host__ __device__ constexpr int limb_shift(int l) { return 22 * l; }
__host__ __device__ constexpr uint64_t limb_mask(int l) { return l == 2 ? 0xFFFFFull : 0x3FFFFFull; }

__host__ __device__ inline uint64_t gl_add(uint64_t a, uint64_t b)
{
    uint64_t s = a + b;
    if (s < a || s >= GP) s -= GP;
    return s;
}
__host__ __device__ inline uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + GP - b; }
// x = hi 2^64 + lo  ->  x mod p:  2^64 = 2^32 - 1, 2^96 = -1  (mod p)
__host__ __device__ inline uint64_t gl_reduce(uint64_t hi, uint64_t lo)
{
    const uint64_t hh = hi >> 32, hl = hi & EPS;
    uint64_t t = lo - hh;
    if (lo < hh) t -= EPS; // wrapped: 2^64 was added, take 2^32 - 1 off again
    const uint64_t m = (hl << 32) - hl; // hl (2^32 - 1) < 2^64
    uint64_t r = t + m;
    if (r < t) r += EPS;
    return r >= GP ? r - GP : r;
}
__device__ inline uint64_t gl_mul(uint64_t a, uint64_t b) { return gl_reduce(__umul64hi(a, b), a * b); }
inline uint64_t gl_mul_host(uint64_t a, uint64_t b)
{
    const unsigned __int128 x = (unsigned __int128)a * b;
    return gl_reduce((uint64_t)(x >> 64), (uint64_t)x);
}
uint64_t gl_pow_host(uint64_t b, uint64_t e)
{
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul_host(r, b);
        b = gl_mul_host(b, b);
        e >>= 1;
    }
    return r;
}

// tables in global memory: psi^j [N], psi^-j / N [N], omega^k [N/2], omega^-k [N/2]
struct Tables {
    const uint64_t *psi, *psi_inv, *w, *w_inv;
};

// forward: natural in, bit-reversed out.  x: LDS polynomial of N residues, 256 threads, 4 butterflies per thread and stage
__device__ void ntt_fwd(uint64_t *x, const uint64_t *w, int tid)
{
    for (int len = NN / 2, step = 1; len >= 1; len >>= 1, step <<= 1) {
        __syncthreads();
        for (int t = tid; t < NN / 2; t += THREADS) {
            const int j = t & (len - 1), b = (t - j) << 1;
            const uint64_t u = x[b + j], v = x[b + j + len];
            x[b + j] = gl_add(u, v);
            x[b + j + len] = gl_mul(gl_sub(u, v), w[j * step]);
        }
    }
    __syncthreads();
}
__device__ void ntt_inv(uint64_t *x, const uint64_t *w_inv, int tid)
{
    for (int len = 1, step = NN / 2; len < NN; len <<= 1, step >>= 1) {
        __syncthreads();
        for (int t = tid; t < NN / 2; t += THREADS) {
            const int j = t & (len - 1), b = (t - j) << 1;
            const uint64_t u = x[b + j], v = gl_mul(x[b + j + len], w_inv[j * step]);
            x[b + j] = gl_add(u, v);
            x[b + j + len] = gl_sub(u, v);
        }
    }
    __syncthreads();
}

// GGSW limbs: dst[(poly * LIMBS + l) * N + ..] = NTT(limb l of polynomial `poly`), one workgroup per (poly, limb)
__global__ __launch_bounds__(THREADS) void ggsw_prepare_kernel(const uint64_t *__restrict__ ggsw_std, uint64_t *__restrict__ dst, Tables T)
{
    __shared__ uint64_t x[NN];
    const int poly = blockIdx.x / LIMBS, l = blockIdx.x % LIMBS, tid = threadIdx.x;
    for (int j = tid; j < NN; j += THREADS) x[j] = gl_mul((ggsw_std[(size_t)poly * NN + j] >> limb_shift(l)) & limb_mask(l), T.psi[j]);
    ntt_fwd(x, T.w, tid);
    for (int j = tid; j < NN; j += THREADS) dst[(size_t)blockIdx.x * NN + j] = x[j];
}

// acc[ct] += ggsw (x) in[ct], exactly (level count 1, k = 1).  ggsw_ntt: [row 2][col 2][limb 3][N]
__global__ __launch_bounds__(THREADS) void exact_external_product_kernel(const uint64_t *__restrict__ ggsw_ntt, const uint64_t *__restrict__ glwe_in,
                                                                         uint64_t *__restrict__ glwe_acc, int base_log, Tables T)
{
    __shared__ uint64_t D[2][NN];
    __shared__ uint64_t X[NN];
    const int tid = threadIdx.x;
    const uint64_t *in = glwe_in + (size_t)blockIdx.x * 2 * NN;
    uint64_t *acc = glwe_acc + (size_t)blockIdx.x * 2 * NN;
    const int shift = 64 - base_log - 1;
    const uint64_t mask = ((uint64_t)1 << base_log) - 1;
    for (int row = 0; row < 2; row++) {
        for (int j = tid; j < NN; j += THREADS) { // closest_representable + decompose_one_level (one level), as a residue times psi^j
            const uint64_t st = ((in[row * NN + j] >> shift) + 1) >> 1;
            const uint64_t res = st & mask, rest = st >> base_log;
            const uint64_t carry = (((res - 1) | rest) & res) >> (base_log - 1);
            const int64_t d = (int64_t)(res - (carry << base_log));
            D[row][j] = gl_mul(d >= 0 ? (uint64_t)d : GP - (uint64_t)(-d), T.psi[j]);
        }
        ntt_fwd(D[row], T.w, tid);
    }
    for (int col = 0; col < 2; col++)
        for (int l = 0; l < LIMBS; l++) {
            const uint64_t *g0 = ggsw_ntt + (size_t)((0 * 2 + col) * LIMBS + l) * NN, *g1 = ggsw_ntt + (size_t)((1 * 2 + col) * LIMBS + l) * NN;
            __syncthreads();
            for (int j = tid; j < NN; j += THREADS) X[j] = gl_add(gl_mul(D[0][j], g0[j]), gl_mul(D[1][j], g1[j]));
            ntt_inv(X, T.w_inv, tid);
            for (int j = tid; j < NN; j += THREADS) {
                const uint64_t a = gl_mul(X[j], T.psi_inv[j]);
                const uint64_t v = a > GP / 2 ? (uint64_t)0 - (GP - a) : a; // centred lift: |value| < 2^56
                acc[col * NN + j] += v << limb_shift(l);
            }
        }
}

#define HIP_TRY(expr)                                                                                             \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return tfhe_hip_set_error((std::string(#expr) + " failed: " + hipGetErrorString(e_)).c_str()); \
    } while (0)

} // namespace

TFHE_HIP_API int tfhe_hip_exact_external_product_batch(int device_id, const uint64_t *ggsw_std, uint32_t pbs_base_log, const uint64_t *glwe_in,
                                                       uint64_t *glwe_acc, size_t B, int timing_reps, double *kernel_ms)
{
    if (!ggsw_std || !glwe_in || !glwe_acc) return tfhe_hip_set_error("null pointer argument");
    if (pbs_base_log < 2 || pbs_base_log > 23) return tfhe_hip_set_error("exact external product: 2 <= pbs_base_log <= 23 (limb products must stay below 2^56)");
    if (B == 0) return 0;
    HIP_TRY(hipSetDevice(device_id));
    // tables (oracle/tfhe_oracle_ntt.c: psi = 7^((p - 1) / 2N))
    std::vector<uint64_t> tab(3 * NN);
    const uint64_t psi = gl_pow_host(7, (GP - 1) / (2ull * NN)), psi_inv = gl_pow_host(psi, GP - 2), n_inv = gl_pow_host(NN, GP - 2);
    uint64_t a = 1, b = n_inv;
    for (int j = 0; j < NN; j++) {
        tab[j] = a; tab[NN + j] = b;
        a = gl_mul_host(a, psi); b = gl_mul_host(b, psi_inv);
    }
    const uint64_t om = gl_mul_host(psi, psi), om_inv = gl_mul_host(psi_inv, psi_inv);
    a = 1; b = 1;
    for (int k = 0; k < NN / 2; k++) {
        tab[2 * NN + k] = a; tab[2 * NN + NN / 2 + k] = b;
        a = gl_mul_host(a, om); b = gl_mul_host(b, om_inv);
    }
    uint64_t *d_tab = nullptr, *d_std = nullptr, *d_ntt = nullptr, *d_in = nullptr, *d_acc = nullptr;
    auto release = [&]() { (void)hipFree(d_tab); (void)hipFree(d_std); (void)hipFree(d_ntt); (void)hipFree(d_in); (void)hipFree(d_acc); };
#define TRY_R(expr)                                                                                               \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            release();                                                                                            \
            return tfhe_hip_set_error((std::string(#expr) + " failed: " + hipGetErrorString(e_)).c_str());        \
        }                                                                                                         \
    } while (0)
    TRY_R(hipMalloc((void **)&d_tab, tab.size() * 8));
    TRY_R(hipMalloc((void **)&d_std, 4 * NN * 8));
    TRY_R(hipMalloc((void **)&d_ntt, 4 * LIMBS * NN * 8));
    TRY_R(hipMalloc((void **)&d_in, B * 2 * NN * 8));
    TRY_R(hipMalloc((void **)&d_acc, B * 2 * NN * 8));
    TRY_R(hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    TRY_R(hipMemcpy(d_std, ggsw_std, 4 * NN * 8, hipMemcpyHostToDevice));
    TRY_R(hipMemcpy(d_in, glwe_in, B * 2 * NN * 8, hipMemcpyHostToDevice));
    TRY_R(hipMemcpy(d_acc, glwe_acc, B * 2 * NN * 8, hipMemcpyHostToDevice));
    const Tables T{d_tab, d_tab + NN, d_tab + 2 * NN, d_tab + 2 * NN + NN / 2};
    ggsw_prepare_kernel<<<4 * LIMBS, THREADS>>>(d_std, d_ntt, T);
    exact_external_product_kernel<<<(unsigned)B, THREADS>>>(d_ntt, d_in, d_acc, (int)pbs_base_log, T);
    TRY_R(hipGetLastError());
    TRY_R(hipMemcpy(glwe_acc, d_acc, B * 2 * NN * 8, hipMemcpyDeviceToHost));
    if (timing_reps > 0 && kernel_ms) { // the accumulator keeps growing: only the duration is of interest
        hipEvent_t e0, e1;
        TRY_R(hipEventCreate(&e0));
        TRY_R(hipEventCreate(&e1));
        TRY_R(hipEventRecord(e0, nullptr));
        for (int r = 0; r < timing_reps; r++) exact_external_product_kernel<<<(unsigned)B, THREADS>>>(d_ntt, d_in, d_acc, (int)pbs_base_log, T);
        TRY_R(hipEventRecord(e1, nullptr));
        TRY_R(hipEventSynchronize(e1));
        float ms = 0;
        TRY_R(hipEventElapsedTime(&ms, e0, e1));
        *kernel_ms = ms / timing_reps;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    release();
    return 0;
}
