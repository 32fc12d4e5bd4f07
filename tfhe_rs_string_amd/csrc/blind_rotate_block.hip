// blind_rotate_block.hip -- variant 3 of the blind rotation: one 512-thread workgroup per ciphertext
// (lane-level source in block_pbs.hpp).  The latency kernel for batches that do not fill the chip:
// a CMUX step of one ciphertext is spread over the 8 wavefronts (all 4 SIMDs) of a CU.
// 64 KiB of LDS and <= 128 registers would allow two workgroups per CU; the kernel is built for one
// (<= 256 registers, twiddles and the bootstrap-key prefetch in registers).
#include "kernels.hpp"
#include "block_pbs.hpp"
#include "lane_xchg.hpp"

namespace tfhe_hip {

using namespace blk;

using xchg::transpose54;
using xchg::wave_sync;

// DIAGNOSTIC BUILDS ONLY (wrong results; tools/build_stamps.sh): what the workgroup barriers cost
#if defined(TFHE_BLOCK_DIAG_NOBARRIER)
#define BLOCK_BARRIER() wave_sync()
#else
#define BLOCK_BARRIER() __syncthreads()
#endif

template <int BASE_LOG>
__global__ __launch_bounds__(THREADS) void blind_rotate_block_kernel(const uint64_t *__restrict__ lwe_small,
                                                                      const uint64_t *__restrict__ luts,
                                                                      const uint32_t *__restrict__ lut_idx,
                                                                      const c64 *__restrict__ bsk_b,
                                                                      const c64 *__restrict__ twist,
                                                                      const c64 *__restrict__ W,
                                                                      uint64_t *__restrict__ lwe_out, int n, const BrCtl ctl)
{
    __shared__ __attribute__((aligned(256))) char lds[LDS_BYTES];
    const Ids d = make_ids((int)threadIdx.x);
    const int ct = blockIdx.x;
    const uint64_t *lwe = lwe_small + (size_t)ct * (n + 1);
    const uint64_t *lut = luts + (size_t)checked_lut_idx(lut_idx, ct, ctl.n_luts, ctl.err) * (2 * N);

    Lane s;
    phase_init(s, lds, lut, twist, W, modswitch(lwe[n], LOG2N), d);
    ggsw_prefetch(s, bsk_b, 0, d);
    __syncthreads();

#ifdef TFHE_BLOCK_STAMPS
    // DIAGNOSTIC BUILD ONLY (never shipped, the output is overwritten): per-phase clock sums of wavefront 0
    unsigned long long t_acc[12] = {0}, t_prev = __builtin_amdgcn_s_memtime();
#define STAMP(k)                                                                                     \
    do {                                                                                             \
        __builtin_amdgcn_s_waitcnt(0);                                                               \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();                               \
        t_acc[k] += t_now - t_prev;                                                                  \
        t_prev = t_now;                                                                              \
    } while (0)
#else
#define STAMP(k) ((void)0)
#endif
    for (int i = 0; i < n; i++) {
        const uint32_t a_hat = modswitch(lwe[i], LOG2N);
        phase_rotate_stage1<BASE_LOG>(s, lds, a_hat, d);
        STAMP(0);
        BLOCK_BARRIER(); // blocks of 256 positions change wavefront
        STAMP(1);
        phase_fwd2(s, lds, d);
        transpose54(s);
        phase_fwd3(s, lds, d);
        wave_sync();
        STAMP(2);
        phase_fwd4(s, lds, d);
        STAMP(3);
        BLOCK_BARRIER(); // both Fourier polynomials complete
        STAMP(4);
        phase_pointwise(s, lds, d);
        if (i + 1 < n) ggsw_prefetch(s, bsk_b, (size_t)i + 1, d); // a whole CMUX step ahead of its use
        wave_sync();
#ifdef TFHE_BLOCK_STAMPS
        { // without draining the prefetch
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();
            t_acc[5] += t_now - t_prev;
            t_prev = t_now;
        }
#endif
        phase_inv4(s, lds, d);
        wave_sync();
        phase_inv3(s, lds, d);
        transpose54(s);
        phase_inv2(s, lds, d);
#ifdef TFHE_BLOCK_STAMPS
        {
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();
            t_acc[6] += t_now - t_prev;
            t_prev = t_now;
        }
#endif
        BLOCK_BARRIER();
#ifdef TFHE_BLOCK_STAMPS
        {
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();
            t_acc[7] += t_now - t_prev;
            t_prev = t_now;
        }
#endif
        phase_inv1_accumulate(s, lds, d);
#ifdef TFHE_BLOCK_STAMPS
        {
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();
            t_acc[8] += t_now - t_prev;
            t_prev = t_now;
        }
#endif
        BLOCK_BARRIER(); // mirror of the new accumulator complete
#ifdef TFHE_BLOCK_STAMPS
        {
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();
            t_acc[9] += t_now - t_prev;
            t_prev = t_now;
        }
#endif
    }
    phase_sample_extract(s, lwe_out + (size_t)ct * (N + 1), d);
#ifdef TFHE_BLOCK_STAMPS
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < 12; k++) lwe_out[(size_t)ct * (N + 1) + 64 + k] = t_acc[k];
#endif
}

// variant 4: the same phases with HALF the threads -- a 256-thread workgroup per ciphertext whose thread (q, l) does the
// work of the two threads (p = 0 / 1, q, l) of the kernel above, in the PACKED LDS layout (64 KiB), so that TWO
// workgroups share a CU (two wavefronts per SIMD from two independent ciphertexts: one's barriers and LDS stores run
// under the other's arithmetic).  The twiddles and twist factors are the same for both polynomials of a thread; stage 5
// of the two Fourier polynomials is done once instead of in both groups.  For levels of 257 .. 512 bootstraps.
template <int BASE_LOG>
__global__ __launch_bounds__(256, 2) void blind_rotate_half_kernel(const uint64_t *__restrict__ lwe_small,
                                                                    const uint64_t *__restrict__ luts,
                                                                    const uint32_t *__restrict__ lut_idx,
                                                                    const c64 *__restrict__ bsk_b,
                                                                    const c64 *__restrict__ twist,
                                                                    const c64 *__restrict__ W,
                                                                    uint64_t *__restrict__ lwe_out, int n, const BrCtl ctl)
{
    __shared__ __attribute__((aligned(256))) char lds[LDS_BYTES_PACKED];
    const int tid = (int)threadIdx.x;
    Ids d0, d1;
    d0.p = 0; d0.q = tid >> 6; d0.l = tid & 63; d0.T = tid;
    d1 = d0;
    d1.p = 1;
    const int ct = blockIdx.x;
    const uint64_t *lwe = lwe_small + (size_t)ct * (n + 1);
    const uint64_t *lut = luts + (size_t)checked_lut_idx(lut_idx, ct, ctl.n_luts, ctl.err) * (2 * N);

    Lane s0, s1;
    const uint32_t b_hat = modswitch(lwe[n], LOG2N);
    phase_init<true>(s0, lds, lut, twist, W, b_hat, d0);
    phase_init<true>(s1, lds, lut, twist, W, b_hat, d1);
#pragma unroll
    for (int t = 0; t < 4; t++) s1.tw[t] = s0.tw[t]; // one copy of the loop constants
#pragma unroll
    for (int m = 0; m < 3; m++) {
        s1.w1[m] = s0.w1[m];
        s1.w2[m] = s0.w2[m];
        s1.w3[m] = s0.w3[m];
        s1.w4[m] = s0.w4[m];
    }
    ggsw_prefetch(s0, bsk_b, 0, d0);
    ggsw_prefetch(s1, bsk_b, 0, d1);
    __syncthreads();
    for (int i = 0; i < n; i++) {
        const uint32_t a_hat = modswitch(lwe[i], LOG2N);
        phase_rotate_stage1<BASE_LOG, true>(s0, lds, a_hat, d0);
        phase_rotate_stage1<BASE_LOG, true>(s1, lds, a_hat, d1);
        __syncthreads(); // blocks of 256 positions change wavefront
        phase_fwd2(s0, lds, d0);
        phase_fwd2(s1, lds, d1);
        transpose54(s0);
        transpose54(s1);
        phase_fwd3(s0, lds, d0);
        phase_fwd3(s1, lds, d1);
        wave_sync();
        phase_fwd4(s0, lds, d0);
        phase_fwd4(s1, lds, d1);
        wave_sync(); // both Fourier polynomials of a block of 256 positions belong to this wavefront
        phase_pointwise_both(s0, s1, lds, d0);
        if (i + 1 < n) {
            ggsw_prefetch(s0, bsk_b, (size_t)i + 1, d0); // a whole CMUX step ahead of its use
            ggsw_prefetch(s1, bsk_b, (size_t)i + 1, d1);
        }
        wave_sync();
        phase_inv4(s0, lds, d0);
        phase_inv4(s1, lds, d1);
        wave_sync();
        phase_inv3(s0, lds, d0);
        phase_inv3(s1, lds, d1);
        transpose54(s0);
        transpose54(s1);
        phase_inv2(s0, lds, d0);
        phase_inv2(s1, lds, d1);
        __syncthreads();
        phase_inv1_accumulate<true>(s0, lds, d0);
        phase_inv1_accumulate<true>(s1, lds, d1);
        __syncthreads(); // mirror of the new accumulator complete
    }
    phase_sample_extract(s0, lwe_out + (size_t)ct * (N + 1), d0);
    phase_sample_extract(s1, lwe_out + (size_t)ct * (N + 1), d1);
}

// dst[i][p][q][e = row * 4 + t][lane] = BSK_PRESCALE * src[i][row][col = p][256 q + 4 lane + t]   (level count 1)
__global__ void bsk_permute_block_kernel(const c64 *__restrict__ src, c64 *__restrict__ dst, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int e = (int)((idx >> 6) & 7);
    const int q = (int)((idx >> 9) & 3);
    const int p = (int)((idx >> 11) & 1);
    const size_t i = idx >> 12;
    const c64 g = src[(i * 4 + (e >> 2) * 2 + p) * FN + 256 * q + 4 * lane + (e & 3)];
    c64 o;
    o.re = g.re * BSK_PRESCALE; // exact (power of two): block_pbs.hpp
    o.im = g.im * BSK_PRESCALE;
    dst[idx] = o;
}

hipError_t launch_bsk_permute_block(const DevParams &p, const double *d_bsk_f, double *d_bsk_b, hipStream_t s)
{
    if (!blind_rotate_wave_supported(p)) return hipErrorInvalidValue;
    const size_t total = (size_t)p.n * 4 * FN;
    bsk_permute_block_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>((const c64 *)d_bsk_f, (c64 *)d_bsk_b, total);
    return hipGetLastError();
}

hipError_t launch_blind_rotate_block(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts,
                                     const uint32_t *d_lut_idx, const double *d_bsk_b, const double *d_twist,
                                     const double *d_w, uint64_t *d_lwe_out, size_t B, const BrCtl &ctl, hipStream_t s)
{
    if (!blind_rotate_wave_supported(p) || B == 0) return hipErrorInvalidValue;
    blind_rotate_block_kernel<23><<<(unsigned)B, THREADS, 0, s>>>(d_lwe_small, d_luts, d_lut_idx, (const c64 *)d_bsk_b,
                                                                   (const c64 *)d_twist, (const c64 *)d_w, d_lwe_out,
                                                                   p.n, ctl);
    return hipGetLastError();
}

hipError_t launch_blind_rotate_half(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts,
                                    const uint32_t *d_lut_idx, const double *d_bsk_b, const double *d_twist,
                                    const double *d_w, uint64_t *d_lwe_out, size_t B, const BrCtl &ctl, hipStream_t s)
{
    if (!blind_rotate_wave_supported(p) || B == 0) return hipErrorInvalidValue;
    blind_rotate_half_kernel<23><<<(unsigned)B, 256, 0, s>>>(d_lwe_small, d_luts, d_lut_idx, (const c64 *)d_bsk_b,
                                                             (const c64 *)d_twist, (const c64 *)d_w, d_lwe_out, p.n, ctl);
    return hipGetLastError();
}

} // namespace tfhe_hip
